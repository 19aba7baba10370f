"""Host-side scene API of the product (the step before the path) against the oracle's restatement, byte for byte."""
import ctypes as C
import os

import numpy as np
import pytest

from cases import TEDDY


@pytest.mark.parametrize("mp", [(8, 8), (3, 3), (16, 9), (5, 4), (64, 33)])
def test_sphere_prerender_matches_oracle(rt3, oracle, mp):
    m, p = mp
    e = rt3.create_sphere((-2.0, 0.3, -5.0), 1.25, m, p, (0.2, 0.4, 1.0))
    assert e.pre_render_faces == 2 * m * (p - 2) and e.pre_render_vertices == 2 + (p - 2) * m      # Sphere.cpp:101-102
    f, v = rt3.pre_render_entity(e)
    of, ov = oracle.prerender_sphere((-2.0, 0.3, -5.0), 1.25, m, p, (0.2, 0.4, 1.0))
    assert f.tobytes() == of.tobytes() and v.tobytes() == ov.tobytes()
    assert (v[:, 3] == 0).all()
    # every face's normal is unit length and the baked colour is colour * |n.z| (Sphere.cpp:155)
    assert np.allclose(np.linalg.norm(f["normal"], axis=1), 1.0, atol=1e-6)
    assert np.array_equal(f["color"], np.float32([0.2, 0.4, 1.0])[None, :] * np.abs(f["normal"][:, 2:3] * np.float32(-1)))


def test_triangle_prerender_matches_oracle(rt3, oracle):
    pts = ((1.0, 0.0, -3.0), (-1.0, 0.25, -3.5), (0.0, 1.0, -2.5))
    f, v = rt3.pre_render_entity(rt3.create_triangle(*pts, (0.9, 0.1, 0.3)))
    of, ov = oracle.prerender_triangle(*pts, (0.9, 0.1, 0.3))
    assert f.tobytes() == of.tobytes() and v.tobytes() == ov.tobytes()
    assert tuple(f["color"][0]) == (np.float32(0.9), np.float32(0.1), np.float32(0.3))          # NOT headlight-shaded


def write_obj(path, one_based=True):
    rng = np.random.RandomState(5)
    verts = rng.uniform(-3, 3, (40, 3))
    with open(path, "w") as fh:
        for v in verts:
            fh.write("v %.6f %.6f %.6f\n" % tuple(v))
        for _ in range(60):
            a, b, c = rng.choice(40, 3, replace=False) + (1 if one_based else 0)
            fh.write("f %d %d %d\n" % (a, b, c))
        if one_based:
            fh.write("f 1 2 3\n")
        else:
            fh.write("f 0 1 2\n")


@pytest.mark.parametrize("one_based", [True, False])
def test_object_prerender_matches_oracle(rt3, oracle, tmp_path, one_based):
    path = str(tmp_path / "mesh.obj")
    write_obj(path, one_based)
    e = rt3.create_object(path, (0.5, -0.25, -6.0), 0.37, (1.0, 0.5, 0.0))
    assert (e.pre_render_faces, e.pre_render_vertices) == (61, 40)
    f, v = rt3.pre_render_entity(e)
    of, ov = oracle.prerender_object(path, (0.5, -0.25, -6.0), np.float32(0.37), (1.0, 0.5, 0.0))
    assert f.tobytes() == of.tobytes() and v.tobytes() == ov.tobytes()
    assert min(f["v1"].min(), f["v2"].min(), f["v3"].min()) == 0                                 # rebased (Object.cpp:181-192)


def test_object_errors_are_fatal(rt3, tmp_path):
    with pytest.raises(rt3.Fatal):
        rt3.create_object(str(tmp_path / "missing.obj"), (0, 0, 0), 1.0, (1, 1, 1))
    bad = tmp_path / "bad.obj"
    bad.write_text("v 1 2 3\n# a comment line is unreadable to the reference (Object.cpp:157-159)\n")
    with pytest.raises(rt3.Fatal):
        rt3.create_object(str(bad), (0, 0, 0), 1.0, (1, 1, 1))


@pytest.mark.parametrize("body", [
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n",            # a face names a vertex the file does not hold
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -3\n",           # negative index
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 1e20\n",         # not a 32-bit index
    "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 nan\n",
])
def test_malformed_object_indices_are_fatal_not_out_of_bounds(rt3, tmp_path, body):
    """The reference indexes through Tools::Array and casts blindly (Object.cpp:157-170, :188-194); this port checks
    the rebased indices against the vertex count before it reads through raw pointers."""
    path = tmp_path / "bad_index.obj"
    path.write_text(body)
    try:
        e = rt3.create_object(str(path), (0, 0, 0), 1.0, (1, 1, 1))
    except rt3.Fatal:
        return                                              # "nan" does not parse as a float with every libc: also fatal
    with pytest.raises(rt3.Fatal):
        rt3.pre_render_entity(e)


def test_teddy_matches_oracle(rt3, oracle):
    if not os.path.exists(TEDDY):
        pytest.skip("/root/reference not present (GPU box)")
    args = ((0.0, 0.0, -3.0), np.float32(1.0) / np.float32(17.0), (1.0, 0.0, 0.0))
    f, v = rt3.pre_render_entity(rt3.create_object(TEDDY, *args))
    of, ov = oracle.prerender_object(TEDDY, *args)
    assert (len(f), len(v)) == (3192, 1598)
    assert f.tobytes() == of.tobytes() and v.tobytes() == ov.tobytes()


def test_merge_rebases_indices_in_entity_order(rt3, oracle):
    ents = [rt3.create_triangle((1, 0, -3), (-1, 0, -3), (0, 1, -3), (1, 0, 0)),
            rt3.create_sphere((0, 0, -4), 1.0, 6, 5, (0, 1, 0)),
            rt3.create_triangle((2, 0, -3), (1, 0, -3), (1.5, 1, -3), (0, 0, 1))]
    parts = [rt3.pre_render_entity(e) for e in ents]
    faces, verts = rt3.merge_entities(parts)
    of, ov = oracle.merge([(f.view(oracle.GFACE), v) for f, v in parts])
    assert faces.tobytes() == of.tobytes() and verts.tobytes() == ov.tobytes()
    assert tuple(faces[0][["v1", "v2", "v3"]]) == (0, 1, 2)
    assert faces["v1"][1:1 + 36].min() >= 3                     # sphere indices rebased by the triangle's 3 vertices
    assert tuple(faces[-1][["v1", "v2", "v3"]]) == (3 + 20, 4 + 20, 5 + 20)


def test_camera_update_matches_reference_formula(rt3, oracle):
    for w, h in ((400, 225), (1920, 1080), (800, 600)):
        cam = rt3.main_camera(w, h)
        ocam = oracle.camera_update(w, h)
        assert bytes(cam.c) == bytes(ocam)
        vw = np.float32(np.float32(w) / np.float32(h)) * np.float32(2)
        assert cam.origin == (0, 0, 0) and cam.horizontal == (vw, 0, 0) and cam.vertical == (0, 2, 0)
        assert cam.lower_left_corner == (np.float32(0) - vw / np.float32(2), -1.0, -2.0)      # Camera.cpp:92
        assert (cam.w(), cam.h()) == (w, h)


def test_look_at_reproduces_update_for_the_default_pose(rt3):
    # looking down -z from the origin with a 90-degree vertical fov is Camera::update(focal 1, vh 2)
    a = rt3.Camera().look_at(200, 100, (0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 1.0)
    b = rt3.Camera().update(200, 100, 1.0, 4.0, 2.0)
    for f in ("origin", "horizontal", "vertical", "lower_left_corner"):
        assert np.allclose(getattr(a, f), getattr(b, f), atol=1e-6)


def test_ppm_bytes_match_frame_to_ppm(rt3, oracle, tmp_path):
    f = rt3.Frame(5, 3)
    f.data[:] = np.arange(15, dtype=np.uint32).reshape(3, 5) * 0x01020304 + 0x10
    want = b"P6\n# Image rendered by the RayTracer-3\n5 3\n255\n" + bytes(
        b for px in f.data.reshape(-1) for b in ((int(px) >> 24) & 255, (int(px) >> 16) & 255, (int(px) >> 8) & 255))
    assert f.ppm_bytes() == want == oracle.ppm_bytes(f.data)
    path = str(tmp_path / "out.ppm")
    f.to_ppm(path)
    assert open(path, "rb").read() == want
    with pytest.raises(rt3.Fatal):
        f.to_ppm(str(tmp_path / "no_such_dir" / "out.ppm"))


def test_benchmark_scenes_are_deterministic(rt3):
    cr, m = rt3.scene_weekend(42)
    cr2, m2 = rt3.scene_weekend(42)
    assert cr.tobytes() == cr2.tobytes() and m.tobytes() == m2.tobytes()
    assert 400 < len(cr) <= 488 and (cr[:, 3] > 0).all()
    assert cr[0, 3] == 1000.0 and set(np.unique(m["kind"])) == {1, 2, 3}
    d = np.linalg.norm(cr[1:-3, :3] - np.float32([4, 0.2, 0]), axis=1)
    assert (d > 0.9).all()                                       # the book's exclusion zone
    assert rt3.scene_weekend(43)[0].tobytes() != cr.tobytes()
    s, sm = rt3.scene_stress(1000, 43)
    assert len(s) == 1000 and (sm["kind"] == 1).all() and s[:, 3].min() >= 0.05 and s[:, 3].max() <= 0.4
    faces, verts, fm = rt3.scene_cornell(64)
    assert 45000 < len(faces) < 55000 and len(verts) == 3 * len(faces)
    assert (fm["kind"] == 0).sum() == 2 and fm["rgb"][fm["kind"] == 0].max() == 15.0   # the emissive quad
