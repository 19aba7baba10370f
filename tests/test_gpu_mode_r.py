"""Mode R on the device == the reference (through the pinned oracle), bit for bit, called through the C ABI."""
import json
import os

import numpy as np
import pytest

from cases import GOLDEN, load_builtin_scene

pytestmark = pytest.mark.gpu
PINS = json.load(open(os.path.join(GOLDEN, "reference_pins.json")))


def hip_mode_r(rt3, renderer, faces, verts, w, h, cam=None):
    renderer.set_mesh(faces.view(rt3.GFACE), verts)
    renderer.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))
    cam = cam or rt3.main_camera(w, h)
    renderer.configure(spp=None)
    renderer.render(cam)
    return cam.get_frame().d().copy()


@pytest.mark.parametrize("size", ["400x225", "1920x1080"])
def test_builtin_scene_reproduces_the_reference_ppm(rt3, renderer, oracle, size):
    """HIP render of the scene of src/Main.cpp:280-283 -> Frame::to_ppm bytes -> SHA-256 recorded from the reference."""
    w, h = map(int, size.split("x"))
    faces, verts = load_builtin_scene()
    img = hip_mode_r(rt3, renderer, faces, verts, w, h)
    ref_like = img.copy()
    ref_like[h - 1] = 0                                   # the reference's CPU loop never writes row H-1 (:286)
    f = rt3.Frame(w, h)
    f.data[:] = ref_like
    assert oracle.sha256(f.ppm_bytes()) == PINS["ppm_sha256"][size]
    # row H-1 follows the GLSL twin (v = 0, raytracer_v3.glsl:193-196): equal to the oracle asked for that row
    last = oracle.render_mode_r(faces, verts, oracle.camera_update(w, h), w, h, h - 1, h)
    assert np.array_equal(img[h - 1], last[h - 1])
    st = renderer.stats()
    assert st.prim_tests == w * h * len(faces) and st.trace_ms > 0


def test_plain_and_filtered_kernels_agree(rt3, renderer, oracle):
    """k_mode_r (brute force, any camera) and k_mode_r_fast (bounding-sphere filter, camera at the origin) give the
    same pixels; the filter is conservative, the exact test is the reference's in both."""
    faces, verts = load_builtin_scene()
    fast = hip_mode_r(rt3, renderer, faces, verts, 640, 360)
    renderer.force_plain_mode_r(True)
    try:
        plain = hip_mode_r(rt3, renderer, faces, verts, 640, 360)
    finally:
        renderer.force_plain_mode_r(False)
    assert np.array_equal(fast, plain)
    assert np.array_equal(fast, oracle.render_mode_r(faces, verts, oracle.camera_update(640, 360), 640, 360))


def test_small_goldens(rt3, renderer, oracle):
    z = np.load(os.path.join(GOLDEN, "mode_r_small.npz"))
    tri = oracle.prerender_triangle((1.0, 0.0, -3.0), (-1.0, 0.0, -3.0), (0.0, 1.0, -3.0), (1.0, 0.0, 0.0))
    sph = oracle.prerender_sphere((0.0, 0.0, -3.0), 1.0, 8, 8, (1.0, 0.0, 0.0))
    assert np.array_equal(hip_mode_r(rt3, renderer, tri[0], tri[1], 64, 36), z["triangle"])
    assert np.array_equal(hip_mode_r(rt3, renderer, sph[0], sph[1], 64, 36), z["sphere8x8"])


def test_entity_api_end_to_end(rt3, renderer, oracle):
    """prerender(entities) -> render(camera) exactly as Main.cpp:280-285 drives the backend."""
    ents = [rt3.create_sphere((-0.5, 0.0, -4.0), 1.0, 24, 17, (0.0, 0.0, 1.0)),
            rt3.create_triangle((1.5, -0.5, -3.0), (0.2, -0.5, -3.0), (0.8, 0.9, -3.5), (1.0, 0.0, 0.0)),
            rt3.create_sphere((1.0, 0.6, -6.0), 1.5, 40, 40, (0.2, 0.9, 0.1))]
    renderer.prerender(ents)
    renderer.configure(spp=None)
    cam = rt3.main_camera(333, 187)                       # ragged: not a multiple of the block or wave size
    renderer.render(cam)
    faces, verts = oracle.merge([oracle.prerender_sphere((-0.5, 0.0, -4.0), 1.0, 24, 17, (0.0, 0.0, 1.0)),
                                 oracle.prerender_triangle((1.5, -0.5, -3.0), (0.2, -0.5, -3.0), (0.8, 0.9, -3.5), (1.0, 0.0, 0.0)),
                                 oracle.prerender_sphere((1.0, 0.6, -6.0), 1.5, 40, 40, (0.2, 0.9, 0.1))])
    ref = oracle.render_mode_r(faces, verts, oracle.camera_update(333, 187), 333, 187)
    assert renderer.n_faces == len(faces) > 512           # several LDS tiles
    assert np.array_equal(cam.get_frame().d(), ref)


def test_empty_scene_is_all_sky(rt3, renderer, oracle):
    empty_f, empty_v = np.zeros(0, oracle.GFACE), np.zeros((0, 4), np.float32)
    img = hip_mode_r(rt3, renderer, empty_f, empty_v, 97, 41)
    ref = oracle.render_mode_r(empty_f, empty_v, oracle.camera_update(97, 41), 97, 41)
    assert np.array_equal(img, ref)


def test_ties_degenerate_faces_and_edge_hits(rt3, renderer, oracle):
    t = ((1.0, -1.0, -3.0), (-1.0, -1.0, -3.0), (0.0, 1.0, -3.0))
    parts = [oracle.prerender_triangle(*t, (1.0, 0.0, 0.0)), oracle.prerender_triangle(*t, (0.0, 1.0, 0.0)),      # coincident
             oracle.prerender_triangle((0, 0, -2), (0, 0, -2), (0, 0, -2), (0.0, 0.0, 1.0)),                     # NaN normal
             oracle.prerender_triangle((-3.0, -1.0, -3.0), (-1.0, -1.0, -3.0), (-2.0, 1.0, -3.0), (1, 1, 0)),   # shares a vertex
             oracle.prerender_triangle((2.0, 0.0, -2.0), (2.0, 1.0, -4.0), (2.0, -1.0, -4.0), (0, 1, 1))]       # edge-on-ish
    faces, verts = oracle.merge(parts)
    img = hip_mode_r(rt3, renderer, faces, verts, 256, 144)
    ref = oracle.render_mode_r(faces, verts, oracle.camera_update(256, 144), 256, 144)
    assert np.array_equal(img, ref)
    centre = int(img[72, 128])
    assert (centre >> 8) == 0xFF0000                      # the FIRST of the coincident triangles wins (red)


def test_off_origin_camera_keeps_the_reference_formula(rt3, renderer, oracle):
    """SequentialRenderer.cpp:70 adds n.o; with a camera off the origin the image is 'wrong' in the reference, and the
    device reproduces exactly that."""
    faces, verts = oracle.prerender_sphere((0.0, 0.0, -4.0), 1.0, 12, 9, (1.0, 0.0, 0.0))
    cam = rt3.main_camera(120, 68)
    cam.c.origin[0], cam.c.origin[2] = 0.25, 0.5
    ocam = oracle.copy_camera(cam.c)
    img = hip_mode_r(rt3, renderer, faces, verts, 120, 68, cam)
    assert np.array_equal(img, oracle.render_mode_r(faces, verts, ocam, 120, 68))


def test_device_output_on_a_torch_stream(rt3, renderer, oracle):
    import torch
    faces, verts = load_builtin_scene()
    renderer.set_mesh(faces.view(rt3.GFACE), verts)
    w, h = 160, 90
    cam = rt3.main_camera(w, h)
    out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        renderer.render_device(cam.c, w, h, out.data_ptr(), s.cuda_stream)
    s.synchronize()
    ref = oracle.render_mode_r(faces, verts, oracle.camera_update(w, h), w, h)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref)


def test_bad_arguments_fail_loudly(rt3, renderer):
    faces, verts = load_builtin_scene()
    bad = faces.copy()
    bad["v2"][7] = len(verts) + 5
    with pytest.raises(rt3.Fatal, match="out of range"):
        renderer.set_mesh(bad.view(rt3.GFACE), verts)
    with pytest.raises(rt3.Fatal):
        renderer.render(rt3.main_camera(1, 1))


def _odd_faces(rt3):
    """Hand-made GFace records the pre-render never produces but the interface accepts: a face whose three vertices coincide
    (with a VALID stored normal the reference's edge tests are -0 >= 0 everywhere: the face is an infinite plane), a face with
    two coincident vertices, three collinear vertices, and a normal that is not perpendicular to its triangle (the reference
    tests the projection along the stored normal).  Plus one ordinary triangle behind them."""
    # (vertex order: the reference's faces have n = normalize(cross(p3 - p1, p2 - p1)), Sphere.cpp:153; a face wound the other
    #  way is never hit)
    tris = [((-2.0, -1.5, -3.0), (0.0, 1.5, -3.0), (2.0, -1.5, -3.0), (0.0, 0.0, 1.0), (0.1, 0.2, 0.9)),      # ordinary, in front
            ((0.3, 0.2, -4.0), (0.3, 0.2, -4.0), (0.3, 0.2, -4.0), (0.0, 0.6, 0.8), (0.9, 0.1, 0.1)),         # a point with a normal
            ((-1.0, 0.5, -2.0), (-1.0, 0.5, -2.0), (-0.5, 1.0, -2.0), (0.0, 0.0, 1.0), (0.1, 0.9, 0.1)),      # two coincident vertices
            ((-1.5, -1.0, -2.0), (0.0, -1.0, -2.0), (1.5, -1.0, -2.0), (0.0, 0.0, 1.0), (0.9, 0.9, 0.1)),     # collinear
            ((1.2, -0.2, -2.5), (1.7, 0.8, -2.5), (2.2, -0.2, -2.5), (0.6, 0.0, 0.8), (0.9, 0.1, 0.9))]       # skewed normal
    n = len(tris)
    faces = np.zeros(n, rt3.GFACE)
    verts = np.zeros((3 * n, 4), np.float32)
    for i, (a, b, c, nrm, col) in enumerate(tris):
        verts[3 * i:3 * i + 3, :3] = (a, b, c)
        faces[i]["v1"], faces[i]["v2"], faces[i]["v3"] = 3 * i, 3 * i + 1, 3 * i + 2
        faces[i]["normal"] = nrm
        faces[i]["color"] = col
    return faces, verts


def test_faces_without_a_bounded_hit_region_behave_as_in_the_reference(rt3, renderer, oracle):
    """The bounding-sphere filters must never hide a face from the reference's literal test (found by tools/fuzz_filter.py:
    tiny faces far from the origin collapse to a point in f32 and then 'cover' the whole frame)."""
    faces, verts = _odd_faces(rt3)
    w, h = 320, 180
    ref = oracle.render_mode_r(faces.view(oracle.GFACE), verts, oracle.camera_update(w, h), w, h)
    assert len(np.unique(ref)) > 3                        # several of the odd faces are visible
    got = hip_mode_r(rt3, renderer, faces, verts, w, h)
    assert np.array_equal(got, ref), "k_mode_r_mfma"
    os.environ["RT3_NO_MFMA"] = "1"
    try:
        assert np.array_equal(hip_mode_r(rt3, renderer, faces, verts, w, h), ref), "k_mode_r_fast"
    finally:
        del os.environ["RT3_NO_MFMA"]
    renderer.force_plain_mode_r(True)
    try:
        assert np.array_equal(hip_mode_r(rt3, renderer, faces, verts, w, h), ref), "k_mode_r"
    finally:
        renderer.force_plain_mode_r(False)
