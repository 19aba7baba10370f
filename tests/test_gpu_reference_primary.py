"""RT3_FLAG_REFERENCE_PRIMARY on the device: Mode X reduces to Mode R byte for byte (consequence 1(i); T7),
so rt3_render_path reproduces the PPM hashes recorded from the reference itself — the thread that ties the Mode-X code (face test,
sky, pack, reduce, sample -> pixel mapping) to the pinned regime."""
import json
import os

import numpy as np
import pytest

from cases import GOLDEN, hip_render, load_builtin_scene, oracle_render

pytestmark = pytest.mark.gpu
PINS = json.load(open(os.path.join(GOLDEN, "reference_pins.json")))


def builtin_case(rt3, w, h, **params):
    faces, verts = load_builtin_scene()
    cam = rt3.main_camera(w, h)
    base = dict(width=w, height=h, spp=1, max_depth=1, seed=1, flags=rt3.FLAG_REFERENCE_PRIMARY, t_min=0.0)
    base.update(params)
    return dict(faces=faces.view(rt3.GFACE), verts=verts, fmats=None, cam=cam.c, params=base)


def ppm_hash(rt3, oracle, img):
    h, w = img.shape
    ref_like = img.copy()
    ref_like[h - 1] = 0                                    # the reference's CPU loop never writes row H-1 (SequentialRenderer.cpp:286)
    f = rt3.Frame(w, h)
    f.data[:] = ref_like
    return oracle.sha256(f.ppm_bytes())


@pytest.mark.parametrize("size", ["400x225", "1920x1080"])
def test_mode_x_with_the_flag_reproduces_the_reference_ppm(rt3, renderer, oracle, size):
    w, h = map(int, size.split("x"))
    case = builtin_case(rt3, w, h)
    img = hip_render(renderer, case)                        # rt3_render_path -> k_trace_mfma_tiled<faces, REF>
    assert renderer.stats().ray_casts == w * h
    assert ppm_hash(rt3, oracle, img) == PINS["ppm_sha256"][size]
    # the Mode-R entry point gives the same frame, row H-1 included
    renderer.configure(spp=None)
    cam = rt3.main_camera(w, h)
    renderer.render(cam)
    assert np.array_equal(img, cam.get_frame().d())


def test_every_kernel_family_agrees_under_the_flag(rt3, renderer, oracle):
    case = builtin_case(rt3, 400, 225)
    want = PINS["ppm_sha256"]["400x225"]
    assert ppm_hash(rt3, oracle, hip_render(renderer, case)) == want
    os.environ["RT3_NO_MFMA"] = "1"                         # vector-ALU scan, k_trace<faces, REF>
    try:
        assert ppm_hash(rt3, oracle, hip_render(renderer, case, upload=False)) == want
    finally:
        del os.environ["RT3_NO_MFMA"]
    renderer.force_brute(True)                              # no filter at all, k_trace_brute<REF>
    try:
        assert ppm_hash(rt3, oracle, hip_render(renderer, case, upload=False)) == want
    finally:
        renderer.force_brute(False)
    # sharded: rows of a 3-way split reassemble to the same frame
    plist = [rt3.make_params(**dict(case["params"], tile_rows=4, tile_index=i, tile_count=3)) for i in range(3)]
    tiles = [hip_render(renderer, case, upload=False, tile_rows=4, tile_index=i, tile_count=3) for i in range(3)]
    assert ppm_hash(rt3, oracle, rt3.deinterleave(tiles, plist, 225, 400)) == want


def test_flag_with_materials_depth_and_samples_equals_the_oracle(rt3, renderer):
    """Beyond the Mode-R corner the flag only changes ray cast 0; everything after it is ordinary Mode X."""
    faces, verts, fmats = rt3.scene_cornell(6)
    cam = rt3.Camera().update(80, 80, 2.0, 2.0, 2.0)
    case = dict(faces=faces, verts=verts, fmats=fmats, cam=cam.c,
                params=dict(width=80, height=80, spp=9, max_depth=8, seed=5, flags=1 | 2 | rt3.FLAG_REFERENCE_PRIMARY, t_min=0.001))
    want, casts = oracle_render(case, threads=16)
    got = hip_render(renderer, case)
    assert np.array_equal(got, want) and renderer.stats().ray_casts == casts


def test_camera_off_the_origin_or_with_a_lens_takes_the_unfiltered_kernel(rt3, renderer):
    """With the reference's '+' (SequentialRenderer.cpp:70) and an origin != 0 the "hit point" leaves the face's plane and no bound
    holds: such renders go to k_trace_brute and still equal the oracle's literal evaluation."""
    faces, verts, fmats = rt3.scene_cornell(3)
    cam = rt3.Camera().look_at(64, 48, (0.3, 0.2, 1.5), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 60.0, 1.0)
    for lens in (0.0, 0.05):
        case = dict(faces=faces, verts=verts, fmats=fmats, cam=cam.c,
                    params=dict(width=64, height=48, spp=4, max_depth=5, seed=2, flags=1 | rt3.FLAG_REFERENCE_PRIMARY, t_min=0.0,
                                lens_radius=lens))
        want, casts = oracle_render(case, threads=16)
        assert np.array_equal(hip_render(renderer, case), want)
        st = renderer.stats()
        assert st.ray_casts == casts and st.mfma_instructions == 0


def test_flag_needs_a_triangle_only_scene(rt3, renderer):
    cr, mats = rt3.scene_three_spheres()
    cam = rt3.Camera().update(32, 18, 1.0, 3.5, 2.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=32, height=18, spp=1, max_depth=1, flags=rt3.FLAG_REFERENCE_PRIMARY))
    with pytest.raises(rt3.Fatal, match="triangle-only"):
        hip_render(renderer, case)
    with pytest.raises(rt3.Fatal, match="t_min"):
        hip_render(renderer, case, flags=0, t_min=-1.0)
    with pytest.raises(rt3.Fatal, match="flags"):
        hip_render(renderer, case, flags=1 << 9)
