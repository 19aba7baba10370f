"""Spheres that nearly every ray is a candidate for are taken out of the matrix filter and tested directly (TraceArgs::direct,
sphere_direct_list in rt3_device.hip).  The nearest hit must not depend on which path found it: exact ties in t between a
direct sphere and a filtered one go to the lower index, as in the sequential loops.  Six coincident big spheres with different flat
colours (four fit on the list, two stay in the filter), a sphere around the camera, small spheres in front — every matrix-filter kernel
against the unfiltered one."""
import os

import numpy as np
import pytest

from cases import hip_render

pytestmark = pytest.mark.gpu


def scene(rt3, n_small, order):
    rng = np.random.default_rng(7)
    big = [(0.0, -1000.0, -6.0, 1000.0)] * 3 + [(0.0, 0.0, -6.0, 60.0)] + [(0.0, -1000.0, -6.0, 1000.0)] * 3   # coincident grounds + an enclosing sphere
    small = np.zeros((n_small, 4), np.float32)
    small[:, :3] = rng.uniform(-1.0, 1.0, (n_small, 3)) * np.float32([5.0, 1.5, 5.0]) + np.float32([0.0, 1.6, -6.0])
    small[:, 3] = rng.uniform(0.05, 0.3, n_small)
    cr = np.concatenate([np.float32(big), small]) if order == "big first" else np.concatenate([small, np.float32(big)])
    mats = np.zeros(len(cr), rt3.MATERIAL)
    mats["kind"] = rt3.MAT_FLAT
    k = np.arange(len(cr))
    mats["rgb"] = np.stack([(k * 37 % 251 + 4) / 255.0, (k * 101 % 241 + 8) / 255.0, (k * 59 % 239 + 12) / 255.0], axis=1)
    glass = rng.random(len(cr)) < 0.3                          # some spheres scatter, so that rays start on and inside spheres too
    mats["kind"][glass] = rt3.MAT_DIELECTRIC
    mats["param"][glass] = 1.5
    cam = rt3.Camera().look_at(96, 64, (0.0, 2.0, 3.0), (0.0, 1.0, -6.0), (0.0, 1.0, 0.0), 50.0, 1.0)
    return dict(cam=cam.c, spheres=cr, smats=mats, params=dict(width=96, height=64, spp=4, max_depth=8, seed=11, flags=1, t_min=0.001))


@pytest.mark.parametrize("order", ["big first", "big last"])
@pytest.mark.parametrize("n_small", [200, 900])             # k_trace_mfma32 | the sphere pass of the tiled kernel
def test_direct_and_filtered_spheres_agree_with_the_unfiltered_kernel(rt3, renderer, n_small, order):
    case = scene(rt3, n_small, order)
    renderer.force_brute(True)
    try:
        want = hip_render(renderer, case)
    finally:
        renderer.force_brute(False)
    got = hip_render(renderer, case, upload=False)
    st = renderer.stats()
    assert st.mfma_instructions > 0 and st.exact_tests > 0
    assert np.array_equal(got, want)
    for knob in ("RT3_MFMA_K64", "RT3_FORCE_TILED", "RT3_NO_MFMA"):
        os.environ[knob] = "1"
        try:
            assert np.array_equal(hip_render(renderer, case, upload=False), want), knob
        finally:
            del os.environ[knob]
