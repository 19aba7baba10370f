"""The matrix-filter kernels against the VALU-scan kernels of the same library, several times over, on ~10^7 samples per
render: the two share the exact tests and the shading but not the candidate search, so any difference — or any difference
between two runs — is a lost or invented candidate.  This is the guard for the hardware hazard documented at pk_bf16() in
rt3_device.hip (a conversion result read too early dropped about one candidate per 10^6 rays, differently in every run)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render_both(renderer, render, runs):
    os.environ["RT3_NO_MFMA"] = "1"
    try:
        ref = render().copy()
    finally:
        del os.environ["RT3_NO_MFMA"]
    for k in range(runs):
        img = render()
        bad = int((img != ref).sum())
        assert bad == 0, "run %d: %d of %d pixels differ from the VALU-scan kernel" % (k, bad, ref.size)


def test_sphere_scene_matrix_filter_equals_valu_scan_every_time(rt3, renderer):
    cr, mats = rt3.scene_weekend(42)
    W, H = 960, 540
    cam = rt3.weekend_camera(W, H)
    renderer.set_spheres(cr, mats)
    p = rt3.make_params(W, H, spp=32, max_depth=50, seed=3, flags=1, lens_radius=0.05)
    _render_both(renderer, lambda: renderer.render_path(cam.c, p), runs=5)


def test_tiled_scene_matrix_filter_equals_valu_scan_every_time(rt3, renderer):
    cr, mats = rt3.scene_stress(3000, 43)
    W, H = 480, 270
    cam = rt3.weekend_camera(W, H)
    renderer.set_spheres(cr, mats)
    p = rt3.make_params(W, H, spp=8, max_depth=20, seed=5, flags=1)
    _render_both(renderer, lambda: renderer.render_path(cam.c, p), runs=3)


def test_differential_fuzz_of_the_two_candidate_searches(renderer):
    """A slice of tools/fuzz_filter.py (the campaign behind DESIGN.md §5.1: 54 000 scenes): random sphere scenes and triangle soups —
    far from the origin, five decades of sizes, degenerate faces, skewed stored normals — rendered by the matrix-filter kernels
    and by the VALU-scan kernels (Mode X) and by the matrix filter and the plain brute force (Mode R)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_filter", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_filter.py"))
    F = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(F)
    messages = []
    bad = F.run(240, 500000, r=renderer, log=messages.append)
    assert bad == 0, "\n".join(m for m in messages if m.startswith("seed"))
