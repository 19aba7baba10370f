"""k_trace_brute — Mode X with no candidate filter at all (every ray against every primitive in index order) — is the on-GPU
arbiter of the filtered kernels: it must agree with the oracle, and the matrix-core and vector-ALU filters must agree with it."""
import os

import numpy as np
import pytest

from cases import GOLDEN, hip_render, mode_x_cases, oracle_render

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(mode_x_cases().keys()))
def test_brute_equals_the_goldens(renderer, name):
    z = np.load(os.path.join(GOLDEN, "mode_x_small.npz"))
    renderer.force_brute(True)
    try:
        got = hip_render(renderer, mode_x_cases()[name])
        st = renderer.stats()
    finally:
        renderer.force_brute(False)
    assert np.array_equal(got, z[name]) and st.mfma_instructions == 0 and st.exact_tests == 0


def test_brute_is_selected_by_the_environment_too(renderer):
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    want = hip_render(renderer, case)
    assert renderer.stats().mfma_instructions > 0
    os.environ["RT3_BRUTE"] = "1"
    try:
        assert np.array_equal(hip_render(renderer, case, upload=False), want)
        assert renderer.stats().mfma_instructions == 0
    finally:
        del os.environ["RT3_BRUTE"]


def random_soup(rng, n_faces, n_sph, scale, rt3):
    """Random triangles (stored normals sometimes skewed, some faces degenerate) and spheres around the view axis."""
    c = rng.uniform(-1.0, 1.0, (n_faces, 1, 3)) * np.float32([2.0, 1.5, 2.0]) + np.float32([0.0, 0.0, -4.0])
    v = (c + rng.normal(0.0, 0.35, (n_faces, 3, 3))).astype(np.float32) * np.float32(scale)
    if n_faces >= 8:
        v[3, 1] = v[3, 0]                                   # two coincident vertices
        v[5, 2] = v[5, 0] + 2 * (v[5, 1] - v[5, 0])         # collinear
        v[7, :] = v[7, 0]                                   # a point
    faces = np.zeros(n_faces, rt3.GFACE)
    verts = np.zeros((3 * n_faces, 4), np.float32)
    verts[:, :3] = v.reshape(-1, 3)
    faces["v1"], faces["v2"], faces["v3"] = 3 * np.arange(n_faces), 3 * np.arange(n_faces) + 1, 3 * np.arange(n_faces) + 2
    with np.errstate(invalid="ignore", divide="ignore"):
        n = np.cross(v[:, 2] - v[:, 0], v[:, 1] - v[:, 0])
        n = n / np.linalg.norm(n, axis=1, keepdims=True)
    n = np.nan_to_num(n).astype(np.float32)
    skew = rng.random(n_faces) < 0.2
    n[skew] += rng.normal(0.0, 0.3, (int(skew.sum()), 3)).astype(np.float32)
    faces["normal"] = n
    fm = np.zeros(n_faces, rt3.MATERIAL)
    fm["kind"] = rng.integers(0, 4, n_faces)
    fm["rgb"] = rng.uniform(0.2, 1.0, (n_faces, 3))
    fm["param"] = np.where(fm["kind"] == 3, 1.5, rng.uniform(0.0, 0.5, n_faces)).astype(np.float32)
    cr = np.zeros((n_sph, 4), np.float32)
    cr[:, :3] = (rng.uniform(-1.0, 1.0, (n_sph, 3)) * np.float32([2.5, 1.5, 2.5]) + np.float32([0.0, 0.0, -4.5])) * np.float32(scale)
    cr[:, 3] = rng.uniform(0.05, 0.5, n_sph) * scale
    sm = np.zeros(n_sph, rt3.MATERIAL)
    sm["kind"] = rng.integers(0, 4, n_sph)
    sm["rgb"] = rng.uniform(0.2, 1.0, (n_sph, 3))
    sm["param"] = np.where(sm["kind"] == 3, 1.5, rng.uniform(0.0, 0.5, n_sph)).astype(np.float32)
    return faces, verts, fm, cr, sm


@pytest.mark.parametrize("seed,n_faces,n_sph,scale", [(1, 700, 0, 1.0), (2, 0, 900, 1.0), (3, 1300, 700, 1.0), (4, 600, 300, 1e3),
                                                      (5, 40, 30, 1e-2)])
def test_filters_agree_with_brute_and_the_oracle_on_random_soups(rt3, renderer, seed, n_faces, n_sph, scale):
    rng = np.random.default_rng(seed)
    faces, verts, fm, cr, sm = random_soup(rng, n_faces, n_sph, scale, rt3)
    cam = rt3.Camera().update(96, 64, 1.0, 3.0, 2.0)
    case = dict(cam=cam.c, params=dict(width=96, height=64, spp=4, max_depth=6, seed=seed, flags=1, t_min=0.001 * scale))
    if n_faces:
        case.update(faces=faces, verts=verts, fmats=fm)
    if n_sph:
        case.update(spheres=cr, smats=sm)
    want, casts = oracle_render(case, threads=16)
    renderer.force_brute(True)
    try:
        brute = hip_render(renderer, case)
        assert renderer.stats().ray_casts == casts
    finally:
        renderer.force_brute(False)
    assert np.array_equal(brute, want)
    mfma = hip_render(renderer, case, upload=False)
    st = renderer.stats()
    assert np.array_equal(mfma, brute) and st.mfma_instructions > 0
    if n_faces or n_sph > 512:
        assert 0 < st.exact_tests < st.prim_tests             # the tiled kernels count the survivors of the filter
    os.environ["RT3_NO_MFMA"] = "1"
    try:
        assert np.array_equal(hip_render(renderer, case, upload=False), brute)
    finally:
        del os.environ["RT3_NO_MFMA"]


def test_the_k64_form_of_the_small_scene_kernel_equals_the_default(renderer):
    """k_trace_mfma (round 1's K = 64 filter on v_mfma_f32_32x32x16_bf16, RT3_MFMA_K64=1) is kept as the A/B reference of k_trace_mfma32."""
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    want = hip_render(renderer, case)
    st = renderer.stats()
    assert st.mfma_flop_per_instruction == 16384             # (all three spheres of this case are on the direct list: no filter candidates)
    os.environ["RT3_MFMA_K64"] = "1"
    try:
        assert np.array_equal(hip_render(renderer, case, upload=False), want)
        assert renderer.stats().mfma_flop_per_instruction == 32768
    finally:
        del os.environ["RT3_MFMA_K64"]
