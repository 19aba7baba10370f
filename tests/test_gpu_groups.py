"""The multi-level candidate filter of k_trace_mfma_tiled: a row of the matrix filter is the bounding sphere of 8 LEAF groups of 8
primitives each (grouped by a spatial median split); a candidate row's ray is tested in f32 against the 8 leaves' bounds, a surviving leaf expands into
member tests.  It only has to be conservative — the exact tests and the (t, kind, index) key are those of the flat filter — so every frame must equal
the flat filter's, the unfiltered kernel's and the oracle's bit for bit, while the matrix filter evaluates 1 / 64 of the (ray, row) pairs."""
import numpy as np
import pytest

from cases import hip_render, mode_x_cases, oracle_render
from test_gpu_brute import random_soup

pytestmark = pytest.mark.gpu


def three_ways(renderer, case):
    """(grouped, flat, brute) frames and the stats of the first two."""
    grouped = hip_render(renderer, case)
    st_g = renderer.stats()
    renderer.force_flat_filter(True)
    try:
        flat = hip_render(renderer, case, upload=False)
        st_f = renderer.stats()
    finally:
        renderer.force_flat_filter(False)
    renderer.force_brute(True)
    try:
        brute = hip_render(renderer, case, upload=False)
    finally:
        renderer.force_brute(False)
    return grouped, flat, brute, st_g, st_f


@pytest.mark.parametrize("seed,n_faces,n_sph,scale", [(11, 700, 0, 1.0), (12, 0, 900, 1.0), (13, 1301, 707, 1.0), (14, 605, 515, 1e3),
                                                      (15, 41, 530, 1e-2), (16, 9, 513, 1.0), (17, 2049, 0, 1.0)])
def test_grouped_filter_equals_flat_filter_equals_brute_on_random_soups(rt3, renderer, seed, n_faces, n_sph, scale):
    """Random triangle soups (skewed stored normals, degenerate faces: always-candidate members) and random spheres, ragged group counts,
    three coordinate scales.  Random order is the WORST case for the groups' bounds (face groups span the scene): still the same pixels."""
    rng = np.random.default_rng(seed)
    faces, verts, fm, cr, sm = random_soup(rng, n_faces, n_sph, scale, rt3)
    cam = rt3.Camera().update(96, 64, 1.0, 3.0, 2.0)
    case = dict(cam=cam.c, params=dict(width=96, height=64, spp=4, max_depth=6, seed=seed, flags=1, t_min=0.001 * scale))
    if n_faces:
        case.update(faces=faces, verts=verts, fmats=fm)
    if n_sph:
        case.update(spheres=cr, smats=sm)
    grouped, flat, brute, st_g, st_f = three_ways(renderer, case)
    assert np.array_equal(flat, brute), "flat filter != brute"
    assert np.array_equal(grouped, brute), "multi-level filter != brute: %d pixels" % int((grouped != brute).sum())
    assert st_g.ray_casts == st_f.ray_casts
    assert st_f.filter_tests == st_f.prim_tests                           # flat: one row per primitive
    rows_g, rows_f = st_g.filter_tests // st_g.ray_casts, st_f.filter_tests // st_f.ray_casts
    assert rows_g <= rows_f // 64 + 3 and st_g.mfma_instructions < st_f.mfma_instructions
    assert st_g.bound_tests > 0 and st_f.bound_tests == 0          # the leaves' (and the faces' own) bounds, tested in f32
    want, casts = oracle_render(case, threads=16)
    assert np.array_equal(grouped, want) and casts == st_g.ray_casts


def test_coherent_scenes_test_fewer_members_than_the_flat_filter_tests_rows(rt3, renderer):
    """Where the order is coherent (a tessellated mesh; spheres after the median split) the groups are tight: the Cornell-style box of config 5
    (small version) and the 100k-sphere scene of config 4 (small version) run far fewer filter tests and about as many exact tests."""
    f, v, m = rt3.scene_cornell(16)
    cam = rt3.Camera().update(128, 128, 2.0, 2.0, 2.0)
    case = dict(faces=f, verts=v, fmats=m, cam=cam.c, params=dict(width=128, height=128, spp=4, max_depth=8, seed=4, flags=1 | 2))
    grouped, flat, brute, st_g, st_f = three_ways(renderer, case)
    assert np.array_equal(grouped, flat) and np.array_equal(flat, brute)
    assert st_g.filter_tests * 50 < st_f.filter_tests
    assert st_g.exact_tests < 2 * st_f.exact_tests and st_g.bound_tests < st_f.filter_tests // 4
    cr, mats = rt3.scene_stress(6000, 7)
    cam = rt3.Camera().look_at(160, 90, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=160, height=90, spp=4, max_depth=8, seed=2, flags=1))
    grouped, flat, brute, st_g, st_f = three_ways(renderer, case)
    assert np.array_equal(grouped, flat) and np.array_equal(flat, brute)
    assert st_g.filter_tests * 50 < st_f.filter_tests and st_g.exact_tests < st_f.filter_tests // 8


def test_spheres_no_exact_test_can_accept_and_direct_spheres_stay_out_of_the_groups(rt3, renderer):
    """Non-finite sphere records (never hit by the exact test, and a NaN centre would poison a group's bound) and the spheres on the direct
    list (a ground sphere: tested for every ray, member of no group) around 600 ordinary spheres."""
    rng = np.random.default_rng(5)
    n = 600
    cr = np.zeros((n, 4), np.float32)
    cr[:, :3] = rng.uniform(-1.0, 1.0, (n, 3)) * np.float32([6.0, 0.5, 6.0]) + np.float32([0.0, 0.3, -8.0])
    cr[:, 3] = rng.uniform(0.05, 0.3, n)
    cr[0] = (0.0, -1000.0, -8.0, 1000.0)                                  # the ground: direct
    cr[17, 0] = np.nan
    cr[99, 1] = np.inf
    cr[300, 3] = np.float32(3e38)                                         # r^2 overflows
    mats = np.zeros(n, rt3.MATERIAL)
    mats["kind"] = rng.integers(1, 4, n)
    mats["rgb"] = rng.uniform(0.3, 1.0, (n, 3))
    mats["param"] = np.where(mats["kind"] == 3, 1.5, 0.1).astype(np.float32)
    cam = rt3.Camera().look_at(120, 68, (0.0, 2.0, 2.0), (0.0, 0.3, -8.0), (0.0, 1.0, 0.0), 50.0, 1.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=120, height=68, spp=4, max_depth=10, seed=8, flags=1))
    grouped, flat, brute, st_g, st_f = three_ways(renderer, case)
    assert np.array_equal(flat, brute) and np.array_equal(grouped, brute)
    want, casts = oracle_render(case, threads=16)
    assert np.array_equal(grouped, want) and casts == st_g.ray_casts


@pytest.mark.parametrize("name", sorted(mode_x_cases().keys()))
def test_environment_switch_selects_the_flat_filter(renderer, name, monkeypatch):
    case = mode_x_cases()[name]
    want = hip_render(renderer, case)
    st = renderer.stats()
    monkeypatch.setenv("RT3_NO_GROUPS", "1")
    assert np.array_equal(hip_render(renderer, case, upload=False), want)
    st_f = renderer.stats()
    if case.get("faces") is not None:                                     # (the sphere cases of this set run k_trace_mfma32: no rows to group)
        assert st_f.filter_tests == st_f.prim_tests and st.filter_tests < st_f.filter_tests


@pytest.mark.parametrize("seed,n_faces,n_sph", [(21, 900, 0), (22, 0, 1100), (23, 1301, 707)])
def test_tiled_rows_equal_resident_rows(rt3, renderer, seed, n_faces, n_sph, monkeypatch):
    """Scenes whose rows fit in LDS run the barrier-free variant; RT3_NO_RESIDENT=1 streams the same rows through the 64-KiB tiles (the path
    of scenes beyond 112 000 primitives): same frame, same counters."""
    rng = np.random.default_rng(seed)
    faces, verts, fm, cr, sm = random_soup(rng, n_faces, n_sph, 1.0, rt3)
    cam = rt3.Camera().update(96, 64, 1.0, 3.0, 2.0)
    case = dict(cam=cam.c, params=dict(width=96, height=64, spp=4, max_depth=6, seed=seed, flags=1, t_min=0.001))
    if n_faces:
        case.update(faces=faces, verts=verts, fmats=fm)
    if n_sph:
        case.update(spheres=cr, smats=sm)
    resident = hip_render(renderer, case)
    st_r = renderer.stats()
    monkeypatch.setenv("RT3_NO_RESIDENT", "1")
    tiled = hip_render(renderer, case, upload=False)
    st_t = renderer.stats()
    assert np.array_equal(tiled, resident)
    assert (st_t.ray_casts, st_t.prim_tests) == (st_r.ray_casts, st_r.prim_tests)
    renderer.force_brute(True)
    try:
        assert np.array_equal(hip_render(renderer, case, upload=False), resident)
    finally:
        renderer.force_brute(False)


def test_scene_beyond_three_levels_scans_super_rows(rt3, renderer):
    """150 000 spheres + 3 000 faces: 2 344 + 47 rows of 64 are more than LDS holds beside the pair lists, so the filter takes FOUR levels — the matrix
    cores scan super-rows of 512 primitives (294 + 6), a candidate super-row's 8 rows are tested in f32, then leaves, then members.  Against the flat
    filter and the unfiltered kernel on a small frame."""
    cr, mats = rt3.scene_stress(150000, 9)
    rng = np.random.default_rng(31)
    faces, verts, fm, _, _ = random_soup(rng, 3000, 0, 1.0, rt3)
    verts = verts.copy()
    verts[:, :3] = verts[:, :3] * np.float32(3.0) + np.float32([0.0, 6.0, -30.0])
    cam = rt3.Camera().look_at(96, 54, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    case = dict(spheres=cr, smats=mats, faces=faces, verts=verts, fmats=fm, cam=cam.c,
                params=dict(width=96, height=54, spp=2, max_depth=8, seed=3, flags=1))
    grouped, flat, brute, st_g, st_f = three_ways(renderer, case)
    assert np.array_equal(flat, brute) and np.array_equal(grouped, brute)
    rows = st_g.filter_tests // st_g.ray_casts
    assert rows == -(-2344 // 8) + -(-47 // 8) and st_g.filter_tests * 400 < st_f.filter_tests


@pytest.mark.parametrize("levels,tiles", [("3", False), ("4", False), ("3", True), ("4", True)])
@pytest.mark.parametrize("seed,n_faces,n_sph", [(41, 1100, 0), (42, 0, 1300), (43, 900, 800)])
def test_three_and_four_levels_resident_and_tiled_give_the_same_frame(rt3, renderer, seed, n_faces, n_sph, levels, tiles, monkeypatch):
    """k_trace_levels in all four forms on the same small soups (RT3_LEVELS forces the level count, RT3_NO_RESIDENT the tiles): the frame of the
    unfiltered kernel every time, the same ray casts; with four levels the matrix cores scan an eighth of the rows."""
    rng = np.random.default_rng(seed)
    faces, verts, fm, cr, sm = random_soup(rng, n_faces, n_sph, 1.0, rt3)
    cam = rt3.Camera().update(96, 64, 1.0, 3.0, 2.0)
    case = dict(cam=cam.c, params=dict(width=96, height=64, spp=4, max_depth=6, seed=seed, flags=1, t_min=0.001))
    if n_faces:
        case.update(faces=faces, verts=verts, fmats=fm)
    if n_sph:
        case.update(spheres=cr, smats=sm)
    want = hip_render(renderer, case)
    st_d = renderer.stats()
    monkeypatch.setenv("RT3_LEVELS", levels)
    if tiles:
        monkeypatch.setenv("RT3_NO_RESIDENT", "1")
    got = hip_render(renderer, case, upload=False)
    st = renderer.stats()
    assert np.array_equal(got, want) and st.ray_casts == st_d.ray_casts
    rows64 = -(-n_faces // 64) + -(-n_sph // 64)
    per_cast = st.filter_tests // st.ray_casts
    assert per_cast <= (rows64 if levels == "3" else -(-rows64 // 8)) + 4
    monkeypatch.delenv("RT3_LEVELS")
    monkeypatch.delenv("RT3_NO_RESIDENT", raising=False)
    renderer.force_brute(True)
    try:
        assert np.array_equal(hip_render(renderer, case, upload=False), want)
    finally:
        renderer.force_brute(False)


def test_nested_form_of_round_three_still_agrees(rt3, renderer, monkeypatch):
    """RT3_OLD_GROUPS=1: k_trace_mfma_tiled's nested three-level form (the A/B reference of k_trace_levels)."""
    rng = np.random.default_rng(51)
    faces, verts, fm, cr, sm = random_soup(rng, 1000, 900, 1.0, rt3)
    cam = rt3.Camera().update(96, 64, 1.0, 3.0, 2.0)
    case = dict(faces=faces, verts=verts, fmats=fm, spheres=cr, smats=sm, cam=cam.c, params=dict(width=96, height=64, spp=4, max_depth=6, seed=5, flags=1, t_min=0.001))
    want = hip_render(renderer, case)
    monkeypatch.setenv("RT3_OLD_GROUPS", "1")
    assert np.array_equal(hip_render(renderer, case, upload=False), want)


@pytest.mark.parametrize("levels", ["3", "4", None])
def test_scene_whose_spheres_are_all_direct_after_a_grouped_scene(rt3, renderer, levels, monkeypatch):
    """A mesh with ONE sphere that every ray meets (the direct list takes it: no sphere rows at all), rendered on a context that held a large sphere
    scene before — the row counts of the previous scene must not survive (regression: k_trace_levels read freed super-row fragments)."""
    cr, mats = rt3.scene_stress(3000, 5)
    cam = rt3.Camera().update(96, 64, 1.0, 3.0, 2.0)
    renderer.set_mesh(np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32))
    renderer.set_spheres(cr, mats)
    renderer.render_path(cam.c, rt3.make_params(96, 64, spp=1, max_depth=2, flags=1))
    rng = np.random.default_rng(61)
    faces, verts, fm, _, _ = random_soup(rng, 900, 0, 1.0, rt3)
    one = np.float32([[0.0, -1000.0, -4.0, 999.0]])
    sm = np.zeros(1, rt3.MATERIAL)
    sm["kind"], sm["rgb"] = 1, (0.5, 0.5, 0.5)
    case = dict(faces=faces, verts=verts, fmats=fm, spheres=one, smats=sm, cam=cam.c, params=dict(width=96, height=64, spp=2, max_depth=5, seed=6, flags=1, t_min=0.001))
    if levels:
        monkeypatch.setenv("RT3_LEVELS", levels)
    got = hip_render(renderer, case)
    monkeypatch.delenv("RT3_LEVELS", raising=False)
    renderer.force_brute(True)
    try:
        assert np.array_equal(hip_render(renderer, case, upload=False), got)
    finally:
        renderer.force_brute(False)


@pytest.mark.parametrize("n", [112600, 112700, 114700])
def test_sphere_counts_around_the_resident_limit_launch(rt3, renderer, n):
    """55 row blocks of 32 rows of 64 spheres = 112 640 spheres is the most the resident three-level form takes (one block of headroom below 160 KiB of
    LDS); the next sphere moves the scene to k_trace_levels.  Both sides of the limit must launch and agree with the flat filter."""
    cr, mats = rt3.scene_stress(n, 3)
    cam = rt3.Camera().look_at(64, 36, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=64, height=36, spp=1, max_depth=4, seed=3, flags=1))
    grouped = hip_render(renderer, case)
    st = renderer.stats()
    renderer.force_flat_filter(True)
    try:
        assert np.array_equal(hip_render(renderer, case, upload=False), grouped)
    finally:
        renderer.force_flat_filter(False)
    rows = st.filter_tests // st.ray_casts
    assert rows == (-(-n // 64) if n <= 112640 else -(-(-(-n // 64)) // 8))
