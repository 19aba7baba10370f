"""Randomised parity: small random scenes (spheres of every material, overlapping and nested spheres, triangle soups, a ground
quad, random look-at cameras, lens, strata, tiles) rendered by the HIP path and by the oracle must be bit-identical.  Seeds are
fixed; the point is to reach rare branches (total internal reflection chains, absorbed metal rays, near-zero Lambert sums,
queue overflow, ragged tails of the 4- and 32-sphere scan blocks, faces whose bounding spheres overlap many rays)."""
import os

import numpy as np
import pytest

from cases import hip_render, oracle_render, rt3

pytestmark = pytest.mark.gpu


def random_case(seed):
    rng = np.random.RandomState(seed)
    w, h = int(rng.randint(17, 72)), int(rng.randint(9, 48))
    case = {}
    n_sph = int(rng.choice([0, 1, 2, 3, 5, 31, 32, 33, 36, 67, 130]))
    if n_sph:
        cr = np.zeros((n_sph, 4), np.float32)
        cr[:, :3] = rng.uniform(-4, 4, (n_sph, 3)) * np.float32([1.0, 0.5, 1.0]) + np.float32([0, 0, -6])
        cr[:, 3] = rng.uniform(0.1, 1.2, n_sph)
        if rng.rand() < 0.5:
            cr[0] = (0.0, -1000.5, -6.0, 1000.0)                     # huge ground sphere
        if n_sph > 2 and rng.rand() < 0.5:
            cr[2, :3] = cr[1, :3]                                    # nested / concentric pair (hollow glass)
            cr[2, 3] = cr[1, 3] * 0.7
        mats = np.zeros(n_sph, rt3.MATERIAL)
        mats["kind"] = rng.randint(0, 4, n_sph)
        mats["rgb"] = rng.uniform(0.05, 1.0, (n_sph, 3))
        mats["param"] = np.where(mats["kind"] == 3, rng.choice([1.5, 1.0 / 1.5, 2.4], n_sph), rng.uniform(0.0, 1.0, n_sph) * (rng.rand(n_sph) < 0.7))
        case.update(spheres=cr, smats=mats)
    n_tri = int(rng.choice([0, 0, 1, 4, 37, 300])) if n_sph else int(rng.choice([1, 5, 64, 257]))
    if n_tri:
        parts, fm = [], []
        for i in range(n_tri):
            c = rng.uniform(-3, 3, 3) * np.float32([1, 0.6, 1]) + np.float32([0, 0, -6])
            p = [tuple(np.float32(c + rng.uniform(-0.8, 0.8, 3))) for _ in range(3)]
            if i == 0:                                               # a big ground quad half
                p = [(-8.0, -1.5, -1.0), (8.0, -1.5, -1.0), (0.0, -1.5, -14.0)]
            if i == 1 and rng.rand() < 0.3:
                p[2] = p[1]                                          # degenerate face (NaN normal)
            e = rt3.create_triangle(*p, tuple(rng.uniform(0, 1, 3)))
            parts.append(rt3.pre_render_entity(e))
            m = np.zeros(1, rt3.MATERIAL)
            m["kind"] = rng.randint(0, 4)
            m["rgb"] = rng.uniform(0.1, 1.0, 3) * (4.0 if m["kind"][0] == 0 and rng.rand() < 0.3 else 1.0)
            m["param"] = 1.5 if m["kind"][0] == 3 else rng.uniform(0, 0.6)
            fm.append(m)
        faces, verts = rt3.merge_entities(parts)
        case.update(faces=faces, verts=verts, fmats=np.concatenate(fm) if rng.rand() < 0.8 else None)
    cam = rt3.Camera().look_at(w, h, tuple(rng.uniform(-2, 2, 3) + np.float32([0, 0.5, 1.5])), (0.0, 0.0, -6.0), (0.0, 1.0, 0.0),
                               float(rng.uniform(25, 70)), float(rng.uniform(4, 8)))
    spp = int(rng.choice([1, 2, 3, 4, 9]))
    tiles = int(rng.choice([1, 1, 2, 3]))
    case.update(cam=cam.c, params=dict(width=w, height=h, spp=spp, max_depth=int(rng.choice([1, 2, 5, 12, 50])), seed=int(rng.randint(1, 1 << 30)),
                                       flags=int(rng.randint(0, 4)), lens_radius=float(rng.choice([0.0, 0.0, 0.08])),
                                       t_min=float(rng.choice([0.001, 0.001, 0.0, 0.01])), tile_rows=int(rng.choice([1, 3, 8])),
                                       tile_index=int(rng.randint(0, tiles)), tile_count=tiles))
    return case


@pytest.mark.parametrize("seed", range(40))
def test_random_scene(renderer, seed):
    case = random_case(seed)
    want, casts = oracle_render(case, threads=16)
    got = hip_render(renderer, case)
    bad = got != want
    assert not bad.any(), "seed %d: %d of %d pixels differ (params %r)" % (seed, bad.sum(), bad.size, case["params"])
    assert renderer.stats().ray_casts == casts


def filter_stress_case(seed):
    """Sphere-only scenes of 100-1100 spheres (both sides of the 512-sphere limit of the all-in-LDS kernel) placed far from the origin and at very different scales: the matrix-core
    candidate filter works on the expanded form |C|^2 - 2 C.o + |o|^2, whose cancellation error grows with the square of the
    coordinates; its margin must grow with it, and the exact test must still see every true hit."""
    rng = np.random.RandomState(1000 + seed)
    scale = float(rng.choice([0.01, 1.0, 1.0, 50.0, 2000.0]))
    offset = rng.choice([0.0, 0.0, 300.0, 5000.0, 60000.0]) * rng.uniform(-1, 1, 3) * scale
    n = int(rng.choice([100, 257, 400, 484, 512, 513, 1100]))
    cr = np.zeros((n, 4), np.float64)
    cr[:, :3] = rng.uniform(-6, 6, (n, 3)) * np.float64([1.0, 0.4, 1.0]) + np.float64([0, 0, -9])
    cr[:, 3] = rng.uniform(0.05, 0.9, n) * rng.choice([1.0, 1.0, 0.05], n)
    if rng.rand() < 0.6:
        cr[0] = (0.0, -1000.6, -9.0, 1000.0)
    cr[:, :3] = cr[:, :3] * scale + offset
    cr[:, 3] *= scale
    mats = np.zeros(n, rt3.MATERIAL)
    mats["kind"] = rng.randint(0, 4, n)
    mats["rgb"] = rng.uniform(0.2, 1.0, (n, 3))
    mats["param"] = np.where(mats["kind"] == 3, 1.5, rng.uniform(0.0, 0.5, n))
    w, h = 64, 36
    eye = np.float64([rng.uniform(-1, 1), rng.uniform(0, 1.5), 2.0]) * scale + offset
    at = np.float64([0.0, 0.0, -9.0]) * scale + offset
    cam = rt3.Camera().look_at(w, h, tuple(eye), tuple(at), (0.0, 1.0, 0.0), 50.0, 1.0)
    return dict(spheres=cr.astype(np.float32), smats=mats, cam=cam.c,
                params=dict(width=w, height=h, spp=4, max_depth=16, seed=seed + 1, flags=1, t_min=float(0.001 * scale)))


@pytest.mark.parametrize("seed", range(24))
def test_matrix_filter_is_conservative_far_from_the_origin(renderer, seed):
    case = filter_stress_case(seed)
    want, casts = oracle_render(case, threads=16)
    got = hip_render(renderer, case)
    bad = got != want
    assert not bad.any(), "seed %d: %d of %d pixels differ" % (seed, bad.sum(), bad.size)
    st = renderer.stats()
    assert st.ray_casts == casts and st.mfma_instructions > 0          # the matrix filter really ran


def _both_kernels_against_the_oracle(renderer, case, what):
    want, casts = oracle_render(case, threads=16)
    got = hip_render(renderer, case)
    assert not (got != want).any(), "%s, matrix filter: %d pixels differ" % (what, (got != want).sum())
    assert renderer.stats().ray_casts == casts
    os.environ["RT3_NO_MFMA"] = "1"
    try:
        got = hip_render(renderer, case, upload=False)
    finally:
        del os.environ["RT3_NO_MFMA"]
    assert not (got != want).any(), "%s, VALU scan: %d pixels differ" % (what, (got != want).sum())


@pytest.mark.parametrize("seed", [2067, 2607, 3167, 2875, 2251])
def test_faces_far_from_the_origin_collapse_to_points_and_still_match(renderer, seed):
    """Scenes tools/fuzz_filter.py found: triangle soups 10^4 scene sizes away from the origin, where the smallest faces collapse
    to a point (or a line) in f32 — the reference's edge tests then accept the whole plane — and every bounding-sphere filter
    that trusted the geometry lost those hits.  Both GPU kernels against the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_filter", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_filter.py"))
    F = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(F)
    cr, mats, cam, p, info = F.scene(seed)
    spread = float(np.abs(cr[:, :3] - np.float32(info["offset"])).mean() / max(info["scale"], 1e-30))
    faces, verts, fm = F.mesh(seed, info["scale"], np.float64(info["offset"]), max(spread, 1.0))
    case = dict(faces=faces, verts=verts, fmats=fm, cam=cam.c,
                params=dict(width=p.width, height=p.height, spp=p.spp, max_depth=p.max_depth, seed=p.seed, flags=p.flags,
                            lens_radius=p.lens_radius, t_min=p.t_min))
    if seed % 4 != 3:
        case.update(spheres=cr, smats=mats)
    _both_kernels_against_the_oracle(renderer, case, "fuzz seed %d" % seed)


def test_mode_x_faces_without_a_bounded_hit_region(renderer):
    """The hand-made odd faces of test_gpu_mode_r.py (coincident vertices with a valid normal, collinear vertices, a normal that
    is not perpendicular to its triangle) with materials, next to spheres, in Mode X."""
    from test_gpu_mode_r import _odd_faces
    faces, verts = _odd_faces(rt3)
    fm = np.zeros(len(faces), rt3.MATERIAL)
    fm["kind"] = [1, 1, 2, 0, 3]
    fm["rgb"] = faces["color"]
    fm["param"] = [0.0, 0.0, 0.2, 0.0, 1.5]
    cr = np.float32([[0.0, -0.3, -2.0, 0.4], [1.2, 0.6, -3.0, 0.5]])
    sm = np.zeros(2, rt3.MATERIAL)
    sm["kind"] = [3, 2]
    sm["rgb"] = [[1, 1, 1], [0.8, 0.7, 0.3]]
    sm["param"] = [1.5, 0.1]
    cam = rt3.main_camera(160, 90)
    case = dict(faces=faces, verts=verts, fmats=fm, spheres=cr, smats=sm, cam=cam.c,
                params=dict(width=160, height=90, spp=4, max_depth=12, seed=5, flags=1))
    _both_kernels_against_the_oracle(renderer, case, "odd faces")
