"""The matrix filter's margin is eps (|C|^2 + r^2 + |o|^2), so it works in coordinates about a centre of the scene:
the median of the spheres' centres, the centre of the mesh's vertex box.  A scene moved far from the world origin must (a) still equal
the unfiltered kernel pixel for pixel and (b) still be FILTERED — without the centre every pair becomes a candidate there."""
import numpy as np
import pytest

from cases import hip_render

pytestmark = pytest.mark.gpu

SHIFTS = [(0.0, 0.0, 0.0), (3000.0, -2000.0, 5000.0)]


def render_both(renderer, case):
    got = hip_render(renderer, case)
    st = renderer.stats()
    assert st.mfma_instructions > 0
    per_cast = st.exact_tests / max(1, st.ray_casts)
    renderer.force_brute(True)
    try:
        want = hip_render(renderer, case, upload=False)
    finally:
        renderer.force_brute(False)
    assert np.array_equal(got, want)
    return per_cast


def test_faces_far_from_the_origin_stay_filtered(rt3, renderer):
    faces, verts, fm = rt3.scene_cornell(16)                # 3000 faces: box x, y in [-1, 1], z in [-4, -2]
    per_cast = []
    for dx, dy, dz in SHIFTS:
        v = verts.copy()
        v[:, :3] += np.float32([dx, dy, dz])
        cam = rt3.Camera().look_at(64, 64, (dx, dy, dz), (dx, dy, dz - 3.0), (0.0, 1.0, 0.0), 53.0, 1.0)
        case = dict(cam=cam.c, faces=faces, verts=v, fmats=fm,
                    params=dict(width=64, height=64, spp=4, max_depth=6, seed=3, flags=3, t_min=0.001))
        per_cast.append(render_both(renderer, case))
    # at 6000 units from the origin f32 resolves 5e-4, the bounds' own inflation (1e-5 (1 + max |coordinate|)) grows with it: some more
    # candidates are expected, every face of the scene (3000 per ray cast, what an uncentred filter yields there) is not
    assert 0 < per_cast[0] < 60 and per_cast[1] < 6 * per_cast[0] + 20, per_cast


@pytest.mark.parametrize("n", [300, 2000])                  # k_trace_mfma32 (all in LDS) | the sphere pass of the tiled kernel
def test_spheres_far_from_the_origin_stay_filtered(rt3, renderer, n):
    rng = np.random.default_rng(n)
    base = np.zeros((n, 4), np.float32)
    base[:, :3] = rng.uniform(-1.0, 1.0, (n, 3)) * np.float32([6.0, 2.0, 6.0]) + np.float32([0.0, 0.0, -8.0])
    base[:, 3] = rng.uniform(0.05, 0.3, n)
    base[0] = (0.0, -1003.0, -8.0, 1000.0)                  # a ground sphere: it must not drag the centre away from the others
    mats = np.zeros(n, rt3.MATERIAL)
    mats["kind"] = rng.integers(1, 4, n)
    mats["rgb"] = rng.uniform(0.2, 1.0, (n, 3))
    mats["param"] = np.where(mats["kind"] == 3, 1.5, rng.uniform(0.0, 0.4, n)).astype(np.float32)
    per_cast = []
    for dx, dy, dz in SHIFTS:
        cr = base.copy()
        cr[:, :3] += np.float32([dx, dy, dz])
        cam = rt3.Camera().look_at(64, 48, (dx, dy + 1.0, dz + 2.0), (dx, dy, dz - 8.0), (0.0, 1.0, 0.0), 50.0, 1.0)
        case = dict(cam=cam.c, spheres=cr, smats=mats, params=dict(width=64, height=48, spp=4, max_depth=8, seed=5, flags=1, t_min=0.001))
        per_cast.append(render_both(renderer, case))
    # n = 2000 runs the two-level filter: a candidate GROUP brings its 8 members to the exact test (84 per ray cast, of 2000); what this test
    # is about is that the count does not grow when the scene moves away from the world origin
    assert 0 < per_cast[0] < (40 if n <= 512 else 160) and per_cast[1] < 3 * per_cast[0] + 5, per_cast
