"""Mode X on the device == the CPU restatement, BIT FOR BIT (tolerance 0 per channel), through the C ABI.

Mode X has no reference implementation: parity here is against this repository's own oracle, which
shares no code with the kernels.  Both sides use the same counter-based RNG and state every fused multiply-add
explicitly, so the comparison is exact; any stated tolerance would only hide a real difference."""
import os

import numpy as np
import pytest

from cases import GOLDEN, hip_render, hip_upload, mode_x_cases, oracle_render, rt3 as _rt3

pytestmark = pytest.mark.gpu


def assert_same(got, want, what=""):
    assert got.shape == want.shape
    bad = got != want
    assert not bad.any(), "%s: %d of %d pixels differ (first at %s: got %08x want %08x)" % (
        what, bad.sum(), bad.size, tuple(np.argwhere(bad)[0]), got[bad][0], want[bad][0])


@pytest.mark.parametrize("name", sorted(mode_x_cases().keys()))
def test_small_goldens(renderer, name):
    z = np.load(os.path.join(GOLDEN, "mode_x_small.npz"))
    assert_same(hip_render(renderer, mode_x_cases()[name]), z[name], name)


def test_config1_three_spheres_full_size(rt3, renderer):
    """BASELINE.json configs[0]: three-sphere Lambertian scene, 400x225, 16 spp, depth 8."""
    cr, mats = rt3.scene_three_spheres()
    cam = rt3.Camera().update(400, 225, 1.0, np.float32(400) / np.float32(225) * np.float32(2.0), 2.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=400, height=225, spp=16, max_depth=8, seed=1, flags=1))
    want, casts = oracle_render(case, threads=16)
    assert_same(hip_render(renderer, case), want, "config 1")
    st = renderer.stats()
    assert st.ray_casts == casts and st.samples == 400 * 225 * 16 and st.prim_tests == casts * 3


def test_weekend_scene_reduced(rt3, renderer):
    cr, mats = rt3.scene_weekend(42)
    cam = rt3.weekend_camera(320, 180)
    case = dict(spheres=cr, smats=mats, cam=cam.c,
                params=dict(width=320, height=180, spp=8, max_depth=50, seed=1, flags=1, lens_radius=0.05))
    want, casts = oracle_render(case, threads=16)
    assert_same(hip_render(renderer, case), want, "weekend 320x180x8")
    assert renderer.stats().ray_casts == casts


def test_config2_full_size_rows_against_the_oracle_and_invariances(rt3, renderer):
    """BASELINE.json configs[1] at full size (1920x1080, 512 spp, depth 50).  The oracle cannot render 1e9 samples, so
    (a) four full-width rows spread over the frame are compared with the oracle bit for bit, and (b) the whole frame
    must be bitwise invariant to the sample-storage batch size and to sharding (size-independent properties)."""
    cr, mats = rt3.scene_weekend(42)
    W, H = 1920, 1080
    cam = rt3.weekend_camera(W, H)
    base = dict(width=W, height=H, spp=512, max_depth=50, seed=1, flags=1, lens_radius=0.05)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=base)
    whole = hip_render(renderer, case)
    rows_case = dict(tile_rows=1, tile_index=135, tile_count=270)            # rows 135, 405, 675, 945
    want, _ = oracle_render(case, threads=16, **rows_case)
    got_rows = np.stack([whole[y] for y in (135, 405, 675, 945)])
    assert_same(got_rows, want, "config 2 rows")
    # (b1) several sample batches instead of one
    renderer.set_sample_storage_cap(3 << 30)
    try:
        batched = hip_render(renderer, case, upload=False)
        assert renderer.stats().launches > 1
    finally:
        renderer.set_sample_storage_cap(16 << 30)
    assert_same(batched, whole, "config 2 batched")
    # (b2) the two shards of a 2-GPU run reassemble to the same frame
    plist = [rt3.make_params(**dict(base, tile_rows=8, tile_index=i, tile_count=2)) for i in range(2)]
    tiles = [hip_render(renderer, case, upload=False, tile_rows=8, tile_index=i, tile_count=2) for i in range(2)]
    assert_same(rt3.deinterleave(tiles, plist, H, W), whole, "config 2 sharded")


def test_config2_full_frame_equals_the_oracle_frame(rt3, renderer):
    """The WHOLE headline frame (BASELINE.json configs[1]: 1920x1080, 512 spp, depth 50 — 1.06e9 samples, 2.9e9 ray casts)
    against the frame the CPU oracle rendered with the same parameters (tests/golden/make_config2_golden.py, minutes of
    CPU time, done once in the build container): SHA-256 of all pixels, the ray-cast count, and a CRC per row to say where."""
    import hashlib
    import json
    import zlib
    gold = json.load(open(os.path.join(GOLDEN, "config2_full.json")))
    cr, mats = rt3.scene_weekend(gold["scene_seed"])
    W, H = gold["width"], gold["height"]
    cam = rt3.weekend_camera(W, H)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=W, height=H, spp=gold["spp"], max_depth=gold["max_depth"],
                                                              seed=gold["seed"], flags=1, lens_radius=gold["lens_radius"]))
    img = np.ascontiguousarray(hip_render(renderer, case), dtype="<u4")
    assert img.shape == (H, W)
    bad_rows = [y for y in range(H) if zlib.crc32(img[y].tobytes()) != gold["row_crc32"][y]]
    assert not bad_rows, "%d rows differ from the oracle's frame, first %r" % (len(bad_rows), bad_rows[:8])
    assert hashlib.sha256(img.tobytes()).hexdigest() == gold["sha256"]
    assert renderer.stats().ray_casts == gold["ray_casts"]


def test_config3_4k_1024spp_rows_against_the_oracle_and_the_union_of_eight_shards(rt3, renderer):
    """BASELINE.json configs[2]: the same scene at 3840x2160, 1024 spp, depth 50, framebuffer tiled across 8 GPUs.  Full size on the
    one GPU of the box: (a) two full-width rows of the whole frame against the oracle, bit for bit; (b) the eight shards of the
    N = 8 run (single-row interleave, as bench.py shards) rendered one after the other on this device reassemble to the whole
    frame bitwise — the image does not depend on the number of GPUs (tiling intent: raytracer_v4.glsl:70-79,193-194)."""
    cr, mats = rt3.scene_weekend(42)
    W, H = 3840, 2160
    cam = rt3.weekend_camera(W, H)
    base = dict(width=W, height=H, spp=1024, max_depth=50, seed=1, flags=1, lens_radius=0.05)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=base)
    whole = hip_render(renderer, case)
    st = renderer.stats()
    assert st.samples == W * H * 1024 and st.launches > 1                      # 8.5e9 samples: several sample batches
    want, _ = oracle_render(case, threads=16, tile_rows=1, tile_index=700, tile_count=1080)    # rows 700 and 1780
    assert_same(np.stack([whole[700], whole[1780]]), want, "config 3 rows")
    plist = [rt3.make_params(**dict(base, tile_rows=1, tile_index=i, tile_count=8)) for i in range(8)]
    tiles = [hip_render(renderer, case, upload=False, tile_rows=1, tile_index=i, tile_count=8) for i in range(8)]
    assert all(t.shape == (H // 8, W) for t in tiles)
    assert_same(rt3.deinterleave(tiles, plist, H, W), whole, "config 3: union of the 8 shards")


def test_config4_many_spheres_multi_tile(rt3, renderer):
    """BASELINE.json configs[3] shape: 100k Lambertian spheres streamed through LDS in 1024-sphere tiles; 1920x1080 at
    reduced spp, two full-width rows against the oracle + shard invariance of the whole frame."""
    cr, mats = rt3.scene_stress(100000, 43)
    W, H = 1920, 1080
    cam = rt3.Camera().look_at(W, H, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    base = dict(width=W, height=H, spp=2, max_depth=6, seed=9, flags=1)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=base)
    whole = hip_render(renderer, case)
    want, _ = oracle_render(case, threads=16, tile_rows=1, tile_index=300, tile_count=540)     # rows 300 and 840
    assert_same(np.stack([whole[300], whole[840]]), want, "config 4 rows")
    plist = [rt3.make_params(**dict(base, tile_rows=8, tile_index=i, tile_count=3)) for i in range(3)]
    tiles = [hip_render(renderer, case, upload=False, tile_rows=8, tile_index=i, tile_count=3) for i in range(3)]
    assert_same(rt3.deinterleave(tiles, plist, H, W), whole, "config 4 sharded")


def test_config4_depth_50_rows_against_the_oracle(rt3, renderer):
    """BASELINE.json configs[3] with its own depth (50) at full width: paths bounce through the 196 LDS tiles up to 50 times.
    Two full-width rows (4 spp of the 256) against the oracle, from a render of just those rows and from the whole frame."""
    cr, mats = rt3.scene_stress(100000, 43)
    W, H = 1920, 1080
    cam = rt3.Camera().look_at(W, H, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=W, height=H, spp=4, max_depth=50, seed=9, flags=1))
    rows = dict(tile_rows=1, tile_index=200, tile_count=540)                   # rows 200 and 740
    want, casts = oracle_render(case, threads=16, **rows)
    got = hip_render(renderer, case, **rows)
    assert_same(got, want, "config 4 depth 50 rows")
    assert renderer.stats().ray_casts == casts
    whole = hip_render(renderer, case, upload=False)
    assert_same(np.stack([whole[200], whole[740]]), want, "config 4 depth 50, whole frame")


def _golden_rows(name):
    import json
    import os
    import zlib
    from cases import GOLDEN
    meta = json.load(open(os.path.join(GOLDEN, "config45_rows.json")))[name]
    rows = np.load(os.path.join(GOLDEN, "config45_rows.npz"))[name]
    assert [int(zlib.crc32(np.ascontiguousarray(r, "<u4").tobytes())) for r in rows] == meta["row_crc32"]      # the fixture is intact
    return meta, rows


def test_config4_full_spp_rows_equal_the_oracle_rows(rt3, renderer):
    """BASELINE.json configs[3] at its OWN sample budget: 100 000 spheres, 1920x1080, 256 spp, depth 50.  Two full-width rows (270 and 810)
    against the oracle's, which were rendered once in the build container (tests/golden/make_config45_golden.py, 450 s on 6 cores) and are
    committed with their CRCs and ray-cast count: every sample index 0 .. 255 of these pixels goes through the tiled matrix-filter kernel."""
    meta, want = _golden_rows("config4")
    cr, mats = rt3.scene_stress(100000, 43)
    W, H = 1920, 1080
    cam = rt3.Camera().look_at(W, H, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    assert meta["params"] == dict(width=W, height=H, spp=256, max_depth=50, seed=9, flags=1)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=meta["params"])
    got = hip_render(renderer, case, **meta["shard"])
    assert_same(got, want, "config 4, 256 spp, rows %s" % meta["frame_rows"])
    assert renderer.stats().ray_casts == meta["ray_casts"]


def test_config4_full_frame_at_256_spp_is_the_union_of_its_eight_shards(rt3, renderer):
    """The complete config-4 frame at 256 spp, depth 50 (5.3e8 samples, 2.5e9 ray casts over 100 000 spheres): the whole frame equals the
    union of the eight single-row-interleaved shards an 8-GPU run renders, and contains the two oracle rows."""
    meta, want = _golden_rows("config4")
    cr, mats = rt3.scene_stress(100000, 43)
    W, H = 1920, 1080
    cam = rt3.Camera().look_at(W, H, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=meta["params"])
    whole = hip_render(renderer, case)
    assert renderer.stats().samples == W * H * 256
    assert_same(np.stack([whole[r] for r in meta["frame_rows"]]), want, "config 4, 256 spp, whole frame")
    plist = [rt3.make_params(**dict(meta["params"], tile_rows=1, tile_index=i, tile_count=8)) for i in range(8)]
    tiles = [hip_render(renderer, case, upload=False, tile_rows=1, tile_index=i, tile_count=8) for i in range(8)]
    assert_same(rt3.deinterleave(tiles, plist, H, W), whole, "config 4, 256 spp: union of the 8 shards")


def test_config5_full_spp_rows_equal_the_oracle_rows(rt3, renderer):
    """BASELINE.json configs[4] at its OWN sample budget: the Cornell-style box (47 106 triangles, emissive quad), 1024x1024, 2048 spp,
    depth 50, black background.  Rows 300 and 812 against the oracle's (make_config45_golden.py: about 80 minutes on 6 cores)."""
    meta, want = _golden_rows("config5")
    faces, verts, fmats = rt3.scene_cornell(64)
    cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
    assert meta["params"] == dict(width=1024, height=1024, spp=2048, max_depth=50, seed=6, flags=3)
    case = dict(faces=faces, verts=verts, fmats=fmats, cam=cam.c, params=meta["params"])
    got = hip_render(renderer, case, **meta["shard"])
    assert_same(got, want, "config 5, 2048 spp, rows %s" % meta["frame_rows"])
    assert renderer.stats().ray_casts == meta["ray_casts"]


def test_stress_scene_small_whole_image(rt3, renderer):
    cr, mats = rt3.scene_stress(3000, 7)                                     # 3 LDS tiles, last one ragged
    cam = rt3.Camera().look_at(160, 90, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=160, height=90, spp=4, max_depth=8, seed=2, flags=1))
    want, casts = oracle_render(case, threads=16)
    assert_same(hip_render(renderer, case), want, "stress 3000")
    assert renderer.stats().ray_casts == casts


def test_config5_cornell_triangles_multi_tile(rt3, renderer):
    """BASELINE.json configs[4] shape: ~47k triangles with an emissive quad, black background, 1024x1024; reduced spp,
    two rows against the oracle, plus a small whole image."""
    faces, verts, fmats = rt3.scene_cornell(64)
    cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
    case = dict(faces=faces, verts=verts, fmats=fmats, cam=cam.c,
                params=dict(width=1024, height=1024, spp=2, max_depth=5, seed=4, flags=1 | 2))
    whole = hip_render(renderer, case)
    want, _ = oracle_render(case, threads=16, tile_rows=1, tile_index=200, tile_count=512)     # rows 200 and 712
    assert_same(np.stack([whole[200], whole[712]]), want, "config 5 rows")
    # the config's own depth (50): paths end at the emitter, through the open front or after 50 casts (7 casts per sample on average)
    deep = dict(case, params=dict(case["params"], max_depth=50, seed=6))
    rows = dict(tile_rows=1, tile_index=300, tile_count=512)                   # rows 300 and 812
    want, casts = oracle_render(deep, threads=16, **rows)
    assert_same(hip_render(renderer, deep, upload=False, **rows), want, "config 5 depth 50 rows")
    assert renderer.stats().ray_casts == casts and casts > 5 * 2 * 2 * 1024
    whole = hip_render(renderer, deep, upload=False)
    assert_same(np.stack([whole[300], whole[812]]), want, "config 5 depth 50, whole frame")
    small_f, small_v, small_m = rt3.scene_cornell(12)                        # 1658 faces: 7 LDS tiles
    cam = rt3.Camera().update(96, 96, 2.0, 2.0, 2.0)
    case = dict(faces=small_f, verts=small_v, fmats=small_m, cam=cam.c,
                params=dict(width=96, height=96, spp=16, max_depth=8, seed=4, flags=1 | 2))
    want, casts = oracle_render(case, threads=16)
    assert_same(hip_render(renderer, case), want, "cornell g12")
    assert renderer.stats().ray_casts == casts
    assert ((want >> 8) & 0xFFFFFF).max() > 0                                # the light is visible


def test_mesh_and_spheres_together(rt3, renderer, oracle):
    """Triangles and analytic spheres in one scene, all four materials, entity API with the analytic-sphere extension."""
    ents = [rt3.create_sphere((0.0, -100.5, -3.0), 100.0, 0, 0, (0, 0, 0), material=rt3.lambertian((0.6, 0.6, 0.3))),
            rt3.create_sphere((0.0, 0.0, -3.0), 0.5, 0, 0, (0, 0, 0), material=rt3.dielectric(1.5)),
            rt3.create_sphere((1.1, 0.0, -3.0), 0.5, 0, 0, (0, 0, 0), material=rt3.metal((0.8, 0.6, 0.2), 0.3)),
            rt3.create_sphere((-1.1, 0.0, -3.0), 0.5, 12, 9, (0.0, 0.0, 1.0), material=rt3.lambertian((0.2, 0.3, 0.8))),
            rt3.create_triangle((-2.0, 1.0, -4.0), (2.0, 1.0, -4.0), (0.0, 2.5, -4.0), (0, 0, 0), material=rt3.emissive((4.0, 3.0, 2.0))),
            rt3.create_triangle((-3.0, -0.5, -2.0), (-3.0, -0.5, -6.0), (-3.0, 2.0, -4.0), (0, 0, 0), material=rt3.metal((0.9, 0.9, 0.9), 0.0))]
    renderer.prerender(ents)
    renderer.configure(spp=9, max_depth=10, seed=3, flags=rt3.FLAG_GAMMA2)
    cam = rt3.Camera().update(200, 112, 1.0, np.float32(200) / np.float32(112) * np.float32(2.0), 2.0)
    renderer.render(cam)
    renderer.configure(spp=None)
    # the same scene, flattened independently for the oracle
    parts, fm = [], []
    for e in ents[3:]:
        f, v = rt3.pre_render_entity(e)
        parts.append((f, v))
        fm.append(np.repeat(e.material, len(f)))
    faces, verts = rt3.merge_entities(parts)
    case = dict(faces=faces, verts=verts, fmats=np.concatenate(fm),
                spheres=np.float32([[0.0, -100.5, -3.0, 100.0], [0.0, 0.0, -3.0, 0.5], [1.1, 0.0, -3.0, 0.5]]),
                smats=np.concatenate([e.material for e in ents[:3]]), cam=cam.c,
                params=dict(width=200, height=112, spp=9, max_depth=10, seed=3, flags=1))
    want, _ = oracle_render(case, threads=16)
    assert_same(cam.get_frame().d(), want, "mesh + spheres")


def test_candidate_queue_overflow(rt3, renderer):
    """A ray down a row of 60 overlapping spheres has more candidates than the per-lane LDS queue holds (16): the
    flush-when-full path must give the same nearest hit as the sequential oracle."""
    n = 60
    cr = np.zeros((n, 4), np.float32)
    cr[:, 2] = -3.0 - 0.35 * np.arange(n)[::-1]          # far spheres first: every one of them improves the best t
    cr[:, 0] = 0.01 * np.sin(np.arange(n))
    cr[:, 3] = 0.5
    mats = np.zeros(n, rt3.MATERIAL)
    mats["kind"] = np.arange(n) % 3 + 1
    mats["rgb"] = np.float32([0.9, 0.7, 0.5])
    mats["param"] = np.where(mats["kind"] == 3, 1.5, 0.2).astype(np.float32)
    cam = rt3.Camera().update(128, 72, 1.0, np.float32(128) / np.float32(72) * np.float32(2.0), 2.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=128, height=72, spp=4, max_depth=12, seed=8, flags=1))
    want, casts = oracle_render(case, threads=16)
    assert_same(hip_render(renderer, case), want, "queue overflow")
    assert renderer.stats().ray_casts == casts


@pytest.mark.parametrize("spp,depth", [(1, 1), (2, 3), (4, 50), (25, 4)])
def test_spp_and_depth_corners(rt3, renderer, spp, depth):
    case = dict(mode_x_cases()["weekend_96x54x4_d50_lens"])
    want, _ = oracle_render(case, threads=16, spp=spp, max_depth=depth)
    assert_same(hip_render(renderer, case, spp=spp, max_depth=depth), want, "spp %d depth %d" % (spp, depth))


def test_ragged_frame_and_tiny_frames(rt3, renderer):
    case = dict(mode_x_cases()["three_spheres_64x36x16_d8"])
    for w, h in ((2, 2), (3, 67), (131, 5), (257, 129)):
        cam = rt3.Camera().update(w, h, 1.0, np.float32(w) / np.float32(h) * np.float32(2.0), 2.0)
        c = dict(case, cam=cam.c)
        want, _ = oracle_render(c, width=w, height=h, spp=3)
        assert_same(hip_render(renderer, c, width=w, height=h, spp=3), want, "%dx%d" % (w, h))


def test_single_gpu_gather_path(rt3, renderer):
    """The product's shard/gather object on one GPU (world = 1): device tile -> device frame."""
    import importlib
    import torch
    shard = importlib.import_module("raytracer-3_amd.shard")
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    hip_upload(renderer, case)
    p = rt3.make_params(**case["params"])
    g = shard.FrameGatherer(rt3, [p], 0, torch.device("cuda", 0))
    renderer.render_path_device(case["cam"], p, g.tile.data_ptr(), torch.cuda.current_stream().cuda_stream)
    frame = g.gather()
    torch.cuda.synchronize()
    z = np.load(os.path.join(GOLDEN, "mode_x_small.npz"))
    assert_same(frame.cpu().numpy().view(np.uint32), z["three_spheres_64x36x16_d8"])


def test_errors(rt3, renderer):
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    hip_upload(renderer, case)
    with pytest.raises(rt3.Fatal):
        hip_render(renderer, case, upload=False, spp=0)
    with pytest.raises(rt3.Fatal):
        hip_render(renderer, case, upload=False, tile_count=2, tile_index=2)
    bad = case["spheres"].copy()
    bad[1, 3] = 0.0
    with pytest.raises(rt3.Fatal, match="radius"):
        renderer.set_spheres(bad, case["smats"])
    renderer.set_mesh(np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32))
    renderer.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))
    with pytest.raises(rt3.Fatal, match="no scene"):
        hip_render(renderer, case, upload=False)
