"""Size-independent properties of the Mode-X restatement (the CPU oracle): they hold for the HIP path too
(tests/test_gpu_mode_x.py runs the same properties on the device)."""
import numpy as np

from cases import load_builtin_scene, mode_x_cases, oracle_render, rt3


def test_image_does_not_depend_on_thread_count():
    case = mode_x_cases()["weekend_96x54x4_d50_lens"]
    a, ca = oracle_render(case, threads=1, width=48, height=27)
    b, cb = oracle_render(case, threads=8, width=48, height=27)
    assert np.array_equal(a, b) and ca == cb


def test_tiles_reassemble_to_the_whole_frame():
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    whole, _ = oracle_render(case)
    for rows, n in ((8, 2), (4, 3), (5, 4)):
        plist = [rt3.make_params(**dict(case["params"], tile_rows=rows, tile_index=i, tile_count=n)) for i in range(n)]
        tiles = [oracle_render(case, tile_rows=rows, tile_index=i, tile_count=n)[0] for i in range(n)]
        assert np.array_equal(rt3.deinterleave(tiles, plist, 36, 64), whole)


def test_mode_x_reduces_to_mode_r_up_to_edge_pixels(oracle):
    """spp 1, flat materials, no gamma, t_min 0: Mode X is Mode R except that it normalises the ray direction, which
    may flip a pixel on a silhouette or shared edge.  >= 99.5 % of pixels must be identical."""
    faces, verts = load_builtin_scene()
    w, h = 200, 112
    ref = oracle.render_mode_r(faces, verts, oracle.camera_update(w, h), w, h)
    cam = rt3.main_camera(w, h)
    case = dict(faces=faces.view(rt3.GFACE), verts=verts, fmats=None, cam=cam.c,
                params=dict(width=w, height=h, spp=1, max_depth=1, seed=1, flags=0, t_min=0.0))
    out, casts = oracle_render(case)
    assert casts == w * h
    same = (out == ref).mean()
    assert same >= 0.995, same
    # sky pixels are identical to the last bit: same gradient formula on the normalised direction can differ by 1 LSB
    diff = np.abs(((out >> 8) & 0xFFFFFF).astype(np.int64) - ((ref >> 8) & 0xFFFFFF).astype(np.int64))
    assert (out != ref).sum() < 0.005 * w * h


def test_reference_primary_flag_makes_mode_x_equal_mode_r_and_the_reference_hash(oracle):
    """SURVEY.md section 0, consequence 1(i) / T7: Mode X with spp 1, depth 1, flat faces, t_min 0, no gamma and
    RT3_FLAG_REFERENCE_PRIMARY (unnormalised primary ray, the reference's literal plane formula, SequentialRenderer.cpp:70,293)
    IS Mode R — every pixel — and so reproduces the PPM SHA-256 recorded from the reference itself."""
    import json
    import os
    from cases import GOLDEN
    pins = json.load(open(os.path.join(GOLDEN, "reference_pins.json")))
    faces, verts = load_builtin_scene()
    w, h = 400, 225
    cam = rt3.main_camera(w, h)
    case = dict(faces=faces.view(rt3.GFACE), verts=verts, fmats=None, cam=cam.c,
                params=dict(width=w, height=h, spp=1, max_depth=1, seed=1, flags=oracle.FLAG_REFERENCE_PRIMARY, t_min=0.0))
    out, casts = oracle_render(case)
    assert casts == w * h
    ref = oracle.render_mode_r(faces, verts, oracle.camera_update(w, h), w, h)        # all rows (row h-1 as the GLSL twin)
    assert np.array_equal(out, ref)
    ref_like = out.copy()
    ref_like[h - 1] = 0                                        # the reference's CPU loop never writes row H-1 (:286)
    assert oracle.sha256(oracle.ppm_bytes(ref_like)) == pins["ppm_sha256"]["400x225"]
    # the seed, the sample count law and the depth do not matter for flat faces: still Mode R with depth 7
    out7, _ = oracle_render(case, max_depth=7, seed=99)
    assert np.array_equal(out7, ref)


def test_progressive_accumulation_is_partition_invariant(oracle):
    """reduce_v1.glsl intent: samples are summed in sample order, so rendering [0, spp) in any consecutive pieces — carrying
    the per-pixel sums (and sums of squares) from call to call — gives the frame, the sums and the ray casts of one call."""
    case = mode_x_cases()["weekend_96x54x4_d50_lens"]
    params = dict(case["params"], width=48, height=27, spp=16, flags=1 | oracle.FLAG_VARIANCE)
    cam = oracle.copy_camera(case["cam"])
    kw = dict(spheres=case["spheres"], smats=np.ascontiguousarray(case["smats"]).view(oracle.MATERIAL))
    p = oracle.make_params(**params)
    whole, acc, sq, casts = oracle.render_path_range(cam, p, 0, 16, **kw)
    plain, plain_casts = oracle_render(case, width=48, height=27, spp=16, flags=1)
    assert np.array_equal(whole, plain) and casts == plain_casts           # the variance flag changes no pixel
    acc2 = sq2 = None
    total = 0
    for begin, count in ((0, 5), (5, 1), (6, 10)):
        img, acc2, sq2, c = oracle.render_path_range(cam, p, begin, count, acc2, sq2, **kw)
        total += c
    assert np.array_equal(img, whole) and total == casts
    assert acc2.tobytes() == acc.tobytes() and sq2.tobytes() == sq.tobytes()
    # a variance estimate from the two sums: non-negative up to rounding, zero where every sample saw the same radiance
    mean, mean_sq = acc[..., :3] / 16.0, sq[..., :3] / 16.0
    assert (mean_sq - mean * mean > -1e-4).all() and (mean_sq - mean * mean).max() > 1e-3


def test_depth_one_lambertian_is_black_and_depth_grows_brightness():
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    d1, _ = oracle_render(case, max_depth=1, spp=4)
    d2, _ = oracle_render(case, max_depth=2, spp=4)
    d8, _ = oracle_render(case, max_depth=8, spp=4)
    lum = lambda img: ((img >> 24) & 255).astype(np.float64).mean()      # noqa: E731
    # at depth 1 a scattering surface returns nothing (book: depth exhausted = black); only sky pixels are lit
    centre = d1[18, 32]
    assert (centre >> 8) == 0
    assert lum(d1) < lum(d2) < lum(d8)


def test_seed_changes_the_noise_not_the_picture():
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    a, _ = oracle_render(case, seed=1)
    b, _ = oracle_render(case, seed=2)
    assert not np.array_equal(a, b)
    ra, rb = ((a >> 24) & 255).astype(np.float64), ((b >> 24) & 255).astype(np.float64)
    assert abs(ra.mean() - rb.mean()) < 2.0
