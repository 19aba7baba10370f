"""Progressive / resumable accumulation (reduce_v1.glsl:28-76, SampleStorage of raytracer_v4.glsl:107-111):
rt3_render_path_range + rt3_accum_download / rt3_accum_upload through the C ABI, against the oracle's sums bit for bit."""
import numpy as np
import pytest

from cases import hip_upload, mode_x_cases, oracle_render

pytestmark = pytest.mark.gpu


def weekend(rt3, w, h, spp, flags):
    cr, mats = rt3.scene_weekend(42)
    return dict(spheres=cr, smats=mats, cam=rt3.weekend_camera(w, h).c,
                params=dict(width=w, height=h, spp=spp, max_depth=50, seed=1, flags=flags, lens_radius=0.05))


def test_four_calls_with_a_checkpoint_in_the_middle_equal_one_call_and_the_oracle(rt3, renderer, oracle):
    """512 spp in 4 uneven calls; after the second the accumulation is downloaded, the context is used for something else, a
    SECOND context restores the checkpoint and finishes: same frame as one call, same as the oracle, sums and squares included."""
    case = weekend(rt3, 160, 90, 512, 1 | rt3.FLAG_VARIANCE)
    p = rt3.make_params(**case["params"])
    hip_upload(renderer, case)
    one = renderer.render_path(case["cam"], p)
    acc_one, sq_one, done = renderer.accum_download(p, want_sq=True)
    assert done == 512
    renderer.render_path_range(case["cam"], p, 0, 100)
    preview = renderer.render_path_range(case["cam"], p, 100, 28)
    acc, sq, done = renderer.accum_download(p, want_sq=True)
    assert done == 128
    # the preview is the frame over the first 128 samples of the 512-sample law, as the oracle resolves it
    ocam = oracle.copy_camera(case["cam"])
    op = oracle.make_params(**case["params"])
    okw = dict(spheres=case["spheres"], smats=np.ascontiguousarray(case["smats"]).view(oracle.MATERIAL), threads=16)
    o_img, o_acc, o_sq, _ = oracle.render_path_range(ocam, op, 0, 128, **okw)
    assert np.array_equal(preview, o_img)
    assert acc.tobytes() == o_acc.tobytes() and sq.tobytes() == o_sq.tobytes()
    # the first context goes on to other work; a fresh one resumes from the checkpoint
    renderer.render_path(case["cam"], rt3.make_params(**dict(case["params"], spp=3, seed=77)))
    other = rt3.initialize_renderer(0)
    try:
        hip_upload(other, case)
        other.accum_upload(case["cam"], p, acc, sq, 128)
        other.render_path_range(case["cam"], p, 128, 300)
        last = other.render_path_range(case["cam"], p, 428, 84)
        acc_end, sq_end, done = other.accum_download(p, want_sq=True)
    finally:
        other.close()
    assert done == 512 and np.array_equal(last, one)
    assert acc_end.tobytes() == acc_one.tobytes() and sq_end.tobytes() == sq_one.tobytes()
    o_img, o_acc, o_sq, _ = oracle.render_path_range(ocam, op, 128, 384, o_acc, o_sq, **okw)
    assert np.array_equal(one, o_img)
    assert acc_one.tobytes() == o_acc.tobytes() and sq_one.tobytes() == o_sq.tobytes()
    # the variance flag never changes a pixel
    assert np.array_equal(one, renderer.render_path(case["cam"], rt3.make_params(**dict(case["params"], flags=1))))


def test_progressive_over_batches_shards_and_faces(rt3, renderer):
    """The same invariance where a call is itself split into sample batches, on a shard, and on a triangle scene."""
    case = mode_x_cases()["cornell_g4_48x48x8_d6_black"]
    hip_upload(renderer, case)
    params = dict(case["params"], spp=32, tile_rows=4, tile_index=1, tile_count=3)
    p = rt3.make_params(**params)
    want, _ = oracle_render(case, threads=16, **{k: params[k] for k in ("spp", "tile_rows", "tile_index", "tile_count")})
    renderer.set_sample_storage_cap(1 << 20)                # 1 MiB: 768 owned pixels x 12 B -> several batches per call
    try:
        renderer.render_path_range(case["cam"], p, 0, 7)
        renderer.render_path_range(case["cam"], p, 7, 24)
        got = renderer.render_path_range(case["cam"], p, 31, 1)
    finally:
        renderer.set_sample_storage_cap(16 << 30)
    assert np.array_equal(got, want)


def test_a_call_that_does_not_continue_the_accumulation_is_refused(rt3, renderer):
    case = mode_x_cases()["three_spheres_64x36x16_d8"]
    hip_upload(renderer, case)
    p = rt3.make_params(**case["params"])
    renderer.render_path_range(case["cam"], p, 0, 4)
    with pytest.raises(rt3.Fatal, match="continue"):
        renderer.render_path_range(case["cam"], p, 5, 4)                        # a gap
    renderer.render_path_range(case["cam"], p, 0, 4)
    with pytest.raises(rt3.Fatal, match="continue"):
        renderer.render_path_range(case["cam"], rt3.make_params(**dict(case["params"], seed=2)), 4, 4)   # other params
    with pytest.raises(rt3.Fatal, match="range"):
        renderer.render_path_range(case["cam"], p, 4, 13)                       # past spp
    with pytest.raises(rt3.Fatal, match="VARIANCE"):
        renderer.accum_download(p, want_sq=True)
    got = renderer.render_path_range(case["cam"], p, 4, 12)                     # the refused calls left the accumulation alone
    assert np.array_equal(got, renderer.render_path(case["cam"], p))
