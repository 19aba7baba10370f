"""rt3_gather_rows: the shards' compact device rows land at their interleaved positions in one device frame — the native
(no host memory, no de-interleave kernel) gather of SURVEY.md section 8e, here with every shard on the one GPU of the box (self-peer)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cases import hip_render, hip_upload, mode_x_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("height,tile_rows,count", [(36, 8, 2), (37, 4, 3), (50, 16, 4), (9, 16, 2), (33, 1, 8), (5, 2, 7)])
def test_gather_reassembles_the_frame_on_the_device(rt3, renderer, height, tile_rows, count):
    """Ragged last blocks, more shards than row blocks, single-row interleave."""
    case = dict(mode_x_cases()["three_spheres_64x36x16_d8"])
    w = 64
    cam = rt3.Camera().update(w, height, 1.0, np.float32(w) / np.float32(height) * np.float32(2.0), 2.0)
    case["cam"] = cam.c
    whole = hip_render(renderer, case, width=w, height=height, spp=3)
    L = rt3.lib()
    d_frame = L.rt3_device_alloc_words(renderer._ctx, w * height)
    shards = [rt3.initialize_renderer(0) for _ in range(count - 1)]          # other contexts: what other GPUs would be
    try:
        for i in range(count):
            r = renderer if i == 0 else shards[i - 1]
            if i:
                hip_upload(r, case)
            p = rt3.make_params(**dict(case["params"], width=w, height=height, spp=3, tile_rows=tile_rows, tile_index=i, tile_count=count))
            rows = rt3.rows_owned(p)
            d_tile = L.rt3_device_alloc_words(r._ctx, max(1, rows * w))
            try:
                if rows:
                    r.render_path_device(case["cam"], p, d_tile, L.rt3_stream(r._ctx))
                renderer.gather_rows(d_frame, r, d_tile, p)
                assert L.rt3_synchronize(r._ctx) == 0
            finally:
                L.rt3_device_free(r._ctx, C.c_void_p(d_tile))
        frame = np.zeros((height, w), np.uint32)
        assert L.rt3_device_read_words(renderer._ctx, C.c_void_p(d_frame), w * height, frame.ctypes.data_as(C.c_void_p)) == 0
    finally:
        L.rt3_device_free(renderer._ctx, C.c_void_p(d_frame))
        for r in shards:
            r.close()
    assert np.array_equal(frame, whole)


def test_cpp_host_multi_device_path_moves_no_pixels_through_host_memory(rt3, tmp_path):
    """The rt3 binary with --gpus 3 (three contexts on the one GPU: RT3_DEVICE_LIST) == --gpus 1, byte for byte."""
    exe = os.path.join(os.path.dirname(rt3.LIB_PATH), "rt3")
    outs = []
    for gpus, env in ((1, {}), (3, {"RT3_DEVICE_LIST": "0,0,0"})):
        out = str(tmp_path / ("g%d.ppm" % gpus))
        subprocess.run([exe, "-f", "ppm", "-W", "200", "-H", "113", "--scene", "weekend", "--spp", "4", "--depth", "12", "--seed", "3",
                        "--gpus", str(gpus), out], check=True, env=dict(os.environ, **env), capture_output=True, timeout=300)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 3 * 200 * 113
