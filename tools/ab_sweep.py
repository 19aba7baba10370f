#!/usr/bin/env python3
"""Interleaved A/B of libraries on the bench workload at several spp and on the N = 8 shard (fixed cost per launch):
    python tools/ab_sweep.py libA.so libB.so ..."""
import ctypes as C
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

rt3 = importlib.import_module("raytracer-3_amd")


def main():
    paths = sys.argv[1:]
    libs = []
    for path in paths:
        L = C.CDLL(os.path.abspath(path))
        L.rt3_create.restype = C.c_void_p
        L.rt3_last_error.restype = C.c_char_p
        libs.append((L, C.c_void_p(L.rt3_create(0))))
    cr, mats = rt3.scene_weekend(42)
    W, H = 1920, 1080
    cam = rt3.weekend_camera(W, H)
    for L, ctx in libs:
        L.rt3_set_spheres(ctx, cr.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p), C.c_uint32(len(cr)))
    cases = [("full frame %4d spp" % spp, dict(spp=spp)) for spp in (8, 64, 512)]
    cases += [("N=8 shard (rows i mod 8 == %d), 512 spp" % i, dict(spp=512, tile_rows=1, tile_index=i, tile_count=8)) for i in (0, 7)]
    for name, kw in cases:
        p = rt3.make_params(W, H, max_depth=50, seed=1, flags=1, lens_radius=0.05, **kw)
        rows = rt3.rows_owned(p)
        outs, times = [], [[] for _ in libs]
        for r in range(5):
            for i, (L, ctx) in enumerate(libs):
                out = np.zeros((rows, W), np.uint32)
                assert L.rt3_render_path(ctx, C.byref(cam.c), C.byref(p), out.ctypes.data_as(C.c_void_p)) == 0, L.rt3_last_error(ctx)
                st = rt3.rt3_stats()
                L.rt3_get_stats(ctx, C.byref(st))
                if r > 0:
                    times[i].append(st.total_ms)
                else:
                    outs.append(out)
        med = [statistics.median(t) for t in times]
        for i in range(1, len(libs)):
            print("%-42s A %8.3f ms  %-26s %8.3f ms  x%.4f  identical %s" % (name, med[0], os.path.basename(paths[i]), med[i], med[i] / med[0],
                                                                             np.array_equal(outs[0], outs[i])), flush=True)


if __name__ == "__main__":
    main()
