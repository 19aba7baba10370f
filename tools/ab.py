#!/usr/bin/env python3
"""Interleaved A/B timing of two builds of librt3hip.so in ONE process on ONE device (cdna_hip_programming.md §5.4 rule 24):
    python tools/ab.py libA.so libB.so [spp] [rounds]
Renders BASELINE config 2 (1920x1080, depth 50, thin lens) at `spp` with each library alternately; prints the k_trace time
(HIP events) per round and the medians.  Also checks that both builds produce the same frame."""
import ctypes as C
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime per process, see raytracer-3_amd/__init__.py)

rt3 = importlib.import_module("raytracer-3_amd")


def load(path):
    L = C.CDLL(os.path.abspath(path))
    L.rt3_create.restype = C.c_void_p
    L.rt3_last_error.restype = C.c_char_p
    return L


def main():
    paths = sys.argv[1:3]
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    W, H = 1920, 1080
    cr, mats = rt3.scene_weekend(42)
    cam = rt3.weekend_camera(W, H)
    p = rt3.make_params(W, H, spp=spp, max_depth=50, seed=1, flags=1, lens_radius=0.05)
    libs, ctxs = [], []
    for path in paths:
        L = load(path)
        ctx = C.c_void_p(L.rt3_create(0))
        assert ctx.value, L.rt3_last_error(None)
        assert L.rt3_set_spheres(ctx, cr.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p), C.c_uint32(len(cr))) == 0
        libs.append(L)
        ctxs.append(ctx)
    out = [np.zeros((H, W), np.uint32) for _ in paths]
    times = [[] for _ in paths]
    for r in range(rounds + 1):
        for i, (L, ctx) in enumerate(zip(libs, ctxs)):
            rc = L.rt3_render_path(ctx, C.byref(cam.c), C.byref(p), out[i].ctypes.data_as(C.c_void_p))
            assert rc == 0, L.rt3_last_error(ctx)
            st = rt3.rt3_stats()
            L.rt3_get_stats(ctx, C.byref(st))
            if r > 0:
                times[i].append(st.trace_ms)
    for path, t in zip(paths, times):
        print("%-40s median %.3f ms  min %.3f  all %s" % (os.path.basename(path), statistics.median(t), min(t), ["%.2f" % x for x in t]))
    if len(paths) == 2:
        print("B/A median ratio: %.4f   frames identical: %s" % (statistics.median(times[1]) / statistics.median(times[0]),
                                                                 np.array_equal(out[0], out[1])))


if __name__ == "__main__":
    main()
