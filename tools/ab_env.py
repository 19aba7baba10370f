#!/usr/bin/env python3
"""Interleaved A/B of an ENVIRONMENT switch of librt3hip.so on the bench workload (kernel time by HIP events, frames compared):
    python tools/ab_env.py RT3_MFMA_32X32 [spp] [rounds]"""
import importlib, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
var = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
W, H = 1920, 1080
r = rt3.HipRenderer()
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.set_spheres(cr, mats)
p = rt3.make_params(W, H, spp=spp, max_depth=50, seed=1, flags=1, lens_radius=0.05)
times, frames, mfma = {0: [], 1: []}, {}, {}
for i in range(rounds + 1):
    for on in (0, 1):
        if on:
            os.environ[var] = "1"
        else:
            os.environ.pop(var, None)
        out = r.render_path(cam.c, p)
        st = r.stats()
        if i:
            times[on].append(st.trace_ms)
        frames[on], mfma[on] = out, (st.mfma_instructions, st.mfma_flop_per_instruction, st.exact_tests)
a, b = statistics.median(times[0]), statistics.median(times[1])
print("%s: off %.3f ms  on %.3f ms  x%.4f  identical %s  (mfma instr, flop/instr, exact tests) off %s on %s" % (var, a, b, b / a, np.array_equal(frames[0], frames[1]), mfma[0], mfma[1]))
