#!/usr/bin/env python3
"""Repeatability of the headline kernel at scale: the bench frame (1920x1080, 64 spp, depth 50) rendered N times by k_trace_mfma32 must be
the same frame every time and equal to the vector-ALU kernels' frame — a candidate lost at random (the v_cvt_pk_bf16_f32
hazard) would show as a few differing pixels.    python tools/repeat_headline.py [renders]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = 1920, 1080
r = rt3.HipRenderer()
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.set_spheres(cr, mats)
p = rt3.make_params(W, H, spp=64, max_depth=50, seed=1, flags=1, lens_radius=0.05)
os.environ["RT3_NO_MFMA"] = "1"
want = r.render_path(cam.c, p).copy()
del os.environ["RT3_NO_MFMA"]
bad = 0
casts = 0
for i in range(n):
    got = r.render_path(cam.c, p)
    st = r.stats()
    casts += st.ray_casts
    d = int((got != want).sum())
    bad += d != 0
    if d:
        print("render %d: %d pixels differ from the vector-ALU frame" % (i, d), flush=True)
print("%d renders by %s, %.3g ray casts, %.3g (ray, sphere) pairs through the filter: %d renders differ from the vector-ALU kernels' frame"
      % (n, "k_trace_mfma32" if st.mfma_flop_per_instruction == 16384 else "k_trace_mfma", casts, casts * float(len(cr)), bad))
sys.exit(1 if bad else 0)
