#!/usr/bin/env python3
"""Kernel time of BASELINE config 2 against spp (fixed cost per launch = intercept of the line)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
W, H = 1920, 1080
r = rt3.HipRenderer()
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.set_spheres(cr, mats)
xs, ys = [], []
for spp in (8, 32, 64, 128, 256, 512):
    p = rt3.make_params(W, H, spp=spp, max_depth=50, seed=1, flags=1, lens_radius=0.05)
    best = 1e9
    for k in range(3):
        r.render_path(cam.c, p)
        best = min(best, r.stats().trace_ms)
    xs.append(spp); ys.append(best)
    print("spp %4d  trace %.3f ms  (%.4f ms/spp)" % (spp, best, best / spp), flush=True)
a, b = np.polyfit(xs, ys, 1)
print("fit: %.4f ms/spp + %.3f ms per launch" % (a, b))
