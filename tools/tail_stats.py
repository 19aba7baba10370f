#!/usr/bin/env python3
"""Fixed cost per launch of the headline kernel: kernel time against spp (slope + intercept) at two depth limits — the intercept is the ramp-up plus the
drain, and the drain is set by the longest paths.    RT3_LIB_PATH=... python tools/tail_stats.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
W, H = 1920, 1080
r = rt3.HipRenderer()
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.set_spheres(cr, mats)
for depth in (50, 8, 2):
    xs, ys = [], []
    for spp in (1, 2, 4, 8, 16, 32):
        p = rt3.make_params(W, H, spp=spp, max_depth=depth, seed=1, flags=1, lens_radius=0.05)
        t = []
        for _ in range(5):
            r.render_path(cam.c, p)
            t.append(r.stats().trace_ms)
        xs.append(spp); ys.append(sorted(t)[2])
    slope, icpt = np.polyfit(xs, ys, 1)
    print("depth %2d: kernel ms at spp 1..32: %s -> %.4f ms/spp + %.3f ms" % (depth, " ".join("%.3f" % y for y in ys), slope, icpt), flush=True)
