#!/usr/bin/env python3
"""Repeatability check on the GPU: the matrix-filter kernel's image against the VALU-scan kernel's, several times over."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
W, H, SPP = 960, 540, int(os.environ.get("SPP", "64"))
r = rt3.HipRenderer()
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.set_spheres(cr, mats)
p = rt3.make_params(W, H, spp=SPP, max_depth=50, seed=1, flags=1, lens_radius=0.05)
os.environ["RT3_NO_MFMA"] = "1"
ref = r.render_path(cam.c, p).copy()
del os.environ["RT3_NO_MFMA"]
bad_total = 0
for k in range(int(os.environ.get("RUNS", "6"))):
    img = r.render_path(cam.c, p)
    bad = int((img != ref).sum())
    bad_total += bad
    print("run %d: %d of %d pixels differ from the VALU kernel" % (k, bad, ref.size), flush=True)
sys.exit(1 if bad_total else 0)
