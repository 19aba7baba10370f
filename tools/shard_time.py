#!/usr/bin/env python3
"""Kernel time of the headline frame and of one of its N shards (single-row interleave, as bench.py --gpus N renders them) on ONE GPU:
what the fixed cost per launch does to N-GPU scaling.    python tools/shard_time.py"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
W, H = 1920, 1080
r = rt3.HipRenderer()
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.set_spheres(cr, mats)
full = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for idx in range(n):
        p = rt3.make_params(W, H, spp=512, max_depth=50, seed=1, flags=1, lens_radius=0.05, tile_rows=1, tile_index=idx, tile_count=n)        # single-row interleave, as bench.py (TILE_ROWS)
        r.render_path(cam.c, p)
        r.render_path(cam.c, p)
        st = r.stats()
        worst = max(worst, st.total_ms)
    full = full or worst
    print("N = %d: slowest shard %.3f ms (kernel + accumulate + resolve), ideal %.3f ms, efficiency %.3f" % (n, worst, full / n, full / n / worst), flush=True)
