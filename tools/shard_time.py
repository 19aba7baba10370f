#!/usr/bin/env python3
"""Step time of ONE rank's shard of the bench workload for N = 1, 2, 4, 8 on a single GPU (what strong scaling can reach at best:
the ranks of a real run do this concurrently, plus one RCCL gather)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
rt3 = importlib.import_module("raytracer-3_amd")
W, H, SPP = 1920, 1080, 512
r = rt3.initialize_renderer(0)
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.prerender([]); r.set_spheres(cr, mats)
stream = torch.cuda.current_stream()
t1 = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for idx in sorted(set((0, n - 1))):
        p = rt3.make_params(W, H, spp=SPP, max_depth=50, seed=1, flags=1, lens_radius=0.05, tile_rows=1, tile_index=idx, tile_count=n)
        tile = torch.zeros((rt3.rows_owned(p), W), dtype=torch.int32, device="cuda")
        r.render_path_device(cam.c, p, tile.data_ptr(), stream.cuda_stream); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            r.render_path_device(cam.c, p, tile.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        worst = max(worst, (time.perf_counter() - t0) / 4 * 1e3)
    if n == 1: t1 = worst
    print("N=%d  shard step %.3f ms  ideal %.3f ms  efficiency %.1f %%" % (n, worst, t1 / n, 100.0 * t1 / n / worst), flush=True)
