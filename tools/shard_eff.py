#!/usr/bin/env python3
"""Estimates strong-scaling efficiency on ONE GPU: renders shard 0 of N of the config-2 frame and compares with 1/N of the
whole-frame time (kernel time and host wall time of a device-output render + sync)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rt3 = importlib.import_module("raytracer-3_amd")
W, H = 1920, 1080
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r = rt3.initialize_renderer(0)
r.prerender([]); r.set_spheres(cr, mats)
base = None
for n in (1, 2, 4, 8):
    rows = []
    for idx in range(n):
        p = rt3.make_params(W, H, spp=512, max_depth=50, seed=1, flags=1, lens_radius=0.05, tile_rows=1, tile_index=idx, tile_count=n)
        out = torch.zeros((rt3.rows_owned(p), W), dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream()
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.render_path_device(cam.c, p, out.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        st = r.stats()
        rows.append((st.trace_ms, st.total_ms, wall))
        if n == 8 and idx >= 1: break
    worst = max(x[2] for x in rows)
    if n == 1: base = worst
    print("N=%d  shard kernel %.2f ms  device total %.2f ms  wall %.2f ms   -> efficiency vs N=1: %.3f" % (n, max(x[0] for x in rows), max(x[1] for x in rows), worst, base / n / worst), flush=True)
