#!/usr/bin/env python3
"""How the multi-level filter scales with the scene: config 4's sphere cloud at growing counts (same box, same camera, 1920x1080, 2 spp, depth 50).
    python tools/scale_spheres.py [counts ...]     default 100000 300000 1000000 3000000"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
counts = [int(v) for v in sys.argv[1:]] or [100000, 300000, 1000000, 3000000]
r = rt3.HipRenderer()
cam = rt3.Camera().look_at(1920, 1080, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
p = rt3.make_params(1920, 1080, spp=2, max_depth=50, flags=1)
for n in counts:
    cr, mats = rt3.scene_stress(n, 43)
    t0 = time.perf_counter()
    r.set_spheres(cr, mats)
    t_set = time.perf_counter() - t0
    r.render_path(cam.c, p)
    r.render_path(cam.c, p)
    st = r.stats()
    print("%8d spheres: rt3_set_spheres %.2f s; trace %.2f ms for 2 spp, %d casts, %d rows scanned per cast, %.1f bound + %.1f member tests per cast, %.3e equivalent tests/s"
          % (n, t_set, st.trace_ms, st.ray_casts, st.filter_tests // max(1, st.ray_casts), st.bound_tests / max(1, st.ray_casts), st.exact_tests / max(1, st.ray_casts),
             st.prim_tests / (st.trace_ms * 1e-3)), flush=True)
