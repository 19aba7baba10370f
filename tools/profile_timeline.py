#!/usr/bin/env python3
"""With a -DRT3_PROFILE build (RT3_LIB_PATH=...): the in-kernel timeline of k_trace_mfma on the bench workload at several sizes."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
W, H = 1920, 1080
r = rt3.HipRenderer()
cr, mats = rt3.scene_weekend(42)
cam = rt3.weekend_camera(W, H)
r.set_spheres(cr, mats)
for kw in (dict(spp=8), dict(spp=64), dict(spp=512), dict(spp=512, tile_rows=1, tile_index=3, tile_count=8)):
    p = rt3.make_params(W, H, max_depth=50, seed=1, flags=1, lens_radius=0.05, **kw)
    r.render_path(cam.c, p)
    r.render_path(cam.c, p)
    st = r.stats()
    print("== %s: trace %.3f ms, total %.3f ms" % (kw, st.trace_ms, st.total_ms), flush=True)
    sys.stderr.flush()
