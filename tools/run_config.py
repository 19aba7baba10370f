#!/usr/bin/env python3
"""Renders one BASELINE.json config shape (4, 5) or the Mode-R fixture (r) once, for profiling under rocprofv3:
    python tools/run_config.py 4|5|r [spp]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
r = rt3.HipRenderer()
if which == "r":
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "builtin_scene.npz"))
    r.set_mesh(z["faces"].view(rt3.GFACE).reshape(-1), z["verts"])
    r.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))
    cam = rt3.main_camera(1920, 1080)
    for _ in range(4):
        r.render(cam)
    st = r.stats()
    print("mode R 1920x1080: kernel %.4f ms" % st.trace_ms)
    sys.exit(0)
if which == "4":
    cr, mats = rt3.scene_stress(100000, 43)
    r.set_spheres(cr, mats)
    cam = rt3.Camera().look_at(1920, 1080, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    p = rt3.make_params(1920, 1080, spp=spp, max_depth=50, flags=1)
else:
    faces, verts, fm = rt3.scene_cornell(64)
    r.set_mesh(faces, verts, fm)
    r.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))
    cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
    p = rt3.make_params(1024, 1024, spp=spp, max_depth=50, flags=3)
r.render_path(cam.c, p)
st = r.stats()
print("config %s spp %d: trace %.3f ms, %d casts, %d mfma, %.2f exact tests per cast, %.1f bound tests per cast, %d filter rows per cast, %.3e prim tests/s"
      % (which, spp, st.trace_ms, st.ray_casts, st.mfma_instructions, st.exact_tests / max(1, st.ray_casts), st.bound_tests / max(1, st.ray_casts),
         st.filter_tests // max(1, st.ray_casts), st.prim_tests / (st.trace_ms * 1e-3)))
