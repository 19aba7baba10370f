#!/usr/bin/env python3
"""Times every BASELINE.json config on ONE MI355X (kernel time from the C ABI's HIP events) and the Mode-R fixture.
Usage: python tools/perf_configs.py [--quick]   (quick = reduced spp for the two brute-force-heavy configs)"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

rt3 = importlib.import_module("raytracer-3_amd")
quick = "--quick" in sys.argv
only = [a for a in sys.argv[1:] if a.isdigit()]                      # e.g. `4 5`: just those configs


def run_path(r, name, cam, params, reps=1, k_slots=64):           # k_slots: K of the filter the kernel runs (2 K FLOP per test)
    r.render_path(cam.c, params)                       # warm-up (allocations)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r.render_path(cam.c, params)
        wall = time.perf_counter() - t0
        st = r.stats()
        row = dict(config=name, samples=st.samples, ray_casts=st.ray_casts, prim_tests=st.prim_tests, kernel_ms=round(st.trace_ms, 3),
                   total_ms=round(st.total_ms, 3), wall_ms=round(wall * 1e3, 3), launches=st.launches,
                   msamples_per_s=round(st.samples / st.total_ms / 1e3, 1), gtests_per_s=round(st.prim_tests / st.trace_ms / 1e6, 1),
                   mfma=st.mfma_instructions, mfma_frac_of_2500TF=round(st.mfma_instructions * float(st.mfma_flop_per_instruction) / (st.trace_ms * 1e-3) / 2.5e15, 4),
                   exact_per_cast=round(st.exact_tests / max(1, st.ray_casts), 2),
                   # MFMA work is per wave whatever the number of live lanes: tests the matrix cores evaluated / tests that were needed
                   lane_efficiency=round(st.prim_tests * 2.0 * k_slots / max(1.0, st.mfma_instructions * float(st.mfma_flop_per_instruction)), 4))
        if best is None or row["total_ms"] < best["total_ms"]:
            best = row
    print(json.dumps(best), flush=True)
    return best


def main():
    r = rt3.initialize_renderer(0)
    empty_mesh = (np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32))
    # Mode-R fixture: built-in scene (needs the committed fixture)
    z = np.load(os.path.join(ROOT, "tests", "golden", "builtin_scene.npz"))
    r.set_mesh(z["faces"].view(rt3.GFACE).reshape(-1), z["verts"])
    r.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))
    for w, h in ((400, 225), (1920, 1080)):
        cam = rt3.main_camera(w, h)
        r.render(cam)
        r.render(cam)
        st = r.stats()
        print(json.dumps(dict(config="mode R built-in scene %dx%d" % (w, h), samples=st.samples, prim_tests=st.prim_tests,
                              kernel_ms=round(st.trace_ms, 4), msamples_per_s=round(st.samples / st.trace_ms / 1e3, 1),
                              gtests_per_s=round(st.prim_tests / st.trace_ms / 1e6, 1))), flush=True)
    r.set_mesh(*empty_mesh)
    if only:
        return tiled_only(r)
    # config 1
    cr, mats = rt3.scene_three_spheres()
    r.set_spheres(cr, mats)
    cam = rt3.Camera().update(400, 225, 1.0, np.float32(400) / np.float32(225) * np.float32(2.0), 2.0)
    run_path(r, "1: three spheres 400x225x16 d8", cam, rt3.make_params(400, 225, spp=16, max_depth=8, flags=1), reps=3)
    # config 2, 3
    cr, mats = rt3.scene_weekend(42)
    r.set_spheres(cr, mats)
    run_path(r, "2: weekend 1920x1080x512 d50", rt3.weekend_camera(1920, 1080),
             rt3.make_params(1920, 1080, spp=512, max_depth=50, flags=1, lens_radius=0.05), reps=3)
    run_path(r, "3: weekend 3840x2160x1024 d50 (all on 1 GPU)", rt3.weekend_camera(3840, 2160),
             rt3.make_params(3840, 2160, spp=64 if quick else 1024, max_depth=50, flags=1, lens_radius=0.05))
    # config 4
    cr, mats = rt3.scene_stress(100000, 43)
    r.set_spheres(cr, mats)
    cam = rt3.Camera().look_at(1920, 1080, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    run_path(r, "4: 100k spheres 1920x1080x%d d50" % (8 if quick else 256), cam, rt3.make_params(1920, 1080, spp=8 if quick else 256, max_depth=50, flags=1), k_slots=32)
    # config 5
    faces, verts, fm = rt3.scene_cornell(64)
    r.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))
    r.set_mesh(faces, verts, fm)
    cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
    spp5 = 4 if quick else 16
    run_path(r, "5: cornell 47k tris 1024x1024x%d d50 (of 2048 spp)" % spp5, cam, rt3.make_params(1024, 1024, spp=spp5, max_depth=50, flags=3), k_slots=32)


def tiled_only(r):
    spp = int(os.environ.get("SPP", "16"))
    if "4" in only:
        cr, mats = rt3.scene_stress(100000, 43)
        r.set_mesh(np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32))
        r.set_spheres(cr, mats)
        cam = rt3.Camera().look_at(1920, 1080, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
        run_path(r, "4: 100k spheres 1920x1080x%d d50" % spp, cam, rt3.make_params(1920, 1080, spp=spp, max_depth=50, flags=1), k_slots=32)
    if "5" in only:
        faces, verts, fm = rt3.scene_cornell(64)
        r.set_spheres(np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))
        r.set_mesh(faces, verts, fm)
        cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
        run_path(r, "5: cornell 47k tris 1024x1024x%d d50" % spp, cam, rt3.make_params(1024, 1024, spp=spp, max_depth=50, flags=3), k_slots=32)


if __name__ == "__main__":
    main()
