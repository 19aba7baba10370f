#!/usr/bin/env python3
"""Interleaved A/B of two builds on the config-4 and config-5 shapes (global gathers in the exact test):
    python tools/ab_cfg.py libA.so libB.so [libC.so ...]      (every library against the first; AB_SPP=16,32 renders the two shapes at bench.py's
                                                               16 / 32 samples per pixel instead of 2 / 2)"""
import ctypes as C
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

rt3 = importlib.import_module("raytracer-3_amd")


def main():
    paths = sys.argv[1:]
    libs = []
    for path in paths:
        L = C.CDLL(os.path.abspath(path))
        L.rt3_create.restype = C.c_void_p
        L.rt3_last_error.restype = C.c_char_p
        libs.append((L, C.c_void_p(L.rt3_create(0))))
    cases = []
    spp4, spp5 = [int(v) for v in os.environ.get("AB_SPP", "2,2").split(",")]
    cr, mats = rt3.scene_stress(100000, 43)
    cam = rt3.Camera().look_at(1920, 1080, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    cases.append(("config 4 (100k spheres) 1920x1080x%d" % spp4, cam, rt3.make_params(1920, 1080, spp=spp4, max_depth=50, flags=1), ("sph", cr, mats)))
    faces, verts, fm = rt3.scene_cornell(64)
    cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
    cases.append(("config 5 (47k faces) 1024x1024x%d" % spp5, cam, rt3.make_params(1024, 1024, spp=spp5, max_depth=50, flags=3), ("tri", faces, verts, fm)))
    e_f, e_v = np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32)
    for name, cam, p, scene in cases:
        outs, times = [], [[] for _ in libs]
        for r in range(4):
            for i, (L, ctx) in enumerate(libs):
                if r == 0:
                    if scene[0] == "sph":
                        L.rt3_set_mesh(ctx, None, 0, None, 0, None)
                        L.rt3_set_spheres(ctx, scene[1].ctypes.data_as(C.c_void_p), scene[2].ctypes.data_as(C.c_void_p), C.c_uint32(len(scene[1])))
                    else:
                        L.rt3_set_spheres(ctx, None, None, 0)
                        L.rt3_set_mesh(ctx, scene[1].ctypes.data_as(C.c_void_p), C.c_uint32(len(scene[1])), scene[2].ctypes.data_as(C.c_void_p),
                                       C.c_uint32(len(scene[2])), scene[3].ctypes.data_as(C.c_void_p))
                out = np.zeros((p.height, p.width), np.uint32)
                assert L.rt3_render_path(ctx, C.byref(cam.c), C.byref(p), out.ctypes.data_as(C.c_void_p)) == 0, L.rt3_last_error(ctx)
                st = rt3.rt3_stats()
                L.rt3_get_stats(ctx, C.byref(st))
                if r > 0:
                    times[i].append(st.trace_ms)
                else:
                    outs.append(out)
        med = [statistics.median(t) for t in times]
        for i in range(1, len(libs)):
            print("%-36s A %.2f ms  %-28s %.2f ms  x%.4f  identical %s" % (name, med[0], os.path.basename(paths[i]), med[i], med[i] / med[0],
                                                                          np.array_equal(outs[0], outs[i])), flush=True)
    # Mode R, built-in scene at 1080p (the fixture with a reference CPU time behind it)
    z = np.load(os.path.join(ROOT, "tests", "golden", "builtin_scene.npz"))
    faces, verts = np.ascontiguousarray(z["faces"].view(rt3.GFACE).reshape(-1)), np.ascontiguousarray(z["verts"])
    cam = rt3.main_camera(1920, 1080)
    outs, times = [], [[] for _ in libs]
    for r in range(6):
        for i, (L, ctx) in enumerate(libs):
            if r == 0:
                L.rt3_set_spheres(ctx, None, None, 0)
                L.rt3_set_mesh(ctx, faces.ctypes.data_as(C.c_void_p), C.c_uint32(len(faces)), verts.ctypes.data_as(C.c_void_p), C.c_uint32(len(verts)), None)
            out = np.zeros((1080, 1920), np.uint32)
            assert L.rt3_render(ctx, C.byref(cam.c), C.c_uint32(1920), C.c_uint32(1080), out.ctypes.data_as(C.c_void_p)) == 0, L.rt3_last_error(ctx)
            st = rt3.rt3_stats()
            L.rt3_get_stats(ctx, C.byref(st))
            if r > 0:
                times[i].append(st.trace_ms)
            else:
                outs.append(out)
    med = [statistics.median(t) for t in times]
    for i in range(1, len(libs)):
        print("%-36s A %.3f ms  %-28s %.3f ms  x%.4f  identical %s" % ("mode R built-in 1920x1080", med[0], os.path.basename(paths[i]), med[i], med[i] / med[0],
                                                                      np.array_equal(outs[0], outs[i])), flush=True)


if __name__ == "__main__":
    main()
