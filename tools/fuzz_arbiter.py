#!/usr/bin/env python3
"""For the given fuzz seeds: which of the two GPU kernels (matrix filter / VALU scan) differs from the CPU oracle?
    python tools/fuzz_arbiter.py seed [seed ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
rt3 = importlib.import_module("raytracer-3_amd")
import oracle_lib as O
import fuzz_filter as F

r = rt3.HipRenderer()
empty = (np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32))
for seed in map(int, sys.argv[1:]):
    cr, mats, cam, p, info = F.scene(seed)
    faces = verts = fm = None
    if seed % 2:
        spread = float(np.abs(cr[:, :3] - np.float32(info["offset"])).mean() / max(info["scale"], 1e-30))
        faces, verts, fm = F.mesh(seed, info["scale"], np.float64(info["offset"]), max(spread, 1.0))
        r.set_mesh(faces, verts, fm)
        if seed % 4 == 3:
            cr, mats = cr[:0], mats[:0]
    else:
        r.set_mesh(*empty)
    r.set_spheres(cr, mats)
    os.environ["RT3_NO_MFMA"] = "1"
    valu = r.render_path(cam.c, p).copy()
    valu_casts = r.stats().ray_casts
    del os.environ["RT3_NO_MFMA"]
    mfma = r.render_path(cam.c, p).copy()
    mfma_casts = r.stats().ray_casts
    op = O.make_params(p.width, p.height, spp=p.spp, max_depth=p.max_depth, seed=p.seed, flags=p.flags, lens_radius=p.lens_radius, t_min=p.t_min)
    kw = {}
    if len(cr):
        kw.update(spheres=cr, smats=np.ascontiguousarray(mats).view(O.MATERIAL))
    if faces is not None:
        kw.update(faces=np.ascontiguousarray(faces).view(O.GFACE), verts=verts, fmats=np.ascontiguousarray(fm).view(O.MATERIAL))
    want, casts = O.render_path(O.copy_camera(cam.c), op, threads=16, **kw)
    print("seed %d: ray casts oracle %d matrix %d VALU %d (samples %d)" % (seed, casts, mfma_casts, valu_casts, p.width * p.height * p.spp))
    print("seed %d: matrix kernel differs from the oracle in %d pixels, VALU kernel in %d, the two from each other in %d  %r" % (
        seed, int((mfma != want).sum()), int((valu != want).sum()), int((mfma != valu).sum()), info), flush=True)
