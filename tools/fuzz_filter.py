#!/usr/bin/env python3
"""Differential fuzz on the GPU: random sphere scenes (counts on both sides of the 512-sphere kernel limit, radii over five decades,
scenes far from the origin, cameras inside spheres, overlapping and nested spheres, all materials) rendered with the matrix-filter
kernels and with the VALU-scan kernels of the same library; any differing pixel is a lost or invented candidate.
    python tools/fuzz_filter.py [scenes] [first seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rt3 = importlib.import_module("raytracer-3_amd")

def scene(seed):
    rng = np.random.RandomState(seed)
    n = int(rng.choice([1, 7, 33, 100, 300, 488, 512, 513, 700, 1500, 4000]))
    scale = float(10.0 ** rng.uniform(-3, 3))
    offset = rng.choice([0.0, 0.0, 10.0, 1e3, 3e4]) * rng.uniform(-1, 1, 3) * scale
    cr = np.zeros((n, 4), np.float64)
    spread = rng.choice([2.0, 8.0, 30.0])
    cr[:, :3] = rng.normal(0, spread, (n, 3)) * np.float64([1.0, rng.choice([0.2, 1.0]), 1.0])
    cr[:, 3] = 10.0 ** rng.uniform(-2.5, 0.5, n) * rng.choice([1.0, 1.0, 0.1, 5.0], n)
    if rng.rand() < 0.4:
        cr[0] = (0.0, -1000.0 - 0.5 * spread, 0.0, 1000.0)          # a ground sphere
    if rng.rand() < 0.3:
        cr[min(1, n - 1)] = (0.0, 0.0, 0.0, 3.0 * spread)            # one sphere around most of the scene (camera may be inside)
    cr[:, :3] = cr[:, :3] * scale + offset
    cr[:, 3] *= scale
    mats = np.zeros(n, rt3.MATERIAL)
    mats["kind"] = rng.randint(0, 4, n)
    mats["rgb"] = rng.uniform(0.2, 1.0, (n, 3))
    mats["param"] = np.where(mats["kind"] == 3, rng.choice([1.5, 1.0 / 1.5, 2.4], n), rng.uniform(0.0, 0.6, n) * (rng.rand(n) < 0.7))
    eye = (rng.normal(0, spread, 3) * np.float64([1.0, 0.3, 1.0]) + np.float64([0, 0.3 * spread, 0])) * scale + offset
    at = rng.normal(0, 0.3 * spread, 3) * scale + offset
    w, h = 192, 108
    cam = rt3.Camera().look_at(w, h, tuple(eye), tuple(at), (0.0, 1.0, 0.0), float(rng.uniform(20, 90)), 1.0)
    p = rt3.make_params(w, h, spp=int(rng.choice([1, 4, 9])), max_depth=int(rng.choice([2, 8, 30])), seed=seed + 1, flags=int(rng.choice([0, 1, 3])),
                        lens_radius=float(rng.choice([0.0, 0.0, 0.02])) * scale, t_min=float(0.001 * scale))
    return cr.astype(np.float32), mats, cam, p, dict(n=n, scale=scale, offset=[float(v) for v in offset])

def mesh(seed, scale, offset, spread):
    """A triangle soup in the same region: ordinary, needle-thin, tiny and huge faces; every other seed gets one."""
    rng = np.random.RandomState(7000 + seed)
    n = int(rng.choice([1, 12, 100, 600, 2500]))
    c = rng.normal(0, spread, (n, 1, 3))
    size = (10.0 ** rng.uniform(-2.5, 0.7, (n, 1, 1))) * rng.choice([1.0, 1.0, 10.0], (n, 1, 1))
    tri = c + rng.normal(0, 1, (n, 3, 3)) * size
    thin = rng.rand(n) < 0.15
    tri[thin, 2] = tri[thin, 0] + (tri[thin, 1] - tri[thin, 0]) * rng.uniform(0.3, 0.7, (thin.sum(), 1)) + rng.normal(0, 1e-4, (thin.sum(), 3)) * size[thin, 0]
    verts = np.zeros((3 * n, 4), np.float32)
    verts[:, :3] = (tri.reshape(-1, 3) * scale + offset).astype(np.float32)
    faces = np.zeros(n, rt3.GFACE)
    faces["v1"], faces["v2"], faces["v3"] = 3 * np.arange(n), 3 * np.arange(n) + 1, 3 * np.arange(n) + 2
    v = verts[:, :3].reshape(n, 3, 3)
    nrm = np.cross(v[:, 2] - v[:, 0], v[:, 1] - v[:, 0]).astype(np.float32)
    ln = np.sqrt((nrm * nrm).sum(1, dtype=np.float32))
    faces["normal"] = np.where(ln[:, None] > 0, nrm / np.maximum(ln, np.float32(1e-30))[:, None], np.float32([0, 0, 1]))
    if rng.rand() < 0.3:                                             # stored normals that are neither unit nor perpendicular
        odd = rng.rand(n) < 0.3
        faces["normal"][odd] = (faces["normal"][odd] + rng.normal(0, 0.7, (odd.sum(), 3))) * rng.uniform(0.2, 3.0, (odd.sum(), 1))
    faces["color"] = rng.uniform(0.1, 1.0, (n, 3))
    fm = np.zeros(n, rt3.MATERIAL)
    fm["kind"] = rng.randint(0, 4, n)
    fm["rgb"] = faces["color"]
    fm["param"] = np.where(fm["kind"] == 3, 1.5, rng.uniform(0.0, 0.5, n))
    return faces, verts, fm

def run(count, first, r=None, log=print):
    r = r or rt3.HipRenderer()
    bad_total = 0
    empty = (np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32))
    for seed in range(first, first + count):
        cr, mats, cam, p, info = scene(seed)
        if seed % 2:
            spread = float(np.abs(cr[:, :3] - np.float32(info["offset"])).mean() / max(info["scale"], 1e-30))
            faces, verts, fm = mesh(seed, info["scale"], np.float64(info["offset"]), max(spread, 1.0))
            r.set_mesh(faces, verts, fm)
            info["faces"] = len(faces)
            if seed % 4 == 3:
                cr, mats = cr[:0], mats[:0]                           # faces only
        else:
            r.set_mesh(*empty)
        r.set_spheres(cr, mats)
        if seed % 2 and seed % 3 == 0:                               # Mode R as well: matrix filter against the plain brute force
            mcam = rt3.main_camera(160, 90)
            shift = np.float32(info["offset"]) + np.float32([0, 0, 3 * info["scale"] * max(spread, 1.0)])
            vr = verts.copy(); vr[:, :3] -= shift                    # the Mode-R camera sits at the origin looking down -z
            r.set_mesh(faces, vr, fm)
            r.configure(spp=None)
            r.render(mcam); a = mcam.get_frame().d().copy()
            r.force_plain_mode_r(True)
            r.render(mcam); b = mcam.get_frame().d().copy()
            r.force_plain_mode_r(False)
            if (a != b).any():
                bad_total += 1
                log("seed %d: MODE R %d pixels differ  %r" % (seed, int((a != b).sum()), info))
            r.set_mesh(faces, verts, fm)
        os.environ["RT3_NO_MFMA"] = "1"
        ref = r.render_path(cam.c, p).copy()
        del os.environ["RT3_NO_MFMA"]
        img = r.render_path(cam.c, p)
        assert r.stats().mfma_instructions > 0
        bad = int((img != ref).sum())
        if bad:
            bad_total += 1
            log("seed %d: %d pixels differ  %r" % (seed, bad, info))
        if seed % 7 in (0, 1, 2):                                    # k_trace_levels (the kernel of scenes > 112 000 primitives) in three of its forms
            os.environ["RT3_LEVELS"] = "4" if seed % 7 != 2 else "3"
            if seed % 7 == 1:
                os.environ["RT3_NO_RESIDENT"] = "1"
            lev = r.render_path(cam.c, p)
            del os.environ["RT3_LEVELS"]
            os.environ.pop("RT3_NO_RESIDENT", None)
            if (lev != ref).any():
                bad_total += 1
                log("seed %d: K_TRACE_LEVELS (seed %% 7 = %d) %d pixels differ  %r" % (seed, seed % 7, int((lev != ref).sum()), info))
        if seed % 5 == 0:                                            # the nested form with its rows streamed through LDS tiles
            os.environ["RT3_NO_RESIDENT"] = "1"
            tiled = r.render_path(cam.c, p)
            del os.environ["RT3_NO_RESIDENT"]
            if (tiled != ref).any():
                bad_total += 1
                log("seed %d: TILED ROWS %d pixels differ  %r" % (seed, int((tiled != ref).sum()), info))
        if (seed - first) % 50 == 49:
            log("... %d scenes, %d with differences" % (seed - first + 1, bad_total))
    log("fuzz: %d scenes, %d with differences" % (count, bad_total))
    r.set_mesh(*empty)
    return bad_total


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    return 1 if run(count, first, log=lambda m: print(m, flush=True)) else 0

if __name__ == "__main__":
    sys.exit(main())
