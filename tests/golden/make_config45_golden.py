#!/usr/bin/env python3
"""Renders rows of BASELINE.json configs[3] and configs[4] AT THEIR OWN SAMPLE BUDGET with the CPU oracle and stores what the GPU
tests compare against (tests/test_gpu_mode_x.py::test_config4_full_spp_rows..., ::test_config5_full_spp_rows...):

    config 4   100 000 spheres, 1920x1080, 256 spp, depth 50: rows 270 and 810 of the frame (tile_rows=1, tile_index=270, tile_count=540)
    config 5   Cornell-style box, 47 106 triangles, emissive quad, 1024x1024, 2048 spp, depth 50, black background: rows 300 and 812

    tests/golden/config45_rows.json   per config: the shard parameters, the oracle's ray-cast count, CRC-32 and SHA-256 of the rows
    tests/golden/config45_rows.npz    the rows themselves (uint32 pixels; 2 x 1920 and 2 x 1024 words) to localise a mismatch

Every sample index 0 .. spp-1 of those pixels goes through the oracle here and through the tiled matrix-filter kernels there: the
per-sample dimension of the design (raytracer_v4.glsl:197-206) at the configs' full sample budget.  Cost: about 10 min (config 4) and
about 80-100 min (config 5) on the 8 cores of the build container.

    python tests/golden/make_config45_golden.py [threads] [4|5|both]
"""
import hashlib
import json
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import oracle_render, rt3  # noqa: E402   (rt3: the host-side scene builders of librt3hip.so, no GPU needed)

JSON_PATH = os.path.join(HERE, "config45_rows.json")
NPZ_PATH = os.path.join(HERE, "config45_rows.npz")


def config4():
    cr, mats = rt3.scene_stress(100000, 43)
    W, H = 1920, 1080
    cam = rt3.Camera().look_at(W, H, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=W, height=H, spp=256, max_depth=50, seed=9, flags=1))
    return case, dict(tile_rows=1, tile_index=270, tile_count=540), "scene_stress(100000, 43), look_at((0,8,12),(0,6,-50),(0,1,0),45,1)"


def config5():
    faces, verts, fmats = rt3.scene_cornell(64)
    cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
    case = dict(faces=faces, verts=verts, fmats=fmats, cam=cam.c,
                params=dict(width=1024, height=1024, spp=2048, max_depth=50, seed=6, flags=1 | 2))
    return case, dict(tile_rows=1, tile_index=300, tile_count=512), "scene_cornell(64), Camera.update(1024,1024,2,2,2)"


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    out = json.load(open(JSON_PATH)) if os.path.exists(JSON_PATH) else {
        "_comment": "CPU oracle (oracle/rt3_oracle.c) rows of BASELINE.json configs[3] / configs[4] at their full sample budget; "
                    "regenerate with tests/golden/make_config45_golden.py"}
    rows = dict(np.load(NPZ_PATH)) if os.path.exists(NPZ_PATH) else {}
    for name, make in (("config4", config4), ("config5", config5)):
        if which not in ("both", name[-1]):
            continue
        case, shard, scene = make()
        t0 = time.time()
        img, casts = oracle_render(case, threads=threads, **shard)
        img = np.ascontiguousarray(img, dtype="<u4")
        p = case["params"]
        out[name] = {"scene": scene, "params": p, "shard": shard,
                     "frame_rows": [shard["tile_index"], shard["tile_index"] + shard["tile_count"]],
                     "ray_casts": int(casts), "sha256": hashlib.sha256(img.tobytes()).hexdigest(),
                     "row_crc32": [int(zlib.crc32(img[y].tobytes())) for y in range(img.shape[0])],
                     "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": threads}
        rows[name] = img
        json.dump(out, open(JSON_PATH, "w"), indent=1)
        np.savez_compressed(NPZ_PATH, **rows)
        print(name, out[name]["sha256"], casts, "casts,", out[name]["oracle_seconds"], "s", flush=True)


if __name__ == "__main__":
    main()
