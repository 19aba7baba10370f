#!/usr/bin/env python3
"""Renders BASELINE.json configs[1] — the headline workload, 1920x1080, 512 spp, depth 50 — COMPLETELY with the CPU oracle
(about 10^9 samples: minutes on 8-16 cores) and stores what the GPU test compares against:

    tests/golden/config2_full.json   SHA-256 of the frame (uint32 pixels, row-major, little-endian), the oracle's ray-cast
                                     count and one CRC-32 per row (to localise a mismatch)

    python tests/golden/make_config2_golden.py [threads]
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
import oracle_lib as O  # noqa: E402
from cases import oracle_render, rt3  # noqa: E402   (rt3: the host-side scene builders of librt3hip.so, no GPU needed)

W, H, SPP, DEPTH, SCENE_SEED, RENDER_SEED = 1920, 1080, 512, 50, 42, 1


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    cr, mats = rt3.scene_weekend(SCENE_SEED)
    cam = rt3.weekend_camera(W, H)
    case = dict(spheres=cr, smats=mats, cam=cam.c,
                params=dict(width=W, height=H, spp=SPP, max_depth=DEPTH, seed=RENDER_SEED, flags=O.FLAG_GAMMA2, lens_radius=0.05))
    t0 = time.time()
    img, casts = oracle_render(case, threads=threads)
    img = np.ascontiguousarray(img, dtype="<u4")
    out = {
        "_comment": "CPU oracle (oracle/rt3_oracle.c) render of BASELINE.json configs[1]; regenerate with tests/golden/make_config2_golden.py",
        "width": W, "height": H, "spp": SPP, "max_depth": DEPTH, "scene_seed": SCENE_SEED, "seed": RENDER_SEED, "lens_radius": 0.05,
        "ray_casts": int(casts),
        "sha256": hashlib.sha256(img.tobytes()).hexdigest(),
        "row_crc32": [int(zlib.crc32(img[y].tobytes())) for y in range(H)],
        "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": threads,
    }
    json.dump(out, open(os.path.join(HERE, "config2_full.json"), "w"))
    print("config2_full.json:", out["sha256"], out["ray_casts"], "casts,", out["oracle_seconds"], "s")


if __name__ == "__main__":
    main()
