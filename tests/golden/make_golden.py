#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Run in the build container (needs /root/reference for teddy.obj — read as DATA,
nothing from the reference is executed):

    python tests/golden/make_golden.py

What is stored
  builtin_scene.npz   the flattened built-in scene of src/Main.cpp:280-283 (teddy.obj scaled 1/17 at (0,0,-3) +
                      8x8 sphere at (-2,0,-5)) as GFace[] / vec4[] arrays — an INPUT fixture, so that the GPU box
                      (which has no /root/reference) can render the scene whose PPM hashes SURVEY.md records.
  mode_r_small.npz    64x36 Mode-R images of the three commented-out scenes of Main.cpp:277-279, rendered by the
                      oracle AFTER it reproduced the reference hashes (reference_pins.json).
  mode_x_small.npz    small Mode-X images rendered by the oracle (regression fixtures: the reference has no Mode X).
reference_pins.json holds the reference outputs recorded by the survey; it is
written by hand, not by this script.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from cases import TEDDY, mode_x_cases, oracle_render  # noqa: E402


def builtin_scene():
    teddy = O.prerender_object(TEDDY, (0.0, 0.0, -3.0), np.float32(1.0) / np.float32(17.0), (1.0, 0.0, 0.0))
    sph = O.prerender_sphere((-2.0, 0.0, -5.0), 1.0, 8, 8, (0.0, 0.0, 1.0))
    return O.merge([teddy, sph]), teddy, sph


def main():
    pins = json.load(open(os.path.join(HERE, "reference_pins.json")))
    (faces, verts), teddy, sph = builtin_scene()
    # the oracle must reproduce the reference before anything it renders is stored
    for key, sha in pins["ppm_sha256"].items():
        w, h = map(int, key.split("x"))
        img = O.render_mode_r(faces, verts, O.camera_update(w, h), w, h, 0, h - 1)
        got = O.sha256(O.ppm_bytes(img))
        assert got == sha, "oracle does not reproduce the reference at %s: %s" % (key, got)
        print("pinned", key, got)
    np.savez_compressed(os.path.join(HERE, "builtin_scene.npz"), faces=faces.view(np.uint8), verts=verts)

    small = {}
    tri = O.prerender_triangle((1.0, 0.0, -3.0), (-1.0, 0.0, -3.0), (0.0, 1.0, -3.0), (1.0, 0.0, 0.0))   # Main.cpp:279
    one_sphere = O.prerender_sphere((0.0, 0.0, -3.0), 1.0, 8, 8, (1.0, 0.0, 0.0))                      # Main.cpp:278
    for name, (f, v) in (("triangle", tri), ("sphere8x8", one_sphere), ("teddy", teddy)):               # Main.cpp:277
        small[name] = O.render_mode_r(f, v, O.camera_update(64, 36), 64, 36)
    np.savez_compressed(os.path.join(HERE, "mode_r_small.npz"), **small)

    modex = {name: oracle_render(case)[0] for name, case in mode_x_cases().items()}
    np.savez_compressed(os.path.join(HERE, "mode_x_small.npz"), **modex)
    for n in ("builtin_scene.npz", "mode_r_small.npz", "mode_x_small.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
