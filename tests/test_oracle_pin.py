"""Pins the CPU oracle to outputs of the REFERENCE itself (copied into golden/reference_pins.json):
PPM SHA-256 of the built-in scene at 400x225 and 1920x1080, two known pixels, hash/float known answers."""
import json
import os

import numpy as np
import pytest

from cases import GOLDEN, TEDDY, load_builtin_scene

PINS = json.load(open(os.path.join(GOLDEN, "reference_pins.json")))


def test_builtin_scene_fixture_matches_reference_asset(oracle):
    """The committed fixture is what the oracle's own pre-render makes from the reference's teddy.obj."""
    if not os.path.exists(TEDDY):
        pytest.skip("/root/reference not present (GPU box)")
    teddy = oracle.prerender_object(TEDDY, (0.0, 0.0, -3.0), np.float32(1.0) / np.float32(17.0), (1.0, 0.0, 0.0))
    sph = oracle.prerender_sphere((-2.0, 0.0, -5.0), 1.0, 8, 8, (0.0, 0.0, 1.0))
    faces, verts = oracle.merge([teddy, sph])
    gf, gv = load_builtin_scene()
    assert len(faces) == PINS["scene"]["faces"] and len(verts) == PINS["scene"]["vertices"]
    assert faces.tobytes() == gf.tobytes() and verts.tobytes() == gv.tobytes()


@pytest.mark.parametrize("size", ["400x225", "1920x1080"])
def test_oracle_reproduces_reference_ppm(oracle, size):
    w, h = map(int, size.split("x"))
    faces, verts = load_builtin_scene()
    img = oracle.render_mode_r(faces, verts, oracle.camera_update(w, h), w, h, 0, h - 1)   # reference loop: rows 0..h-2
    ppm = oracle.ppm_bytes(img)
    if size == "400x225":
        assert len(ppm) == PINS["ppm_size_400x225"]
        for key, rgb in PINS["pixels_400x225"].items():
            x, y = map(int, key.split(","))
            px = int(img[y, x])
            assert [(px >> 24) & 255, (px >> 16) & 255, (px >> 8) & 255] == rgb
        assert ppm.startswith(b"P6\n# Image rendered by the RayTracer-3\n400 225\n255\n")
    assert oracle.sha256(ppm) == PINS["ppm_sha256"][size]


def test_hash_known_answers(oracle, rt3):
    for key, want in PINS["hash_kat"].items():
        assert oracle.lib().oracle_hash_u32(int(key)) == int(want, 16)
        assert rt3.lib().rt3_hash_u32(int(key)) == int(want, 16)
    for key, want in PINS["random_kat"].items():
        h = oracle.lib().oracle_hash_u32(int(key))
        assert abs(oracle.lib().oracle_random_float(h) - want) < 1e-9
        assert rt3.lib().rt3_random_float(h) == oracle.lib().oracle_random_float(h)


def test_sky_and_pack_known_answers(oracle):
    r, g, b = oracle.sky((0.0, 1.0, 0.0))
    assert (r, g, b) == (0.5, np.float32(0.7), 1.0)
    px = oracle.pack_pixel(r, g, b)
    assert [(px >> 24) & 255, (px >> 16) & 255, (px >> 8) & 255, px & 255] == PINS["sky_up_bytes"] + [255]
    # rounding is half away from zero, with clamping (func_packing.inl:67-83)
    assert (oracle.pack_pixel(0.5, 0.7, 2.0) >> 8) == (128 << 16 | 179 << 8 | 255)
    assert oracle.pack_pixel(-3.0, 0.0, 1.0) == (0xFF | 255 << 8)
