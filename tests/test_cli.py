"""The rt3 command line (raytracer-3_amd/host/Main.cpp) keeps the reference's CLI contract (src/Main.cpp:89-239, :251-254):
flags, value forms, messages, exit codes — and renders through the same call sequence as the reference's main()."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer-3_amd", "rt3")


def run(*args, cwd=None):
    p = subprocess.run([EXE, *args], capture_output=True, text=True, cwd=cwd)
    rc = p.returncode if p.returncode < 128 else p.returncode - 256      # main returns -1 -> 255
    return rc, p.stdout, p.stderr


def test_binary_exists():
    assert os.path.exists(EXE), "build it with __graft_entry__.build()"


def test_help_exits_zero_and_lists_reference_options():
    rc, out, err = run("-h")
    assert rc == 0
    assert out.startswith("Usage: %s [<options>] <output_path>" % EXE)
    for frag in ("-f,--format", "(default: png)", "-W,--width", "(default: 800)", "-H,--height", "(default: 600)", "-h,--help"):
        assert frag in out
    assert run("--help")[0] == 0


# parse_cli's error table (src/Main.cpp:132,146,167-176,210,233); tests/test_sanitizers.py runs the same rows through the ASan/UBSan build
USAGE_ERRORS = [
    ((), "No output path given."),
    (("-W",), "-W has no value."),
    (("--width", "-H", "3", "o.ppm"), "--width has no value."),
    (("-f", "jpg", "o"), "Unknown output format 'jpg'"),
    (("-W", "abc", "o"), "Invalid width 'abc'"),
    (("-H", "99999999999999999999", "o"), "Height too large '99999999999999999999'"),
    (("-W", "4294967296", "o"), "Width too large '4294967296'"),
    (("--bogus", "o"), "Unknown option '--bogus'"),
]


@pytest.mark.parametrize("args,msg", USAGE_ERRORS)
def test_usage_errors_return_minus_one(args, msg):
    rc, out, err = run(*args)
    assert rc == -1 and msg in err
    if "Unknown option" in msg:
        assert "Run '%s -h' to see a list of valid options." % EXE in err


def decode_png(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 6)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.gpu
def test_mode_r_through_the_cli_matches_the_oracle(tmp_path, oracle):
    """`rt3 -f ppm -W 160 -H 90 out.ppm` with cwd holding bin/objects/teddy.obj — the reference's own invocation."""
    from test_host_scene import write_obj
    objdir = tmp_path / "bin" / "objects"
    objdir.mkdir(parents=True)
    write_obj(str(objdir / "teddy.obj"))
    for flags in (("-f", "ppm", "-W", "160", "-H", "90"), ("-fppm", "-W160", "--height", "90"), ("--format=ppm", "--width=160", "--height=90")):
        rc, out, err = run(*flags, "out.ppm", "ignored_second_positional", cwd=str(tmp_path))
        assert rc == 0, err
        obj = oracle.prerender_object(str(objdir / "teddy.obj"), (0.0, 0.0, -3.0), np.float32(1.0) / np.float32(17.0), (1.0, 0.0, 0.0))
        sph = oracle.prerender_sphere((-2.0, 0.0, -5.0), 1.0, 8, 8, (0.0, 0.0, 1.0))
        faces, verts = oracle.merge([obj, sph])
        ref = oracle.render_mode_r(faces, verts, oracle.camera_update(160, 90), 160, 90)
        assert (tmp_path / "out.ppm").read_bytes() == oracle.ppm_bytes(ref)
    # tessellating the sphere on the device gives the same file
    rc, out, err = run("-f", "ppm", "-W", "160", "-H", "90", "--gpu-prerender", "gpu.ppm", cwd=str(tmp_path))
    assert rc == 0, err
    assert (tmp_path / "gpu.ppm").read_bytes() == oracle.ppm_bytes(ref)
    # default format is png (Main.cpp:76): same pixels, alpha 255
    rc, out, err = run("-W", "160", "-H", "90", "out.png", cwd=str(tmp_path))
    assert rc == 0, err
    rgba = decode_png((tmp_path / "out.png").read_bytes())
    want = np.stack([(ref >> 24) & 255, (ref >> 16) & 255, (ref >> 8) & 255, np.full_like(ref, 255)], axis=-1).astype(np.uint8)
    assert np.array_equal(rgba, want)
    # a missing scene file is fatal: message + exit -1 (Main.cpp:305-308)
    rc, out, err = run("-f", "ppm", "x.ppm", cwd=str(tmp_path / "bin"))
    assert rc == -1 and "fatal" in err


@pytest.mark.gpu
def test_mode_x_through_the_cli_matches_the_oracle(tmp_path, rt3, oracle):
    rc, out, err = run("-f", "ppm", "-W", "96", "-H", "54", "--scene", "three", "--spp", "4", "--depth", "5", "--seed", "9", "--gpus", "1",
                       str(tmp_path / "three.ppm"))
    assert rc == 0, err
    from cases import oracle_render
    cr, mats = rt3.scene_three_spheres()
    cam = rt3.Camera().update(96, 54, 1.0, np.float32(96) / np.float32(54) * np.float32(2.0), 2.0)
    case = dict(spheres=cr, smats=mats, cam=cam.c, params=dict(width=96, height=54, spp=4, max_depth=5, seed=9, flags=1))
    ref, _ = oracle_render(case)
    assert (tmp_path / "three.ppm").read_bytes() == oracle.ppm_bytes(ref)
