"""The HOST side under AddressSanitizer + UndefinedBehaviorSanitizer ("Race detection / sanitizers"; the reference's
analogue is the Vulkan validation layer it routes into its logger, src/lib/compute/Instance.cpp:29-60).

`make -C raytracer-3_amd asan` compiles csrc/rt3_host.cpp, host/HostApi.cpp, host/sceneparser/SceneParser.cpp and host/Main.cpp with
g++ -fsanitize=address,undefined -fno-sanitize-recover=all, oracle/rt3_oracle.c with gcc and the same flags, and links them with "no device"
stubs for the device half of the C ABI (tools/asan/) — no HIP anywhere, so it runs in the build container.  Any report aborts the process:
a passing test is a clean run.  The GPU kernels cannot run under a sanitizer on this pool (GPU ASan is refused); their memory safety rests on
the host-side shape checks in rt3_device.hip and on the parity tests."""
import os
import subprocess

import pytest

from test_cli import EXE, USAGE_ERRORS, run
from test_sceneparser import EXPRESSIONS, SAMPLE, SCENE_ERRORS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = os.path.join(ROOT, "raytracer-3_amd", "rt3_asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")


@pytest.fixture(scope="module")
def asan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "raytracer-3_amd"), "asan"])
    return ASAN


def run_asan(exe, *args, cwd=None):
    p = subprocess.run([exe, *args], capture_output=True, text=True, cwd=cwd, env=ENV)
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr and p.returncode not in (98, 99), p.stderr[-4000:]
    rc = p.returncode if p.returncode < 128 else p.returncode - 256
    return rc, p.stdout, p.stderr


def test_host_scene_api_and_oracle_are_clean(asan, tmp_path):
    """OBJ loader (good / missing / unparsable / out-of-range files), tessellator, merge, cameras, PPM writer and scene builders with exact
    and short buffers — each against the oracle's restatement byte for byte — and the oracle's own Mode-R / Mode-X loops."""
    rc, out, err = run_asan(asan, "selftest", str(tmp_path))
    assert rc == 0 and "selftest: ok" in out, out + err


def test_cli_error_table_is_clean_and_says_what_the_product_says(asan):
    for args, msg in USAGE_ERRORS:
        rc, out, err = run_asan(asan, "cli", *args)
        assert rc == -1 and msg in err
        prc, pout, perr = run(*args)
        assert (prc, pout) == (rc, out) and perr.replace(EXE, "X") == err.replace(asan, "X")
    rc, out, err = run_asan(asan, "cli", "-h")
    assert rc == 0 and "-f,--format" in out
    # past parse_cli the sanitizer build has no device: the backend's error convention (fatal -> exit -1, src/Main.cpp:305-308)
    rc, out, err = run_asan(asan, "cli", "-f", "ppm", "-W", "32", "-H", "18", "--scene", "three", "--spp", "2", "out.ppm")
    assert rc == -1 and "fatal" in err


def test_every_sceneparser_input_is_clean(asan, tmp_path):
    (tmp_path / "mesh.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3\nf 2 4 3\n")
    (tmp_path / "common.scene").write_text('data { extern .obj quad: "%s"; }\nglobal { vec3 where: 0 0 -5; }\n' % (tmp_path / "mesh.obj"))
    good = [SAMPLE, EXPRESSIONS,
            '#include "common.scene"\nentities { object q { center: global.where; scale: 2.0; data: .obj quad; color: 1 1 0; } }']
    texts = [(t, None) for t in good] + list(SCENE_ERRORS)
    # more malformed input than the product's tests hold: truncations of the sample at every 40th byte, binary noise, deep nesting
    texts += [(SAMPLE[:n], "") for n in range(0, len(SAMPLE), 40)]
    texts += [("entities { sphere s { center: " + "(" * 60 + "1" + ")" * 60 + " 0 0; radius: 1; } }", None),
              ("entities { sphere s { center: " + "(" * 5000 + "1 0 0; radius: 1; } }", ""),
              ("\x00\xff\xfe entities { \x01 }", ""), ("global { float x: 1 / 0; int y: 1 % 0; int z: 7 / 0; } entities { }", ""),
              ("global { uint u: 4294967295 + 1; int i: -2147483648 - 1; float f: 1e38 * 1e38; } entities { }", ""),
              ('#include "self.scene"\n', ""), ("data { .obj t { v 0 0\n f 1 2 3 } }", ""), ("data { .obj t { v 0 0 0\n v 1 0 0\n v 0 1 0\n f 1 2 9 } } "
               "entities { object o { center: 0 0 0; scale: 1; data: .obj t; color: 1 1 1; } }", "")]
    (tmp_path / "self.scene").write_text('#include "self.scene"\n')
    for i, (text, msg) in enumerate(texts):
        p = tmp_path / ("s%d.scene" % i if text != '#include "self.scene"\n' else "self.scene")
        p.write_bytes(text.encode("latin-1"))
        rc, out, err = run_asan(asan, "cli", "--scene", str(p), "--dump-scene", cwd=str(tmp_path))
        if msg is None:
            assert rc == 0, err
        else:
            assert rc in (0, -1) and (rc == 0 or msg in err), (text, err)
        prc, pout, perr = run("--scene", str(p), "--dump-scene", cwd=str(tmp_path))
        assert (prc, pout) == (rc, out)
