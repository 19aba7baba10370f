"""bench.py's output contract: one JSON line with the fields the driver reads, plus the `roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = run_bench("--steps", "1", "--warmup", "0", "--cpu-seconds", "0")
    assert p.returncode != 0
    assert "needs a GPU" in (p.stdout + p.stderr)              # no CPU fallback: the product path fails loudly


def test_bench_refuses_a_world_size_that_does_not_match():
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stdout + p.stderr)


def test_counter_provenance_follows_the_kernel_code_not_its_comments(tmp_path, monkeypatch):
    """bench.py reports hardware-counter figures from profiles/ only for the kernel code they were collected on: the fingerprint must
    change with the code and stay put when a comment or the layout changes; a stale profile yields nulls with the reason."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    src = tmp_path / "raytracer-3_amd" / "csrc"
    src.mkdir(parents=True)
    (src / "k.hip").write_text('__global__ void k(float* p) { p[0] = 1.0f; /* one */ asm("s_nop 0 // not a comment"); }\n')
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    a = B.source_fingerprint()
    (src / "k.hip").write_text('// a new comment\n__global__ void k(float* p)\n{\n    p[0] = 1.0f;   /* two */\n    asm("s_nop 0 // not a comment");\n}\n')
    assert B.source_fingerprint() == a
    (src / "k.hip").write_text('__global__ void k(float* p) { p[0] = 2.0f; asm("s_nop 0 // not a comment"); }\n')
    assert B.source_fingerprint() != a
    (src / "k.hip").write_text('__global__ void k(float* p) { p[0] = 1.0f; asm("s_nop 1 // not a comment"); }\n')
    assert B.source_fingerprint() != a                          # string literals (inline asm) are code
    (tmp_path / "profiles").mkdir()
    (tmp_path / B.PMC_PROFILE).write_text(json.dumps({"_source_fingerprint": "0123456789abcdef"}))
    traffic, why, valu = B.counters_from_profile(B.source_fingerprint())
    assert traffic is None and valu is None and "stale" in why


@pytest.mark.gpu
def test_bench_line_has_every_field_of_the_contract():
    p = run_bench("--spp", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                         # ONE JSON line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1920 * 1080 * 8 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 < r["frac"] <= 1.0                                  # a roofline fraction: executed work over the peak of the unit that does it
    assert r["peak"] == 2500.0 and "achieved" in r["live"] and "traffic_source" in r
    assert r["traffic"] is None                                    # not the profiled workload (8 spp): counters are never guessed
    a = d["algorithmic_equiv"]
    assert a["flop_per_test"] == 20.0 and abs(a["tflops"] - d["prim_tests"] * 20.0 / (r["kernel_ms"] * 1e-3) / 1e12) / a["tflops"] < 0.02
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"]
