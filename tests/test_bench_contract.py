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
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "mfma"):
        assert key in r, key
    # 8 spp is not the profiled workload: hardware counters are never guessed, so the binding unit's figures are absent and the live
    # executed-matrix fraction stands in for achieved / peak / frac (bound "mfma"); at the full workload with a fresh profile bound is "valu_busy"
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 < r["frac"] <= 1.0                                  # a roofline fraction: executed work over the peak of the unit that does it
    assert r["peak"] == 2500.0 and "achieved" in r["live"] and "traffic_source" in r and r["valu_busy"] is None
    assert r["traffic"] is None
    m = r["mfma"]
    assert m["live"] is True and m["filter_k"] == 32 and abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-3 and m["frac"] == r["frac"]
    a = d["algorithmic_equiv"]
    assert a["flop_per_test"] == 20.0 and abs(a["tflops"] - d["prim_tests"] * 20.0 / (r["kernel_ms"] * 1e-3) / 1e12) / a["tflops"] < 0.02
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"]


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    return B


def test_the_line_names_what_binds_with_one_definition(tmp_path, monkeypatch):
    """With a counter profile of the running kernel sources `roofline.bound` is the saturated unit, "valu_busy", defined as SQ_ACTIVE_INST_VALU x 4 /
    SIMD-cycles — the same expression for the headline (counters_from_profile) and for the tiled kernels (tiled_counters).  (The counter advances per
    4-cycle pass, so it prices the 64-cycle block conversion of the decode as what it occupies.)  Round 2's instruction-slot figure, (SQ_INSTS_VALU -
    SQ_INSTS_MFMA) x 4 / the same cycles, is reported beside it under its own name and never as the busy fraction."""
    B = _load_bench()
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    (tmp_path / "raytracer-3_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "raytracer-3_amd" / "csrc" / "k.hip").write_text("__global__ void k() {}\n")
    fp = B.source_fingerprint()
    one = lambda v, n=1: {"sum_over_dispatches": float(v), "dispatches": n}
    cycles_simd = 4.0e9                                            # SIMD-cycles of the launch = SQ_BUSY_CYCLES / 32 * 1024
    pmc = {"_source_fingerprint": fp, "FETCH_SIZE": one(1000), "WRITE_SIZE": one(500), "SQ_BUSY_CYCLES": one(cycles_simd / 1024 * 32),
           "SQ_INSTS_VALU": one(1.0e9), "SQ_INSTS_MFMA": one(1.0e8), "SQ_ACTIVE_INST_VALU": one(0.95e9), "SQ_VALU_MFMA_BUSY_CYCLES": one(1.6e9)}
    (tmp_path / B.PMC_PROFILE).write_text(json.dumps(pmc))
    traffic, why, valu = B.counters_from_profile(fp)
    assert traffic == int((2 * 1000 + 500) * 1024)                # FETCH_SIZE doubled (gfx950), KiB
    assert valu["frac"] == round(0.95e9 * 4 / cycles_simd, 3) == 0.95
    assert valu["mfma_pipe_busy"] == 0.4 and valu["instruction_slot_frac"] == 0.9 and valu["instruction_slot_frac"] != valu["frac"]
    assert valu["busy_simd_cycles_per_launch"] == int(3.8e9) and valu["simd_cycles_per_launch"] == int(4.0e9)
    tiled = {"_source_fingerprint": fp, "config_5": dict(pmc, _x=0)}
    tiled["config_5"].pop("_source_fingerprint")
    tiled["config_5"].pop("_x")
    (tmp_path / B.TILED_PMC_PROFILE).write_text(json.dumps(tiled))
    counters, src = B.tiled_counters(fp)
    assert counters["config_5"] == {"valu_busy": 0.95, "mfma_busy": 0.4} and "stale" not in src
    counters, src = B.tiled_counters("0" * 16)
    assert counters == {} and "stale" in src


@pytest.mark.gpu
def test_full_workload_line_carries_the_binding_unit_and_the_extra_workloads(tmp_path):
    """The default command (what the driver runs) at reduced steps: extra_workloads are there with the two-level filter's executed / equivalent counts,
    and `roofline` follows the fingerprint rule — "valu_busy" with counter figures when profiles/r03_bench_pmc_k_trace.json belongs to this build, the
    live matrix figures otherwise."""
    p = run_bench("--steps", "1", "--warmup", "1", "--cpu-seconds", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    r = d["roofline"]
    B = _load_bench()
    fresh = B.counters_from_profile(B.source_fingerprint())[2] is not None
    assert r["bound"] == ("valu_busy" if fresh else "mfma")
    if fresh:
        assert r["frac"] == r["valu_busy"]["frac"] and 0.5 < r["frac"] <= 1.05 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 and r["traffic"] > 1e10
    assert r["mfma"]["live"] is True and 0.2 < r["mfma"]["frac"] < 0.6
    ew = {w["workload"].split(":")[0]: w for w in d["extra_workloads"]}
    assert set(ew) == {"Mode R", "config 4", "config 5"}
    for name in ("config 4", "config 5"):
        w = ew[name]
        f = w["filter"]
        assert f["levels"] == 3 and f["filter_tests_executed"] * 50 < w["prim_tests"] and w["tests_per_s"] > 5e13       # brute-force equivalent
        assert "valu_busy" in w and "mfma_busy" in w and "counters_source" in w
    assert ew["config 5"]["filter"]["bound_tests_per_cast"] > 0 and ew["config 4"]["exact_tests_per_cast"] > 100
