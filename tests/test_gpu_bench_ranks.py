"""bench.py's multi-rank control flow with REAL HIP tiles, on the one GPU of the box: two fresh worker processes
under torch.distributed.run both render their interleaved-row shard of the bench workload (reduced spp) on device 0 through the C ABI and
gather the rows on rank 0 — over gloo through host memory, because RCCL refuses two ranks on one device; that staging is the only thing a
real N-GPU run does differently (`--rehearse-on-one-gpu`).  The assembled frame must be the one-rank frame byte for byte and rank 0's JSON
line must be well-formed with n_gpus = 2 and the one-rank ray-cast count (the all_reduce of tests / casts / elapsed)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(n, ppm, spp=8):
    common = ["bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--spp", str(spp), "--cpu-seconds", "0", "--no-extra", "--save-ppm", str(ppm)]
    if n == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + common + ["--rehearse-on-one-gpu"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                                  # ONE line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_on_one_gpu_render_their_shards_and_gather_the_one_rank_frame(tmp_path, ranks):
    one = run_bench(1, tmp_path / "one.ppm")
    many = run_bench(ranks, tmp_path / "many.ppm")
    assert (tmp_path / "many.ppm").read_bytes() == (tmp_path / "one.ppm").read_bytes()
    assert many["n_gpus"] == ranks and one["n_gpus"] == 1 and many["steps"] == 2 and many["warmup"] == 1
    assert many["ray_casts"] == one["ray_casts"] and many["prim_tests"] == one["prim_tests"]       # summed over the ranks == the whole frame
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in many
    assert many["value"] > 0 and many["scaling"] == "strong" and "REHEARSAL" in many["config"]["sharding"]
    assert "cpu_baseline" not in many                                         # rank 0 at N = 1 only


def test_rccl_gather_path_with_one_rank(tmp_path):
    """RT3_BENCH_FORCE_DIST=1: one rank under torch.distributed.run with backend "nccl" (RCCL) — process-group initialisation with device_id,
    dist.gather of the tile on the render's stream, the indexed copy into the frame, the all_reduce of the statistics: the code an N-GPU run executes,
    with N = 1 (two ranks cannot share the one GPU of the box under RCCL).  Same frame as the native world-1 path (rt3_gather_rows)."""
    one = run_bench(1, tmp_path / "native.ppm")
    common = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--spp", "8", "--cpu-seconds", "0", "--no-extra", "--save-ppm", str(tmp_path / "rccl.ppm")]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + common
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", RT3_BENCH_FORCE_DIST="1")
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert (tmp_path / "rccl.ppm").read_bytes() == (tmp_path / "native.ppm").read_bytes()
    assert d["n_gpus"] == 1 and "RCCL gather" in d["config"]["sharding"] and d["ray_casts"] == one["ray_casts"]
