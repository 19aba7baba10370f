#!/usr/bin/env python3
"""bench.py — Msamples/s of the render hot path on MI355X.

Workload (BASELINE.json configs[1]): the book's final random-spheres scene (484 spheres, Lambertian / metal /
dielectric), 1920x1080, 512 spp, depth 50, thin-lens camera, gamma 2.  One "step" = one full render of the frame
(Mode X, rt3_render_path_device) with scene and camera already resident in HBM; the frame stays in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU): the same frame is sharded in interleaved rows
(rt3_params.tile_*), every rank renders its rows, then ONE RCCL gather brings the packed RGBA8 rows to rank 0, which
puts them at their frame rows with one indexed copy.  Total work is fixed, so scaling = "strong".
N = 1: the rows go through rt3_gather_rows (the C ABI's device-to-device gather), no collective.

Prints one JSON line (rank 0).  Beside the driver's fields:
  roofline         the dominant kernel (k_trace_mfma32) against the unit it saturates.  `bound` names that unit: "valu_busy" — the vector ALU:
                   SQ_ACTIVE_INST_VALU x 4 cycles over the SIMD-cycles of the launch, ONE definition everywhere (README, DESIGN, profiles/README.md,
                   extra_workloads).  The counter advances once per issued vector instruction and once more per further 4-cycle pass of a multi-pass
                   one (measured: 17.8 per v_cvt_scalef32_2xpk16_fp6_f32, the 64-cycle block conversion that decodes 32 results; 1.2 per MFMA), so it
                   is the time the unit is occupied whatever the instruction mix; round 2's instruction-slot figure ((SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4)
                   sits beside it as `instruction_slot_frac` and no longer says how busy the unit is.  achieved / peak / frac come from the
                   committed rocprofv3 --pmc passes of this same command (only rocprofv3 can collect them), gated by the fingerprint of the kernel
                   sources they were collected on.  `mfma` beside it is measured LIVE in this run: executed bf16 matrix FLOP
                   (v_mfma_f32_16x16x32_bf16 instructions counted by the kernel itself x 16384) / kernel time from HIP events on the launch
                   stream, over the 2.5 PFLOP/s dense bf16 peak.  With a stale or missing profile `bound` falls back to "mfma" and the top-level
                   achieved / peak / frac are the live matrix figures (the counters are then null, never reused on other code)
  algorithmic_equiv  SURVEY.md §8d's per-test figure (20 f32 FLOP per ray-sphere test) priced against the f32 peak — the
                   work a scalar formulation would do; NOT a roofline fraction of this kernel (it exceeds 1)
  cpu_baseline     the CPU oracle timed on this host's cores on a bounded sample of the same workload
  extra_workloads  short runs of the other BASELINE.json configs and of the Mode-R fixture (the only workload with a
                   reference CPU time behind it), each with its own executed-MFMA fraction
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: must be set before HIP initialises (RCCL)

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 512, 50
SCENE_SEED, RENDER_SEED = 42, 1
TILE_ROWS = 1                        # single-row interleave: 1080 rows split exactly evenly over 2, 4 or 8 ranks
FLOP_PER_SPHERE_TEST = 20.0          # 3 sub, 6 (b), 7 (c), 4 (D); hit-only sqrt/divide excluded
FLOP_PER_MFMA = 32768.0              # v_mfma_f32_32x32x16_bf16: 32 x 32 x 16 multiply-adds
PEAK_FP32_VALU_TFLOPS = 157.3        # MI355X_MICROARCH.md:41
PEAK_HBM_GBS = 8000.0                # MI355X_MICROARCH.md:36
PEAK_BF16_MFMA_TFLOPS = 2500.0       # dense, MI355X_MICROARCH.md:43
PMC_PROFILE = os.path.join("profiles", "r03_bench_pmc_k_trace.json")
TILED_PMC_PROFILE = os.path.join("profiles", "r03_tiled_pmc.json")   # configs 4 / 5 / Mode R (tools/profile_tiled.sh)
MODE_R_REFERENCE_CPU_S = 60.2        # SequentialRenderer (the reference's own CPU backend), built-in scene at 1920x1080:
                                     # SURVEY.md §6/§8d, measured by the survey on an 8-vCPU Xeon 2.1 GHz, 1 thread, -O2


def source_fingerprint():
    """SHA-256 (first 16 hex digits) of the kernel sources with comments and white space removed: a committed profile only speaks for
    the code it was made from (and keeps speaking for it when a comment changes)."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer-3_amd", "csrc")
    strip = re.compile(r'//[^\n]*|/\*.*?\*/|("(?:\\.|[^"\\])*")', re.S)        # comments go, string literals stay
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            text = open(os.path.join(d, name), "r", encoding="utf-8", errors="replace").read()
            text = strip.sub(lambda m: m.group(1) or " ", text)
            h.update(name.encode())
            h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max) if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = max(1, min(n, quota // period))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(rt3, cr, mats, cam, budget_s):
    """Times the CPU oracle (kind 'port': the reference has no Mode-X renderer and cannot be built here) on a
    bounded sample of the same workload: same scene / camera / depth / spp law, reduced frame and spp."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = host_cores()
    ocam = O.Camera()
    for f in ("origin", "horizontal", "vertical", "lower_left_corner"):
        setattr(ocam, f, getattr(cam.c, f))
    smats = mats.view(O.MATERIAL)

    def run(w, h, spp, threads):
        p = O.make_params(w, h, spp=spp, max_depth=DEPTH, seed=RENDER_SEED, flags=O.FLAG_GAMMA2, lens_radius=0.05)
        t0 = time.perf_counter()
        O.render_path(ocam, p, spheres=cr, smats=smats, threads=threads)
        return w * h * spp / (time.perf_counter() - t0) / 1e6

    # a GPU box hands one job a share of a big host (16 cores per GPU): use the thread count that is actually faster
    if cores > 16 and run(240, 135, 2, 16) > run(240, 135, 2, cores):
        cores = 16
    # calibrate on a small frame, then size the timed sample to ~budget_s
    probe = run(240, 135, 2, cores)
    total = max(240 * 135 * 2, int(probe * 1e6 * budget_s))
    # the workload's own frame with the first `spp` of its 512 samples per pixel (not a perfect square, like 512, so the
    # same un-stratified sampling law applies); a slow host falls back to a smaller 16:9 frame at 2 spp
    w, h = WIDTH, HEIGHT
    spp = int(total // (w * h))
    if spp < 2:
        spp = 2
        h = max(54, int((total / spp / (16.0 / 9.0)) ** 0.5))
        w = h * 16 // 9
    elif int(spp ** 0.5) ** 2 == spp:
        spp += 1
    spp = min(spp, SPP)
    rate_n = run(w, h, spp, cores)
    sample = "%dx%dx%dspp depth %d (same scene/camera/seed), %d threads" % (w, h, spp, DEPTH, cores)
    rate_1 = run(max(32, w // 4), max(18, h // 4), max(2, spp // 4), 1)
    return {"value": round(rate_n, 4), "unit": "Msamples/s", "cores": cores, "kind": "port", "sample": sample,
            "value_1thread": round(rate_1, 4)}


def counters_from_profile(fingerprint):
    """HBM traffic per k_trace launch and vector-ALU issue utilisation from the committed rocprofv3 --pmc passes of THIS command
    (tools/profile_final.sh).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE taken as is; both
    count KiB.  Returns (traffic_bytes, traffic_source, valu_busy dict) — Nones when there is no profile or it is stale."""
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
    except (OSError, ValueError):
        return None, "no committed profile (%s)" % PMC_PROFILE, None
    made_from = pmc.get("_source_fingerprint")
    if made_from != fingerprint:
        return None, "%s was collected on kernel sources %s, this build is %s: stale, not reported" % (PMC_PROFILE, made_from, fingerprint), None
    src = "%s (separate rocprofv3 --pmc passes of `python3 bench.py`, kernel sources %s)" % (PMC_PROFILE, made_from)
    try:
        traffic = int((2.0 * pmc["FETCH_SIZE"]["sum_over_dispatches"] / pmc["FETCH_SIZE"]["dispatches"] +
                       pmc["WRITE_SIZE"]["sum_over_dispatches"] / pmc["WRITE_SIZE"]["dispatches"]) * 1024)
        cycles = pmc["SQ_BUSY_CYCLES"]["sum_over_dispatches"] / 32.0 * 1024.0                     # SIMD-cycles over all dispatches
        n_valu = pmc["SQ_INSTS_VALU"]["sum_over_dispatches"] - pmc["SQ_INSTS_MFMA"]["sum_over_dispatches"]
        busy = pmc["SQ_ACTIVE_INST_VALU"]["sum_over_dispatches"] * 4.0
        disp = pmc["SQ_INSTS_VALU"]["dispatches"]
        valu = {"frac": round(busy / cycles, 3),
                "busy_simd_cycles_per_launch": int(busy / disp), "simd_cycles_per_launch": int(cycles / disp),
                "mfma_pipe_busy": round(pmc["SQ_VALU_MFMA_BUSY_CYCLES"]["sum_over_dispatches"] / cycles, 3),
                "instruction_slot_frac": round(n_valu * 4.0 / cycles, 3),
                "valu_instructions_per_launch": int(pmc["SQ_INSTS_VALU"]["sum_over_dispatches"] / disp),
                "source": src, "note": "frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x busy cycles): the share of all SIMD-cycles in which the vector ALU "
                                       "is occupied by an instruction — one count per issued vector instruction (matrix instructions included: their issue "
                                       "occupies the same port) and one more per further 4-cycle pass of a multi-pass instruction (17.8 per "
                                       "v_cvt_scalef32_2xpk16_fp6_f32: tools/ubench_fp6_decode.hip measures 63-65 cycles for it).  instruction_slot_frac = "
                                       "(SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 / the same cycles is round 2's figure (0.913 when the decode was 32 v_alignbit_b32 "
                                       "per row block; one conversion now does that work, so the instruction count fell and the unit is as busy as before); "
                                       "mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / the same SIMD-cycles (matrix instructions only: the conversion, which also "
                                       "holds the matrix pipe, is not in it)"}
        return traffic, src, valu
    except (KeyError, ZeroDivisionError):
        return None, "%s lacks FETCH_SIZE / WRITE_SIZE" % PMC_PROFILE, None


def tiled_counters(fingerprint):
    """Vector-ALU busy fraction and matrix-pipe busy fraction of the kernels behind extra_workloads from profiles/r03_tiled_pmc.json (tools/profile_tiled.sh),
    under the same rule as the headline's: only for the kernel sources they were collected on.  One definition: SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles."""
    try:
        pmc = json.load(open(os.path.join(ROOT, TILED_PMC_PROFILE)))
    except (OSError, ValueError):
        return {}, "no committed profile (%s)" % TILED_PMC_PROFILE
    made_from = pmc.get("_source_fingerprint")
    if made_from != fingerprint:
        return {}, "%s was collected on kernel sources %s, this build is %s: stale, not reported" % (TILED_PMC_PROFILE, made_from, fingerprint)
    out = {}
    for key, c in pmc.items():
        if not isinstance(c, dict) or "SQ_BUSY_CYCLES" not in c:
            continue
        cycles = c["SQ_BUSY_CYCLES"]["sum_over_dispatches"] / 32.0 * 1024.0
        out[key] = {"valu_busy": round(c["SQ_ACTIVE_INST_VALU"]["sum_over_dispatches"] * 4.0 / cycles, 3),
                    "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum_over_dispatches"] / cycles, 3)}
    return out, "%s (rocprofv3 --pmc passes of tools/run_config.py, kernel sources %s)" % (TILED_PMC_PROFILE, made_from)


def executed(st, k_slots=64):
    """Executed-work figures of the last render on a context, from the kernel's own counters and its HIP-event time.
    k_slots: K of the filter contraction the kernel ran (64, or 32 for the sphere pass of the tiled kernels): 2 K FLOP per test."""
    k_ms = st.trace_ms
    tf = st.mfma_instructions * float(st.mfma_flop_per_instruction) / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    d = {"kernel_ms": round(k_ms, 3), "launches": st.launches, "ray_casts": int(st.ray_casts), "prim_tests": int(st.prim_tests),
         "tests_per_s": round(st.prim_tests / (k_ms * 1e-3), 1) if k_ms > 0 else 0.0,
         "mfma_tflops": round(tf, 1), "mfma_frac_of_bf16_peak": round(tf / PEAK_BF16_MFMA_TFLOPS, 4)}
    if st.mfma_instructions:
        # the matrix cores work per wave whatever the number of live lanes: tests needed / tests the issued MFMAs evaluated
        d["lane_efficiency"] = round(st.prim_tests * 2.0 * k_slots / (st.mfma_instructions * float(st.mfma_flop_per_instruction)), 4)
        d["filter_k"] = k_slots
    if k_slots != 64 and k_ms > 0:
        # the same test rate priced in the K = 64 form of round 1 / the other kernels: an equivalence figure for comparisons across rounds,
        # not work this kernel executes (it runs half the matrix work per test, which is why it is faster)
        d["k64_equivalent_frac_of_bf16_peak"] = round(st.prim_tests * 128.0 / (k_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
    if st.exact_tests:
        d["exact_tests_per_cast"] = round(st.exact_tests / max(1, st.ray_casts), 2)
    if st.filter_tests and st.filter_tests != st.prim_tests:
        # multi-level filter: the matrix cores evaluated one row per 8 leaf groups of 8 primitives, the leaves' bounds were tested in f32; prim_tests / tests_per_s above are the
        # brute-force-equivalent count (ray casts x primitives: what the reference's loop would execute), these are what was executed
        d["filter"] = {"levels": 3, "filter_rows_per_cast": int(st.filter_tests // max(1, st.ray_casts)),
                       "filter_tests_executed": int(st.filter_tests), "filter_tests_per_s": round(st.filter_tests / (k_ms * 1e-3), 1) if k_ms > 0 else 0.0,
                       "bound_tests_per_cast": round(st.bound_tests / max(1, st.ray_casts), 2),
                       "note": "tests_per_s is brute-force EQUIVALENT (ray casts x primitives / kernel time); the matrix filter executed "
                               "filter_tests_executed (ray, row) pairs — a row bounds 64 primitives — then f32 bound tests of the candidate rows' leaf groups "
                               "(and, for faces, of their members) and exact tests on the survivors"}
        d["lane_efficiency"] = round(st.filter_tests * 2.0 * k_slots / (st.mfma_instructions * float(st.mfma_flop_per_instruction)), 4) if st.mfma_instructions else None
    return d


def extra_workloads(rt3, r, np, fingerprint):
    """Short, driver-visible runs of the other workloads (kernel time = HIP events inside the C ABI; scene upload excluded)."""
    out = []
    counters, counters_source = tiled_counters(fingerprint)
    empty_f, empty_v = np.zeros(0, rt3.GFACE), np.zeros((0, 4), np.float32)
    no_sph = (np.zeros((0, 4), np.float32), np.zeros(0, rt3.MATERIAL))

    def path(name, cam, params, kernel, k_slots=64, counter_key=None):
        r.render_path(cam.c, params)                                     # warm-up: allocations, occupancy query
        r.render_path(cam.c, params)
        st = r.stats()
        d = {"workload": name, "kernel": kernel, "samples": int(st.samples), "ms": round(st.total_ms, 3),
             "msamples_per_s": round(st.samples / st.total_ms / 1e3, 2)}
        d.update(executed(st, k_slots))
        d.update(counters.get(counter_key, {"valu_busy": None, "mfma_busy": None}))
        d["counters_source"] = counters_source
        out.append(d)

    # Mode R: the reference's own render (SequentialRenderer::render) of its built-in scene, the one workload with a reference CPU time
    fixture = os.path.join(ROOT, "tests", "golden", "builtin_scene.npz")
    if os.path.exists(fixture):
        z = np.load(fixture)
        faces = z["faces"].view(rt3.GFACE).reshape(-1)
        r.set_mesh(faces, z["verts"])
        r.set_spheres(*no_sph)
        cam = rt3.main_camera(1920, 1080)
        times = []
        for _ in range(6):
            r.render(cam)
            times.append(r.stats().trace_ms)
        ms = sorted(times[1:])[len(times[1:]) // 2]
        npix, nf = 1920 * 1080, len(faces)
        waves, blocks = -(-npix // 1024) * 16, -(-nf // 32)
        tf = waves * blocks * 8 * 16384.0 / (ms * 1e-3) / 1e12                 # K = 32 filter: 8 x v_mfma_f32_16x16x32_bf16 (16384 FLOP) per row block and wave
        out.append({"workload": "Mode R: built-in scene of src/Main.cpp:280-283 (teddy.obj + 8x8 sphere, %d faces), 1920x1080, 1 ray per pixel" % nf,
                    "kernel": "k_mode_r_mfma (K = 32 filter: one v_mfma_f32_16x16x32_bf16 per 16 x 16 tests)", "filter_k": 32, "samples": npix, "ms": round(ms, 4), "msamples_per_s": round(npix / ms / 1e3, 1),
                    "prim_tests": npix * nf, "tests_per_s": round(npix * nf / (ms * 1e-3), 1),
                    "mfma_tflops": round(tf, 1), "mfma_frac_of_bf16_peak": round(tf / PEAK_BF16_MFMA_TFLOPS, 4),
                    "k64_equivalent_frac_of_bf16_peak": round(2.0 * tf / PEAK_BF16_MFMA_TFLOPS, 4),
                    "reference_cpu_s": MODE_R_REFERENCE_CPU_S, "speedup_vs_reference_cpu": round(MODE_R_REFERENCE_CPU_S / (ms * 1e-3), 0),
                    "note": "reference_cpu_s: the reference's SequentialRenderer on this frame, 1 thread, measured by the survey (SURVEY.md §6); "
                            "pixels equal the reference's PPM SHA-256 (tests/test_gpu_mode_r.py)",
                    "counters_source": counters_source, **counters.get("config_r", {"valu_busy": None, "mfma_busy": None})})
        r.set_mesh(empty_f, empty_v)
    # (config 3 runs the headline kernel on the headline scene: leaving it out keeps k_trace_mfma32's rocprofv3 average = the headline launch)
    # config 4: 100 000 spheres
    cr, mats = rt3.scene_stress(100000, 43)
    r.set_mesh(empty_f, empty_v)
    r.set_spheres(cr, mats)
    cam = rt3.Camera().look_at(1920, 1080, (0.0, 8.0, 12.0), (0.0, 6.0, -50.0), (0.0, 1.0, 0.0), 45.0, 1.0)
    path("config 4: 100 000 spheres, 1920x1080, 16 of 256 spp, depth 50", cam,
         rt3.make_params(1920, 1080, spp=16, max_depth=50, flags=rt3.FLAG_GAMMA2),
         "k_trace_mfma_tiled<spheres> (three-level filter: K = 32 matrix filter over rows of 64 spheres, f32 bounds of their 8 leaf groups, exact tests of the members)", k_slots=32,
         counter_key="config_4")
    # config 5: Cornell-style box, 47 106 triangles, emissive quad
    faces, verts, fm = rt3.scene_cornell(64)
    r.set_spheres(*no_sph)
    r.set_mesh(faces, verts, fm)
    cam = rt3.Camera().update(1024, 1024, 2.0, 2.0, 2.0)
    path("config 5: Cornell-style box, %d triangles, emissive quad, 1024x1024, 32 of 2048 spp, depth 50" % len(faces), cam,
         rt3.make_params(1024, 1024, spp=32, max_depth=50, flags=rt3.FLAG_GAMMA2 | rt3.FLAG_BLACK_BACKGROUND),
         "k_trace_mfma_tiled<faces> (three-level filter: K = 32 matrix filter over rows of 64 faces, f32 bounds of leaf groups and of faces, the reference's test on the survivors)",
         k_slots=32, counter_key="config_5")
    r.set_mesh(empty_f, empty_v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_workloads runs")
    ap.add_argument("--save-ppm", default="", help="rank 0 writes the last frame here")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that ALL render on cuda:0 and gather over gloo through host memory: exercises this file's multi-rank control flow "
                         "(sharding, gather, stat reduction, the JSON line) where only one GPU is present; its numbers are not a measurement")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    rehearsal = args.rehearse_on_one_gpu and world > 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("RT3_BENCH_FORCE_DIST") == "1"     # the env knob rehearses the RCCL path on one GPU
    if use_dist:
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    red_dev = torch.device("cpu") if rehearsal else dev                        # where the reduced scalars live (gloo reduces host tensors)

    rt3 = importlib.import_module("raytracer-3_amd")
    W, H = args.width, args.height
    cr, mats = rt3.scene_weekend(SCENE_SEED)
    cam = rt3.weekend_camera(W, H)
    r = rt3.initialize_renderer(local_rank)
    r.prerender([])
    r.set_spheres(cr, mats)

    params = [rt3.make_params(W, H, spp=args.spp, max_depth=args.depth, seed=RENDER_SEED, flags=rt3.FLAG_GAMMA2,
                              lens_radius=0.05, tile_rows=TILE_ROWS, tile_index=i, tile_count=world) for i in range(world)]
    my = params[rank]
    shard = importlib.import_module("raytracer-3_amd.shard")
    g = shard.FrameGatherer(rt3, params, rank, dev, force_collective=use_dist and world == 1, renderer=r, stage_host=rehearsal)
    tile = g.tile
    # One explicit stream for the render, the gather and everything torch queues around them: the C ABI is handed its handle (never NULL,
    # which include/rt3.h defines as "the context's own stream"), RCCL orders its kernels behind torch's CURRENT stream, which is this one
    stream = torch.cuda.Stream(device=dev)

    def step():
        with torch.cuda.stream(stream):
            r.render_path_device(cam.c, my, tile.data_ptr(), stream.cuda_stream)
            g.gather(stream.cuda_stream)       # N>1: ONE RCCL gather over xGMI (8.3 MB / N per peer); N=1: rt3_gather_rows on the same stream

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    st = r.stats()                                                       # HIP events of the last step, on `stream`
    trace_ms, tests, casts, launches = st.trace_ms, st.prim_tests, st.ray_casts, st.launches

    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    agg = torch.tensor([float(tests), float(casts), float(trace_ms)], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg[:2], op=dist.ReduceOp.SUM)
    elapsed = float(t.item())

    if rank == 0:
        samples = W * H * args.spp
        ms_per_step = elapsed / args.steps * 1e3
        value = samples / (elapsed / args.steps) / 1e6
        n_launch = max(1, launches)
        k_ms = trace_ms / n_launch                                       # average duration of one k_trace launch on THIS rank (HIP events)
        mfma_tf = st.mfma_instructions * float(st.mfma_flop_per_instruction) / n_launch / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        alg_tf = st.prim_tests * FLOP_PER_SPHERE_TEST / n_launch / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        hbm_bytes = st.samples * 12.0 / n_launch                         # algorithmic HBM bytes per launch: one 12-B radiance record per sample
        fingerprint = source_fingerprint()
        full_workload = world == 1 and (W, H, args.spp, args.depth) == (WIDTH, HEIGHT, SPP, DEPTH)
        traffic, traffic_source, valu = counters_from_profile(fingerprint) if full_workload else (None, "not the profiled workload", None)
        if st.mfma_instructions:
            k32 = st.mfma_flop_per_instruction == 16384                      # k_trace_mfma32 (default) | k_trace_mfma (RT3_MFMA_K64=1, round 1's form)
            mfma = {"achieved": round(mfma_tf, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(mfma_tf / PEAK_BF16_MFMA_TFLOPS, 4),
                    "mfma_instructions_per_launch": int(st.mfma_instructions / n_launch),
                    "mfma_instruction": "v_mfma_f32_16x16x32_bf16" if k32 else "v_mfma_f32_32x32x16_bf16",
                    "flop_per_mfma_instruction": int(st.mfma_flop_per_instruction), "filter_k": 32 if k32 else 64,
                    # the same test rate priced in round 1's K = 64 form (128 FLOP per test): for comparisons across rounds, not executed work
                    "k64_equivalent_frac": round(st.prim_tests * 128.0 / n_launch / (k_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if k_ms > 0 else 0.0,
                    "live": True,
                    "note": "EXECUTED matrix work, measured in this run: wave-instructions counted by the kernel x FLOP per instruction / kernel time (HIP "
                            "events on the launch stream) over the dense bf16 peak.  The K = 32 filter (DESIGN.md 5.2b) needs half the matrix work per "
                            "test of round 1's K = 64 form; the matrix pipe is about a third busy with them, and one block conversion per 8 of them (which holds the vector ALU and the matrix pipe for 64 cycles) reads the 32 signs"}
            common = {"kernel": "k_trace_mfma32" if k32 else "k_trace_mfma", "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": round(k_ms, 3),
                      "launches_per_step": launches, "kernel_sources": fingerprint, "mfma": mfma}
            if valu is not None:
                roofline = dict(common, bound="valu_busy", achieved=valu["busy_simd_cycles_per_launch"], peak=valu["simd_cycles_per_launch"],
                                unit="SIMD-cycles per launch in which the vector ALU is occupied (peak: all SIMD-cycles of the launch)", frac=valu["frac"],
                                valu_busy=valu, live=["kernel_ms", "mfma (all of it)"],
                                note="bound = the unit this kernel saturates: the vector ALU is occupied in `frac` of all SIMD-cycles of the launch "
                                     "(hardware counters of the committed rocprofv3 passes of this command, valid for these kernel sources only); the "
                                     "executed matrix fraction, measured live, is under `mfma`")
            else:
                roofline = dict(common, bound="mfma", achieved=mfma["achieved"], peak=mfma["peak"], unit=mfma["unit"], frac=mfma["frac"], valu_busy=None,
                                live=["achieved", "frac", "kernel_ms", "mfma"],
                                note="no valid counter profile for this run (%s): the live executed-matrix fraction stands in; the kernel's binding "
                                     "unit is the vector ALU (see profiles/README.md)" % traffic_source)
        else:
            roofline = {"bound": "valu_busy", "kernel": "k_trace (vector-ALU scan, RT3_NO_MFMA / RT3_BRUTE)", "achieved": None, "peak": None,
                        "unit": "SIMD-cycles per launch in which the vector ALU is occupied", "frac": None, "traffic": None, "traffic_source": "A/B build, not profiled",
                        "kernel_ms": round(k_ms, 3), "launches_per_step": launches, "live": ["kernel_ms"], "mfma": None,
                        "note": "no matrix instructions issued by this kernel selection (A/B reference)"}
        out = {
            "metric": "Msamples/sec (pixels x spp) at %dx%dx%dspp" % (W, H, args.spp),
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "In-One-Weekend final random-spheres scene, %d spheres (scene seed %d), %dx%d, %d spp, "
                                   "depth %d, thin lens, gamma 2" % (len(cr), SCENE_SEED, W, H, args.spp, args.depth),
                       "sharding": ("REHEARSAL: %d ranks on ONE GPU, interleaved %d-row blocks, gloo gather through host memory — control flow only, not a "
                                    "measurement" % (world, TILE_ROWS)) if rehearsal else
                                   ("interleaved %d-row blocks over %d GPU(s), RCCL gather to rank 0" % (TILE_ROWS, world)) if use_dist else
                                   "one GPU, rows through rt3_gather_rows (device-to-device)"},
            "ray_casts": int(agg[1].item()), "prim_tests": int(agg[0].item()),
            "tests_per_s": round(agg[0].item() / (elapsed / args.steps), 1),
            "roofline": roofline,
            "algorithmic_equiv": {"flop_per_test": FLOP_PER_SPHERE_TEST, "tests_per_launch": int(st.prim_tests / n_launch), "tflops": round(alg_tf, 2),
                                  "f32_peak_tflops": PEAK_FP32_VALU_TFLOPS, "ratio_to_f32_peak": round(alg_tf / PEAK_FP32_VALU_TFLOPS, 4),
                                  "note": "SURVEY.md 8d's ALGORITHMIC work (20 f32 FLOP per ray-sphere test, tests = ray casts x spheres) per kernel "
                                          "second, beside the f32 peak it would be priced against if it ran as scalar f32 code.  It is not executed in "
                                          "that form, so this is an equivalence figure, not a roofline fraction"},
            "hbm": {"algorithmic_bytes_per_launch": int(hbm_bytes), "achieved_gbs": round(hbm_bytes / (k_ms * 1e-3) / 1e9, 2) if k_ms > 0 else 0.0,
                    "peak_gbs": PEAK_HBM_GBS, "frac": round(hbm_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5) if k_ms > 0 else 0.0,
                    "traffic": traffic, "traffic_source": traffic_source,
                    "note": "the figure north_star asks for: 12 B radiance record per sample; the 7.7 KB scene streams through LDS; "
                            "a high fraction here would mean lost reuse, not success"},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(rt3, cr, mats, cam, args.cpu_seconds)
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        if args.save_ppm:
            f = rt3.Frame(W, H)
            f.data[:] = g.frame.cpu().numpy().view(np.uint32)
            f.to_ppm(args.save_ppm)
        if world == 1 and full_workload and not args.no_extra:
            out["extra_workloads"] = extra_workloads(rt3, r, np, fingerprint)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
