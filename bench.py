#!/usr/bin/env python3
"""Headline benchmark: full 6-DoF tracker iterations per second on 640x480 event frames with
2 000 active points (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one pass of the hot path over one batch: every rank solves `--batch` independent
(keyframe, event-frame) alignments — `--iters` damped Gauss-Newton iterations each, every
iteration being a residual+Jacobian pass over all points, the J^T J / J^T r reduction, the 6x6
solve, the SE(3) update and the accept test — with all inputs already resident in HBM, then the
per-alignment results are gathered across ranks (one small RCCL all-gather).  `value` is the
whole-job number of tracker iterations per second (weak scaling: the per-GPU batch is fixed).

Round 5: the timed steps solve every frame THE WAY A FRAME IS SOLVED THE FIRST TIME — the reference's call pattern is one optimize
per event frame (Tracker.cpp:104), and for such a solve the library samples the 4x4 tiles the frame was written in (its rule:
csrc/eds_strips.hip; the strip copies that make RE-solves of a frame faster cost more than one solve gains).  The bench lets the
library pick that kernel by itself on the frames' real first solve (an untimed step), checks that the timed steps launch the same
one, and pins it with the handle's EDS_FUSED_LAYOUT=tiles knob only because the SAME frames are solved again every step.  The rate
on frames whose strip copies are resident (round 4's headline) is reported beside it as `value_resident_frames`.

The same JSON line carries
  roofline      the kernel of the timed region (persistent per-alignment solver).  `achieved` / `frac` are PHYSICAL: bytes through the
                fabric per launch from the committed rocprofv3 PMC passes of this workload (profiles/traffic_r05.json), corrected as
                MI355X_MICROARCH.md prescribes for gfx950 (2 x FETCH_SIZE + WRITE_SIZE: a 128-byte request is tallied at 64), divided by
                the kernel's duration measured live (HIP events on the library's own stream) and by the 8 TB/s peak — never above 1.
                `frac_credit_8d` keeps SURVEY 8d's crediting (140 B per point-evaluation bicubic: 112 B residual/Jacobian + 28 B
                reduction, J bytes included although a fused kernel never moves them: not a bandwidth fraction); `frac_must_move`
                counts what a fused kernel has to move (20 B of point constants + 64 B of taps = 84 B per point-evaluation)
  roofline_resjac  the stand-alone residual/Jacobian kernel (the streaming two-kernel path), 112 B per point-evaluation
  reference_problem  the same batch solved as the reference's own 12-parameter Ceres-LM problem (eds_fused12_kernel), with
                its own roofline block (196 B credited / 92 B must-move per point-evaluation)
  latency       the regime the reference really runs in (one optimize per event slice, Tracker.cpp:104): one alignment at a
                time (LM6, REF12), one launch of 64 (configs[4] on one GPU), one full live slice
  parity        >= 32 result rows of the timed batch against the CPU oracle; the run FAILS above 1e-4
  cpu_baseline  the CPU oracle (a port: the reference needs Ceres and cannot be built here) running the SAME damped 6-DoF
                iterations on this host's cores, on a bounded sample; cpu_baseline_fast = the optimised fp32 analytic-row CPU
                variant (oracle/eds_cpu_fast.hpp); cpu_baseline_ref12 = the reference-faithful leg: Jet<13> autodiff, Ceres-LM,
                T residual blocks evaluated on T threads (Tracker.cpp:178-195) for T in {1, 8, all}
"""
import argparse
import gc
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E nominal (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 achievable)
BYTES_RESJAC = {"bicubic": 112, "bilinear": 64}      # SURVEY §8d, per point-evaluation
BYTES_REDUCE = 28
BYTES_MUST_MOVE = {"bicubic": 84, "bilinear": 36}    # a fused kernel: 20 B of point constants + the taps; no J planes written or re-read
BYTES_REF12 = {"credited": {"bicubic": 196, "bilinear": 148},      # §8d 12-DoF: reads 28 B + taps, writes r + J[12] = 52 B, reduction reads 52 B
               "must_move": {"bicubic": 92, "bilinear": 44}}       # 28 B of point constants + the taps
PARITY_TOL = 1e-4                                    # SE(3) distance to the oracle's solved pose


def _gen_alignment(args):
    """Pool worker (spawned interpreter, never touches the GPU): one synthetic alignment reduced to what the bench keeps — the keyframe,
    the start, the truth, and the frame as fp32 (the first KEEP_WHOLE also keep the fp64 frame: parity rows and CPU baselines)."""
    seed, H, W, N, keep_frame = args
    synth = importlib.import_module("slam-eds_amd.synth")
    x = synth.make_alignment(seed, H=H, W=W, N=N)
    f32 = np.ascontiguousarray(x.frame, dtype=np.float32)
    if not keep_frame:
        x.frame = None
    return x, f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="alignments per GPU (weak scaling)")
    ap.add_argument("--iters", type=int, default=10, help="tracker iterations per alignment")
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--sampling", choices=["bicubic", "bilinear"], default="bicubic")
    ap.add_argument("--solver", choices=["lm6", "gn6"], default="lm6")
    ap.add_argument("--no-ref12", dest="no_ref12", action="store_true", help="skip the informational REF12 measurement")
    ap.add_argument("--lambda0", type=float, default=0.01, help="initial LM6 damping (DSO template: 0.01)")
    ap.add_argument("--exec", dest="exec_", choices=["device", "host"], default="device")
    ap.add_argument("--distinct", type=int, default=4096, help="distinct synthetic alignments (replicated to fill the batch); the default makes every "
                    "alignment of the 4 096-slot batch its own (seeds 5000 + b)")
    ap.add_argument("--gen-workers", dest="gen_workers", type=int, default=-1, help="worker PROCESSES that generate the synthetic inputs (default: one per core, "
                    "up to 96); 0 = threads inside this process — required under rocprofv3, whose preloaded library initialises the GPU before Python starts: "
                    "such a process must not start children")
    ap.add_argument("--no-configs", dest="no_configs", action="store_true", help="skip the block of the other BASELINE.json configs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-shared", dest="no_shared", action="store_true", help="skip the informational two-batches-in-flight and shared-frame measurements (the latter launches the "
                    "headline kernel on another workload: keep it out of profiler runs)")
    return ap.parse_args()


def pmc_traffic(kernel_prefix, a):
    """Bytes through the fabric per launch from the committed rocprofv3 PMC passes (profiles/traffic_*.json, produced by
    tools/profile.sh + tools/summarise_profile.py in separate --pmc runs) — only when they were taken on exactly this workload;
    otherwise None.  Corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies a 128-byte request at 64 bytes, so
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        c = t.get("bench_config", {})
        if (c.get("alignments_per_gpu"), c.get("points"), c.get("iterations"), c.get("solver"), c.get("sampling"), c.get("exec")) != \
                (a.batch, a.points, a.iters, a.solver, a.sampling, a.exec_) or c.get("frame") != [a.height, a.width]:
            continue
        # (the profiler prints every template argument — "eds_fused6_kernel<0, 4, 512, 1, 1, 1>" with the defaulted GROUPS —, the library's
        # eds_trk_last_launch only those that differ from the default: match the name, or the name continued by further arguments)
        stem = kernel_prefix[:-1] if kernel_prefix.endswith(">") else kernel_prefix
        for k, v in t.get("kernels", {}).items():
            if (k == kernel_prefix or k.startswith(stem + ",") or (not kernel_prefix.endswith(">") and k.startswith(stem))) \
                    and v.get("fetch_kb") is not None and v.get("write_kb") is not None:
                best = {"bytes": (2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0, "raw_bytes": (v["fetch_kb"] + v["write_kb"]) * 1024.0,
                        "read_requests": (v.get("l2") or {}).get("TCC_EA0_RDREQ_sum"), "profiled_avg_us": v.get("avg_us"),
                        "source": os.path.relpath(f, ROOT)}
    return best


def physical_roofline(kernel, k_ms, units, per_unit_credit, per_unit_must_move, a, extra=None):
    """The roofline block of one kernel: PHYSICAL fraction from the committed counters (never above 1 by construction of what it
    divides), SURVEY 8d's credit and the must-move figure beside it.  units = point-evaluations per launch."""
    t = pmc_traffic(kernel, a)
    cred = units * per_unit_credit / (k_ms * 1e-3) / 1e9
    mm = units * per_unit_must_move / (k_ms * 1e-3) / 1e9
    r = {"kernel": kernel, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": k_ms,
         "achieved_credit_8d": cred, "frac_credit_8d": cred / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": units * per_unit_credit,
         "achieved_must_move": mm, "frac_must_move": mm / HBM_PEAK_GBS, "must_move_bytes_per_launch": units * per_unit_must_move}
    if t:
        ach = t["bytes"] / (k_ms * 1e-3) / 1e9
        r.update({"achieved": ach, "frac": ach / HBM_PEAK_GBS, "traffic": t["bytes"], "traffic_source": t["source"],
                  "traffic_read_requests": t["read_requests"], "traffic_profiled_kernel_us": t["profiled_avg_us"],
                  "basis": "physical: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch from the committed PMC passes / the kernel time measured in this run"})
    else:                       # no counters of this workload committed: the must-move bytes are the only honest numerator
        r.update({"achieved": mm, "frac": mm / HBM_PEAK_GBS, "traffic": None,
                  "basis": "must-move bytes (no rocprofv3 PMC passes of this exact workload under profiles/): a LOWER bound of the physical rate"})
    if extra:
        r.update(extra)
    return r


def _under_profiler():
    """rocprofv3 preloads its tool library, which initialises the GPU before Python starts: such a process must not start children."""
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ)


def _usable_cpus():
    """CPUs this process may really use: the scheduler affinity, capped by the cgroup's CPU quota (cpu.max: the GPU boxes of the pool
    show 256 hardware threads and a quota of 16 CPUs — 256 busy threads there share 16 CPUs' worth of time and are throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def _cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable, quota = _usable_cpus()
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "usable_cpus": usable, "cgroup_cpu_quota": quota}


def cpu_baselines(als, iters, sampling, budget_s):
    """Oracle (test infrastructure) timed as the CPU baseline on this host's cores: (1) the same LM6 iterations as the headline,
    (2) the optimised variant of them (fp32 SoA, analytic rows, AVX2 over points), (3) the reference problem the way the reference
    runs it.  Round 5: the all-core figures are driven from C (oracle/eds_oracle_capi.cpp: eds_oracle_bench_lm6 — persistent
    std::threads taking solves off an atomic counter, no interpreter in the loop), and the block evaluations of the REF12 leg run on
    a persistent pool like Ceres' (oracle/eds_oracle.hpp: EvalPool)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    samp = po.BICUBIC if sampling == "bicubic" else po.BILINEAR
    info = _cpu_info()
    cores = info["usable_cpus"]                   # threads actually used: what the box lets this process have (affinity and cgroup quota), not its thread count
    oracles = [po.Oracle(a, sampling=samp) for a in als]
    starts = [(a.p0, a.q0, a.v0) for a in als]
    out = {}
    one = po.bench_lm6(oracles, starts, iters=iters, lambda0=0.01, threads=1, budget_s=budget_s * 0.1)
    allc = po.bench_lm6(oracles, starts, iters=iters, lambda0=0.01, threads=cores, budget_s=budget_s * 0.25)
    out["cpu_baseline"] = {"value": allc["iterations_per_s"], "unit": "iterations/s", "cores": cores, "kind": "port",
                           "sample": f"{allc['solves']} alignments x {iters} LM6 iterations (640x480-class, same inputs; oracle pose6_lm, Jet autodiff) over "
                                     f"{allc['seconds']:.1f} s on {cores} threads driven from C; one core: {one['iterations_per_s']:.1f} iterations/s",
                           "one_core_value": one["iterations_per_s"], "scaling_vs_one_core": allc["iterations_per_s"] / max(one["iterations_per_s"], 1e-9), **info}
    if sampling == "bicubic":
        fast = [po.FastLM6(o, a.v0) for o, a in zip(oracles, als)]
        one_f = po.bench_lm6(oracles, starts, iters=iters, lambda0=0.01, threads=1, budget_s=budget_s * 0.08, fast=fast)
        all_f = po.bench_lm6(oracles, starts, iters=iters, lambda0=0.01, threads=cores, budget_s=budget_s * 0.2, fast=fast)
        out["cpu_baseline_fast"] = {"value": all_f["iterations_per_s"], "unit": "iterations/s", "cores": cores, "kind": "port",
                                    "sample": f"optimised CPU variant (fp32 sampling, analytic 1x6 rows, SoA, "
                                              f"{'AVX2 over points' if po.fast_is_vectorised() else 'scalar'}, inputs converted once): {all_f['solves']} alignments x {iters} LM6 "
                                              f"iterations over {all_f['seconds']:.1f} s on {cores} threads driven from C; one core: {one_f['iterations_per_s']:.1f} iterations/s",
                                    "one_core_value": one_f["iterations_per_s"], "scaling_vs_one_core": all_f["iterations_per_s"] / max(one_f["iterations_per_s"], 1e-9),
                                    "vectorised": po.fast_is_vectorised()}
    # the reference-faithful leg: 12 parameters, Jet<13> autodiff, Ceres-LM rules, `num_threads` = T residual blocks evaluated by
    # T threads (Tracker.cpp:178-195), ONE alignment at a time like Tracker::optimize
    ref = {}
    a0 = als[0]
    for T in sorted({1, min(8, cores), cores}):
        o12 = po.Oracle(a0, sampling=samp, num_blocks=T, eval_threads=T, max_num_iterations=iters)
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.3:          # warm-up: the pool's workers exist and are spread over the cores
            o12.solve_lm(a0.p0, a0.q0, a0.v0)
        t0 = time.perf_counter(); its = 0; n = 0
        while time.perf_counter() - t0 < budget_s * 0.1 or n < 2:
            its += o12.solve_lm(a0.p0, a0.q0, a0.v0)["num_iterations"]; n += 1
        dt = time.perf_counter() - t0
        ref[f"T{T}"] = {"lm_iterations_per_s": its / dt, "ms_per_alignment": 1e3 * dt / n, "threads": T, "solves": n}
    out["cpu_baseline_ref12"] = {"kind": "port", "unit": "LM iterations/s (one alignment at a time, T blocks on T threads of a persistent pool)", **ref, **info,
                                 "sample": f"oracle solve_lm (Jet<13>, Ceres-LM restatement), 640x480-class / {a0.N} points, {iters} iterations"}
    return out


def latency_block(capi, synth, al, a):
    """The regime the reference runs in — one optimize per event slice (Tracker.cpp:104-241): wall time of one alignment at a time
    (LM6, REF12 with 4 blocks + Huber), of one launch of 64 alignments (configs[4] on a single GPU), and of one live slice
    (100 k events -> event frame on the device -> REF12 solve warm-started -> MAD loss scale -> getCoord)."""
    def med(f, reps=20, warm=3):
        for _ in range(warm):
            f()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(t))

    samp = capi.SAMPLE_BICUBIC if a.sampling == "bicubic" else capi.SAMPLE_BILINEAR
    out = {}
    h = capi.Handle(capi.default_config(sampling=samp, solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0),
                    1, al.N, al.H, al.W)
    h.set_alignment(0, al)
    out["B1_lm6_ms"] = med(lambda: h.optimize(0, p=al.p0, q=al.q0, v=al.v0))
    out["B1_lm6_kernel_ms"] = h.info(0)["device_time_us"] * 1e-3
    # the same calls looped INSIDE the library (eds_trk_bench_live): what a C++ caller pays — the figures above include the Python
    # binding's own work around every call (argument conversion, ~10 us)
    out["B1_lm6_c_ms"] = h.bench_live(0, al.p0, al.q0, al.v0, reps=100)["optimize_us"] * 1e-3
    h.set_config(capi.default_config(sampling=samp, solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, num_blocks=1))
    out["B1_ref12_ms"] = med(lambda: h.optimize(0, p=al.p0, q=al.q0, v=al.v0))
    out["B1_ref12_kernel_ms"] = h.info(0)["device_time_us"] * 1e-3
    out["B1_ref12_c_ms"] = h.bench_live(0, al.p0, al.q0, al.v0, reps=100)["optimize_us"] * 1e-3
    # one live slice
    h.set_config(capi.default_config(sampling=samp, solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, num_blocks=4,
                                     loss_type=capi.LOSS_HUBER, loss_param=0.3))
    rng = np.random.default_rng(0)
    strong = np.argwhere(np.abs(al.frame) > 0.25 * np.abs(al.frame).max())
    pick = strong[rng.integers(0, len(strong), 100_000)]
    ex, ey = pick[:, 1].astype(np.uint16), pick[:, 0].astype(np.uint16)
    pol = (al.frame[pick[:, 0], pick[:, 1]] > 0).astype(np.uint8)

    def one_slice():
        h.build_event_frame(0, ex, ey, pol)
        h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
        h.loss_param(0, capi.LP_MAD)
        h.update_points(0, False)
    out["slice_ms"] = med(one_slice)

    # the same call as the drop-in shim makes it (Tracker::optimize with a HOST fp64 frame): depths, frame upload, solve,
    # residuals + MAD loss scale in one call (kf->residuals as the MAD's reorder leaves them)
    frame64 = np.ascontiguousarray(al.frame, dtype=np.float64)
    idp64 = np.ascontiguousarray(al.idp, dtype=np.float64)

    def live_call():
        h.set_idepth(0, idp64)
        h.set_event_frame(0, frame64)
        h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
        h.residuals_and_loss(0, capi.LP_MAD)
    out["live_call_ref12_ms"] = med(live_call)
    out["live_call_ref12_kernel_ms"] = h.info(0)["device_time_us"] * 1e-3
    lc = h.bench_live(0, al.p0, al.q0, al.v0, idp=idp64, frame=frame64, method=capi.LP_MAD, reps=200)
    out["live_call_ref12_c_ms"] = lc["total_us"] * 1e-3
    out["live_call_ref12_c_kernel_ms"] = lc["kernel_us"] * 1e-3
    out["live_call_ref12_c_steps_us"] = {k[:-3]: round(v, 1) for k, v in lc.items() if k not in ("total_us", "kernel_us")}
    h.close()

    # configs[3]: one coarse-to-fine call, 4 levels of one scene, 2 000 -> 16 000 points, the pose carried on
    counts = [16000, 8000, 4000, 2000]
    alp = synth.make_alignment(3234, H=480, W=640, N=16000, rot_deg=0.6, trans_norm=0.012, blur_ksize=15, blur_sigma=4.0)
    for solver, key in ((capi.SOLVER_LM6, "config3_lm6_ms"), (capi.SOLVER_REF12, "config3_ref12_ms")):
        pyr = capi.Pyramid(capi.default_config(sampling=samp, solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters), counts, 480, 640)
        for l, n in enumerate(counts):
            pyr.set_keyframe(l, alp.norm_coord[:n], alp.grad[:n], alp.idp[:n], alp.weights[:n], alp.fx, alp.fy, alp.cx, alp.cy)
        pyr.set_event_frame(alp.frame)
        out[key] = med(lambda: pyr.optimize(alp.p0, alp.q0, alp.v0), reps=10)
        pyr.close()
    B64 = 64
    als64 = [synth.make_alignment(5000 + b, H=al.H, W=al.W, N=al.N) for b in range(8)]
    h = capi.Handle(capi.default_config(sampling=samp, solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0),
                    B64, al.N, al.H, al.W)
    for b in range(B64):
        h.set_alignment(b, als64[b % 8])
    P0 = np.stack([als64[b % 8].p0 for b in range(B64)]); Q0 = np.stack([als64[b % 8].q0 for b in range(B64)]); V0 = np.stack([als64[b % 8].v0 for b in range(B64)])

    def batch64():
        h.set_states(0, P0, Q0, V0)
        h.optimize_batch(0, 0, B64, sync=True)
    out["B64_ms"] = med(batch64)
    out["B64_kernel_ms"] = h.info(0)["device_time_us"] * 1e-3

    # a whole tracking step of 64 trackers at once (configs[4] end to end on one GPU): 64 event slices of 20 k events -> frames (one
    # batched call), the 64 solves, the 64 MAD scales, getCoord / culling / keyframe criterion of all 64
    slices = []
    for b in range(B64):
        fr = als64[b % 8].frame
        strong64 = np.argwhere(np.abs(fr) > 0.25 * np.abs(fr).max())
        pk = strong64[rng.integers(0, len(strong64), 20_000)]
        slices.append((pk[:, 1].astype(np.uint16), pk[:, 0].astype(np.uint16), (fr[pk[:, 0], pk[:, 1]] > 0).astype(np.uint8)))
    import ctypes as C
    offs = (np.arange(B64 + 1) * 20_000).astype(np.int32)
    cx = np.concatenate([s_[0] for s_ in slices]); cy = np.concatenate([s_[1] for s_ in slices]); cp = np.concatenate([s_[2] for s_ in slices])

    def step64():
        rc = capi.lib().eds_trk_build_event_frame_batch(h._h, 0, B64, offs.ctypes.data_as(C.POINTER(C.c_int32)), cx.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                        cy.ctypes.data_as(C.POINTER(C.c_uint16)), cp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 0.5, 1, None)
        assert rc == 0
        h.set_states(0, P0, Q0, V0)
        h.optimize_batch(0, 0, B64, sync=True)
        h.loss_param_batch(capi.LP_MAD, 0, B64)
        h.update_points_batch(0, B64, False, want_points=False)
    out["B64_step_ms"] = med(step64, reps=10)
    h.close()
    out["note"] = ("wall time per call through the C ABI, inputs resident; B1: one 640x480-class alignment, B64: one launch of 64 (configs[4] on "
                   "one GPU); slice: 100 k events -> frame -> REF12 (4 blocks, Huber) -> MAD -> getCoord; live_call: the shim's sequence with a host fp64 "
                   "frame; config3: one 4-level coarse-to-fine call (2 000 .. 16 000 points); B64_step: events -> 64 frames -> 64 solves -> MAD -> getCoord criterion, batched calls; "
                   "*_c_ms: the same calls looped inside ONE C call (eds_trk_bench_live, std::chrono around each): the C ABI's own cost, without the Python binding's ~10 us per call")
    return out


def configs_block(capi, synth, a):
    """The other BASELINE.json configs as throughput + roofline + parity in the same record (each outside the headline's timed region):
    configs[2] batched (256 x 1280x720 / 8 000 points, per-point Huber at 1.345 MAD), configs[3] batched (64 four-level pyramids,
    2 000 .. 16 000 points, one launch per level for all of them), configs[4] (64 alignments of configs[1], one launch).  LM6,
    `--iters` iterations (per level), bicubic; >= 8 result rows of each against the CPU oracle (the checker: never inside a timing)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    import np_pyramid_oracle as pyo
    per_pt, mm = BYTES_RESJAC["bicubic"] + BYTES_REDUCE, BYTES_MUST_MOVE["bicubic"]
    pool = ThreadPoolExecutor(min(16, os.cpu_count() or 1))
    out = {}

    def rounded(x):            # the alignment with its frame as the library holds it (fp32)
        return synth.Alignment(**{**x.__dict__, "frame": np.ascontiguousarray(x.frame, dtype=np.float32).astype(np.float64)})

    def timed(f, reps=5):
        f(); f()               # (the second solve on the same frames is the one that makes their strip copies)
        t = []
        for _ in range(reps):
            t0 = time.perf_counter(); r = f(); t.append(time.perf_counter() - t0)
        return float(np.median(t)), r

    def roof(points_passes, k_ms):          # points_passes = sum over launches of alignments x points x passes
        # (no PMC passes are committed for these workloads: `frac` is the must-move figure — a lower bound of the physical rate, never above 1;
        # SURVEY 8d's credit beside it)
        return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": k_ms, "basis": "must-move bytes (no PMC passes of this workload)",
                "achieved": points_passes * mm / (k_ms * 1e-3) / 1e9, "frac": points_passes * mm / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "achieved_credit_8d": points_passes * per_pt / (k_ms * 1e-3) / 1e9, "frac_credit_8d": points_passes * per_pt / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "achieved_must_move": points_passes * mm / (k_ms * 1e-3) / 1e9, "frac_must_move": points_passes * mm / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # ---- configs[2] ------------------------------------------------------------------------------------------------------------
    B2, N2, H2, W2, D2 = 256, 8000, 720, 1280, 8
    als = list(pool.map(lambda i: synth.make_alignment(2234 + i, H=H2, W=W2, N=N2), range(D2)))
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0), B2, N2, H2, W2)
    fr = [np.ascontiguousarray(x.frame, dtype=np.float32) for x in als]
    for b in range(B2):
        x = als[b % D2]
        h.set_keyframe(b, x.norm_coord, x.grad, x.idp, x.weights, x.fx, x.fy, x.cx, x.cy); h.set_event_frame(b, fr[b % D2])
    r0 = h.eval(0, als[0].p0, als[0].q0, als[0].v0, ncols=6, want_jacobian=False)["r"]
    tau = float(1.345 * 1.4826 * np.median(np.abs(r0 - np.median(r0))))            # 1.345 MAD of the start residuals of alignment 0
    h.set_config(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0, huber_tau=tau))
    P0 = np.stack([als[b % D2].p0 for b in range(B2)]); Q0 = np.stack([als[b % D2].q0 for b in range(B2)]); V0 = np.stack([als[b % D2].v0 for b in range(B2)])

    def step2():
        h.set_states(0, P0, Q0, V0); h.optimize_batch(0, 0, B2, sync=True)
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")         # as for the headline: every solve as a frame's first solve (the library's rule for a new frame)
    wall, _ = timed(step2)
    tab = h.results(0, B2); launch = h.last_launch(); k_ms = h.info(0)["device_time_us"] * 1e-3
    h.set_knob("EDS_FUSED_LAYOUT", None)
    h.prepare_frames(0, B2)                         # ... and on frames whose strip copies exist (re-solved frames)
    wall_r, _ = timed(step2)
    launch_r = h.last_launch(); k_ms_r = h.info(0)["device_time_us"] * 1e-3
    its = float(np.mean(tab[:, 14])); passes = a.iters + 1
    worst, mism = 0.0, 0
    for d in range(D2):
        ref = po.Oracle(rounded(als[d])).pose6_lm(als[d].p0, als[d].q0, als[d].v0, iters=a.iters, lambda0=a.lambda0, huber_tau=tau)
        worst = max(worst, po.se3_distance(tab[d, 0:3], tab[d, 3:7], ref["p"], ref["q"])); mism += int(tab[d, 14] != ref["iterations"])
    out["config2"] = {"workload": f"{B2} alignments x {N2} points on {W2}x{H2}, {a.iters} LM6 iterations, per-point Huber tau = 1.345 MAD = {tau:.4g}; frames new for the solve",
                      "iterations_per_s": B2 * its / wall, "ms_per_step": 1e3 * wall, "kernel": launch["kernel"], "cus_per_alignment": launch["cus_per_alignment"],
                      "roofline": roof(B2 * N2 * passes, k_ms), "success_fraction": float(np.mean(tab[:, 15])),
                      "resident_frames": {"iterations_per_s": B2 * its / wall_r, "ms_per_step": 1e3 * wall_r, "kernel": launch_r["kernel"], "roofline": roof(B2 * N2 * passes, k_ms_r)},
                      "parity": {"rows_checked": D2, "parity_max_se3": worst, "iteration_count_mismatches": mism, "tolerance": PARITY_TOL}}
    h.close()

    # ---- configs[3] ------------------------------------------------------------------------------------------------------------
    counts, B3, D3 = [16000, 8000, 4000, 2000], 64, 8
    als = list(pool.map(lambda i: synth.make_alignment(3234 + i, H=480, W=640, N=16000, rot_deg=0.6, trans_norm=0.012, blur_ksize=15, blur_sigma=4.0), range(D3)))
    saved_layout = os.environ.get("EDS_FUSED_LAYOUT")
    os.environ["EDS_FUSED_LAYOUT"] = "tiles"        # the level handles read their knobs at create: every solve as a frame's first solve (as for the headline)
    pyr = capi.Pyramid(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0), counts, 480, 640, batch=B3)
    if saved_layout is None: os.environ.pop("EDS_FUSED_LAYOUT", None)
    else: os.environ["EDS_FUSED_LAYOUT"] = saved_layout
    for b in range(B3):
        x = als[b % D3]
        for l, n in enumerate(counts):
            pyr.set_keyframe_slot(b, l, x.norm_coord[:n], x.grad[:n], x.idp[:n], x.weights[:n], x.fx, x.fy, x.cx, x.cy)
        pyr.set_event_frame_slot(b, x.frame)
    P0 = np.stack([als[b % D3].p0 for b in range(B3)]); Q0 = np.stack([als[b % D3].q0 for b in range(B3)]); V0 = np.stack([als[b % D3].v0 for b in range(B3)])
    wall, (P, Q, V, infos) = timed(lambda: pyr.optimize_batch(P0, Q0, V0))
    its_l = [float(np.mean([infos[l][k]["num_iterations"] for k in range(B3)])) for l in range(len(counts))]
    k_ms = sum(infos[l][0]["device_time_us"] for l in range(len(counts))) * 1e-3
    worst, mism = 0.0, 0
    for d in range(D3):
        rp, rq, rv, per_level = pyo.track(po, synth, als[d], counts, [a.iters] * len(counts), solver="lm6")
        worst = max(worst, po.se3_distance(P[d], Q[d], rp, rq))
        mism += sum(int(infos[l][d]["num_iterations"] != per_level[l]["iterations"]) for l in range(len(counts)))
    out["config3"] = {"workload": f"{B3} coarse-to-fine pyramids, levels 80x60 .. 640x480 with {counts[::-1]} points, {a.iters} LM6 iterations per level, "
                                  f"one launch per level for all pyramids; frames new for the solve",
                      "iterations_per_s": B3 * sum(its_l) / wall, "pyramids_per_s": B3 / wall, "ms_per_step": 1e3 * wall,
                      "iterations_per_level_finest_first": its_l, "kernel": "eds_fused6_kernel, teams of 1 024 or 2 048 points per CU above 2 048 points (one launch per level)",
                      "roofline": roof(sum(B3 * n * (a.iters + 1) for n in counts), k_ms),
                      "parity": {"rows_checked": D3, "parity_max_se3": worst, "iteration_count_mismatches": mism, "tolerance": PARITY_TOL}}
    pyr.close()

    # ---- configs[4] on one GPU -------------------------------------------------------------------------------------------------
    B4 = 64
    als = list(pool.map(lambda b: synth.make_alignment(5000 + b, H=a.height, W=a.width, N=a.points), range(B4)))
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0), B4, a.points, a.height, a.width)
    for b in range(B4):
        h.set_alignment(b, als[b])
    P0 = np.stack([x.p0 for x in als]); Q0 = np.stack([x.q0 for x in als]); V0 = np.stack([x.v0 for x in als])

    def step4():
        h.set_states(0, P0, Q0, V0); h.optimize_batch(0, 0, B4, sync=True)
    wall, _ = timed(step4, reps=20)
    tab = h.results(0, B4); launch = h.last_launch(); k_ms = h.info(0)["device_time_us"] * 1e-3
    worst, mism = 0.0, 0
    for d in range(0, B4, 8):
        ref = po.Oracle(rounded(als[d])).pose6_lm(als[d].p0, als[d].q0, als[d].v0, iters=a.iters, lambda0=a.lambda0)
        worst = max(worst, po.se3_distance(tab[d, 0:3], tab[d, 3:7], ref["p"], ref["q"])); mism += int(tab[d, 14] != ref["iterations"])
    out["config4_one_gpu"] = {"workload": f"{B4} alignments (seeds 5000..5063) x {a.points} points on {a.width}x{a.height}, {a.iters} LM6 iterations, ONE launch "
                                          f"(the 8-GPU config shards them 8 per GPU; this is all 64 on one)",
                              "iterations_per_s": B4 * float(np.mean(tab[:, 14])) / wall, "ms_per_step": 1e3 * wall, "kernel": launch["kernel"],
                              "cus_per_alignment": launch["cus_per_alignment"], "roofline": roof(B4 * a.points * (a.iters + 1), k_ms),
                              "parity": {"rows_checked": len(range(0, B4, 8)), "parity_max_se3": worst, "iteration_count_mismatches": mism, "tolerance": PARITY_TOL}}
    h.close()
    pool.shutdown()
    return out


def strong_scaling_config4(capi, synth, batchmod, a, rank, world, local_rank, dev, forced, dist, torch, steps=200, warmup=20):
    """BASELINE.json configs[4] LITERALLY, as a strong-scaling figure beside the weak-scaling `value` (Next #3b): 64 alignments
    in all (seeds 5000 + b), 64 / N per GPU, one launch per rank and step, then the all-gather of the 64 rows — ms per step including the
    gather, MAX over ranks, bracketed by barrier + synchronize like the headline.  With 8 GPUs every rank holds 8 alignments: the
    latency regime (4 CUs per alignment), so the curve is expected to be nearly flat — the step is one ~60-70 us solve whatever N."""
    TOTAL = 64
    first, count = batchmod.shard_range(TOTAL, world, rank)
    als = [synth.make_alignment(5000 + b, H=a.height, W=a.width, N=a.points) for b in range(first, first + count)]
    cfg = capi.default_config(device=local_rank if world > 1 else 0, solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0,
                              sampling=capi.SAMPLE_BICUBIC if a.sampling == "bicubic" else capi.SAMPLE_BILINEAR)
    h = capi.Handle(cfg, max(1, count), a.points, a.height, a.width)
    for i, x in enumerate(als):
        h.set_alignment(i, x)
    P0 = np.stack([x.p0 for x in als]) if count else np.zeros((0, 3))
    Q0 = np.stack([x.q0 for x in als]) if count else np.zeros((0, 4))
    V0 = np.stack([x.v0 for x in als]) if count else np.zeros((0, 6))
    g = batchmod.ResultGatherer(TOTAL, device=dev, to_host=(rank == 0), force=forced)

    dbg = bool(os.environ.get("EDS_BENCH_DEBUG"))

    def step():
        ta = time.perf_counter()
        if count:
            h.set_states(0, P0, Q0, V0)
            tb = time.perf_counter()
            h.optimize_batch(0, 0, count, sync=True)
        tc = time.perf_counter()
        g.start(h.results(0, count) if count else np.zeros((0, batchmod.RESULT_WIDTH)))
        out_ = g.finish()
        if dbg and count and time.perf_counter() - ta > 1e-3:
            sys.stderr.write(f"[bench] strong-scaling step on rank {rank}: set_states {1e3 * (tb - ta):.3f} ms, optimize {1e3 * (tc - tb):.3f} ms, gather "
                             f"{1e3 * (time.perf_counter() - tc):.3f} ms, kernel {h.info(0)['device_time_us']:.1f} us, flags {h.info(0)['flags']}\n")
        return out_

    tg_ = time.perf_counter(); gc.collect(); gc_ms = 1e3 * (time.perf_counter() - tg_)
    gc.disable()                                     # (as for the headline loop: no interpreter collection inside a timed region, no idle gap in front of it)
    for _ in range(warmup):
        table = step()
    if world > 1 or forced:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms, timeouts, paused = [], 0, 0
    for _ in range(steps):
        ts_ = time.perf_counter()
        table = step()
        step_ms.append(1e3 * (time.perf_counter() - ts_))
        if count:                                    # the library's own diagnostics of the step (include/eds_hip.h: EDS_INFO_*)
            fl = h.info(0)["flags"]
            timeouts += 1 if fl & capi.INFO_TEAM_TIMEOUT else 0
            paused += 1 if fl & capi.INFO_TEAMS_PAUSED else 0
    torch.cuda.synchronize()
    if world > 1 or forced:
        dist.barrier()
    el = time.perf_counter() - t0
    gc.enable()
    if world > 1 or forced:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if dev is not None else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    launch = h.last_launch() if count else None
    k_us = h.info(0)["device_time_us"] if count else 0.0
    h.close()
    if rank != 0:
        return None
    its = float(np.mean(table[:, 14]))
    return {"workload": f"BASELINE.json configs[4]: {TOTAL} alignments in all (seeds 5000..5063) x {a.points} points on {a.width}x{a.height}, {a.iters} LM6 iterations, "
                        f"{-(-TOTAL // world)} per GPU, one all-gather of 16 doubles per alignment per step",
            "scaling": "strong", "n_gpus": world, "alignments_total": TOTAL, "alignments_per_gpu": -(-TOTAL // world), "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * el / steps, "iterations_per_s": TOTAL * its / (el / steps), "alignments_per_s": TOTAL / (el / steps),
            "kernel": launch["kernel"] if launch else None, "cus_per_alignment": launch["cus_per_alignment"] if launch else None, "kernel_ms_rank0": k_us * 1e-3,
            "success_fraction": float(np.mean(table[:, 15])), "rows_gathered": int(table.shape[0]),
            "median_ms_per_step_rank0": float(np.median(step_ms)), "max_ms_per_step_rank0": float(np.max(step_ms)),
            "team_timeouts_rank0": timeouts, "steps_with_teams_paused_rank0": paused, "step_ms_rank0": [round(x, 3) for x in step_ms],
            "interpreter_full_gc_ms": gc_ms,        # what ONE full collection of CPython's collector costs in this process (kept out of the timed regions)
            "note": "strong scaling (total work fixed): informational beside `value`, which is weak scaling at 4 096 alignments per GPU; ms_per_step is the mean "
                    "over the steps (MAX over ranks); a team of CUs that did not assemble within 50 ms is re-run on one CU per alignment and counted here"}


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (torch.distributed.run) before this
    process has made any HIP / torch.cuda call — a process that has initialised the GPU must never exec or be re-used as a
    rank — and exit with the launcher's code.  The children see WORLD_SIZE and take the normal path."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
                         f"number for a configuration that is not the one asked for")
    import torch
    import torch.distributed as dist
    # test hooks (not used by the driver): EDS_BENCH_BACKEND=gloo and EDS_BENCH_DEVICE=<ordinal> let several ranks share
    # one GPU, so that the sharded path can be exercised on a single-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("EDS_BENCH_BACKEND", "nccl")
    if "EDS_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["EDS_BENCH_DEVICE"])
    # EDS_BENCH_FORCE_DIST=1 (single-GPU boxes): a process group of ONE rank over RCCL and the sharded step below — the overlap of the
    # torch stream's all-gather with the library stream's solve and all_gather_into_tensor on a device tensor run on the hardware
    # before the multi-GPU node sees them.  Same workload, same metric; the JSON line says "forced_dist": true.
    forced = world == 1 and os.environ.get("EDS_BENCH_FORCE_DIST") == "1"
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:      # device_count() does not initialise the GPU
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()} "
                         f"(one rank per GPU: RCCL refuses two ranks on one device)")
    # Input generation starts HERE, in spawned worker processes, before this process touches the GPU (a process that has initialised
    # HIP starts no children): 4 096 distinct alignments are ~10 minutes of numpy on one core, seconds on the host's cores.
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    KEEP_WHOLE = 64
    distinct = min(a.distinct, a.batch)
    t_gen = time.perf_counter()
    # (the workers inherit the environment at spawn: ONE BLAS / OpenMP thread each — 64 interpreters with a 256-thread OpenBLAS pool apiece
    # spend their time spinning: 158 s instead of seconds on the 256-thread host)
    saved_env = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update({k: "1" for k in saved_env})
    if a.gen_workers != 0 and _under_profiler():
        sys.stderr.write("[bench] profiler detected (rocprofiler in LD_PRELOAD / ROCP* in the environment): inputs are generated on threads of this process (--gen-workers 0)\n")
        a.gen_workers = 0
    if a.gen_workers == 0:
        from concurrent.futures import ThreadPoolExecutor
        gen_pool = ThreadPoolExecutor(min(32, os.cpu_count() or 1))
    else:
        nw = a.gen_workers if a.gen_workers > 0 else max(4, min(96, _usable_cpus()[0] // max(world, 1)))
        gen_pool = ProcessPoolExecutor(max_workers=nw, mp_context=multiprocessing.get_context("spawn"))
    gen_jobs = [gen_pool.submit(_gen_alignment, (5000 + ((rank * a.batch + i) % max(distinct * world, 1)), a.height, a.width, a.points, i < KEEP_WHOLE))
                for i in range(distinct)]
    for k, v in saved_env.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    if forced:
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port_))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world > 1 or forced:
        world = dist.get_world_size()               # n_gpus is reported from the collective, not from the environment
        assert world == a.gpus, (world, a.gpus)
    capi = importlib.import_module("slam-eds_amd.capi")
    synth = importlib.import_module("slam-eds_amd.synth")
    batchmod = importlib.import_module("slam-eds_amd.batch")
    if capi.device_count() < 1:
        gen_pool.shutdown(wait=False, cancel_futures=True)
        raise SystemExit("bench.py needs a GPU: libeds_hip has no CPU fallback")

    B, N, H, W = a.batch, a.points, a.height, a.width
    total = B * world
    # BASELINE.json configs[4] seeds: 5000 + b for alignment b; `distinct` of them, replicated
    from concurrent.futures import ThreadPoolExecutor
    cfg = capi.default_config(device=local_rank if world > 1 else 0,
                              sampling=capi.SAMPLE_BICUBIC if a.sampling == "bicubic" else capi.SAMPLE_BILINEAR,
                              solver=capi.SOLVER_LM6 if a.solver == "lm6" else capi.SOLVER_GN6,
                              exec=capi.EXEC_DEVICE if a.exec_ == "device" else capi.EXEC_HOST,
                              max_num_iterations=a.iters, lambda0=a.lambda0)
    h = capi.Handle(cfg, B, N, H, W)
    # Every distinct alignment arrives from the generation pool (started before the GPU was touched), is handed to its slot(s) and kept
    # as keyframe + fp32 frame: 4 096 distinct ones would otherwise hold 15 GB of fp64 frames on the host.
    als, frames32 = [], []
    for i, job in enumerate(gen_jobs):
        x, f32 = job.result()
        for b in range(i, B, distinct):              # every slot owns its copy in HBM
            h.set_keyframe(b, x.norm_coord, x.grad, x.idp, x.weights, x.fx, x.fy, x.cx, x.cy)
            h.set_event_frame(b, f32)
        als.append(x); frames32.append(f32)
    gen_pool.shutdown()
    t_gen = time.perf_counter() - t_gen
    # The contract's timed region starts with the inputs resident: keyframes and frames are in HBM, as any frame writer of the library
    # leaves them (4x4 tiles).  What the timed steps launch is what the library launches for a frame it has not solved before — see
    # the module docstring; the strip copies are NOT made (the resident-frames leg below makes them, outside the timed region).
    gc.collect(); gc.disable()                       # (see the note at the warm-up loop below)
    p0 = np.stack([als[b % distinct].p0 for b in range(B)])
    q0 = np.stack([als[b % distinct].q0 for b in range(B)])
    v0 = np.stack([als[b % distinct].v0 for b in range(B)])
    dev = torch.device("cuda", local_rank) if ((world > 1 and backend == "nccl") or forced) else None

    res_buf = np.empty((B, 16))                     # the step's result table, allocated once (a fresh 512 KB array per step is an mmap + page faults)

    def step():
        h.set_states(0, p0, q0, v0)                  # same start every step (host-side, 104 B per slot)
        h.optimize_batch(0, 0, B, sync=True)
        return h.results(0, B, out=res_buf)

    gatherer = batchmod.ResultGatherer(total, device=dev, to_host=(rank == 0), force=forced)    # buffers, stream and event allocated once

    def gather(local):
        gatherer.start(local)
        return gatherer.finish()

    def step_sharded(prev_local):
        """Several GPUs: the all-gather of step k-1 (RCCL, on the gatherer's own stream) overlaps the solve of step k (library stream)."""
        h.set_states(0, p0, q0, v0)
        h.optimize_batch(0, 0, B, sync=False)
        if prev_local is not None:
            gatherer.start(prev_local)
        h.sync()
        table = gatherer.finish() if prev_local is not None else None
        return h.results(0, B), table

    # The timed region measures the library, not the interpreter: with torch imported a full (generation-2) collection of CPython's cyclic
    # garbage collector walks several 10^5 objects — 40-150 ms in this process, dozens of headline steps — whenever its allocation counters
    # happen to trip (seen as ONE 74.7 ms step in 2 of 5 runs of the strong-scaling block below).  Collect now — BEFORE the warm-up, so that
    # no idle gap separates warm-up and timed steps — and keep the collector off until the clock stops.  (The collection itself sits in
    # front of prepare_frames above, so that the GPU is busy from there to the last timed step.)
    # the frames' REAL first solve (untimed, not one of the --warmup steps): the library picks the kernel by its own rule ...
    table = step()
    first_solve = h.last_launch() if a.exec_ == "device" else None
    # ... and the handle keeps to that layout from here on (the same frames are solved again every step; left alone the library would
    # make their strip copies at the second solve — the re-solve regime, reported as value_resident_frames)
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")
    for _ in range(a.warmup):
        table = step()
        if world > 1 or forced:
            table = gather(table)
    if world > 1 or forced:
        dist.barrier()
    torch.cuda.synchronize()
    dev_us = []
    t0 = time.perf_counter()
    if world == 1 and not forced:
        for _ in range(a.steps):
            table = step()
            dev_us.append(h.info(0)["device_time_us"])
    else:
        prev = None
        for _ in range(a.steps):
            ts_ = time.perf_counter()
            prev, t_prev = step_sharded(prev)
            if t_prev is not None:
                table = t_prev
            dev_us.append(h.info(0)["device_time_us"])
            if os.environ.get("EDS_BENCH_DEBUG"):
                sys.stderr.write(f"[bench] rank {rank} step {1e3 * (time.perf_counter() - ts_):.3f} ms, kernel {dev_us[-1] * 1e-3:.3f} ms\n")
        t_last = gather(prev)                        # the last step's results: every step's gather ends inside the timed region
        if t_last is not None:
            table = t_last
    torch.cuda.synchronize()
    if world > 1 or forced:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if table is res_buf:
        table = table.copy()                         # (later legs compare their tables with this one)
    if world > 1 or forced:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dev is not None else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / a.steps
    iters_done = float(np.mean(table[:, 14])) if rank == 0 else 0.0      # only rank 0 holds the gathered table
    value = total * iters_done / (ms_per_step * 1e-3)

    strong = None
    if a.exec_ == "device" and a.solver == "lm6" and not a.no_configs:           # every rank takes part
        strong = strong_scaling_config4(capi, synth, batchmod, a, rank, world, local_rank, dev, forced, dist, torch)

    out = None
    if rank == 0:
        # residual/Jacobian passes per solve: LM6 = initial linearisation + one per iteration when the kernel keeps the accepted pose's
        # residuals in registers (eds_fused6_kernel with register-resident points: second template argument > 0); the streaming
        # kernels and the host-driven loop add a final residual pass; GN6 = one per iteration + the final residual pass.  Which
        # kernel ran is what the LIBRARY says it launched (eds_trk_last_launch), not a copy of its selection rule.
        launch = h.last_launch() if a.exec_ == "device" else None
        in_regs = False
        if launch and launch["kernel"].startswith("eds_fused6_kernel<"):
            in_regs = int(launch["kernel"].split("<")[1].split(",")[1]) > 0
        passes = a.iters + (1 if (a.solver == "lm6" and in_regs) else (2 if a.solver == "lm6" else 1))
        per_pt = BYTES_RESJAC[a.sampling] + BYTES_REDUCE
        mm = BYTES_MUST_MOVE[a.sampling]
        roof = None
        if a.exec_ == "device":
            k_ms = float(np.mean(dev_us)) * 1e-3
            roof = physical_roofline(launch["kernel"], k_ms, B * N * passes, per_pt, mm, a, extra={
                "frame_layout": {0: "row-major", 1: "4x4 tiles", 2: "strips"}.get(launch["layout"], "?"),
                "first_solve_kernel": first_solve["kernel"], "timed_kernel_is_first_solve_kernel": bool(first_solve["kernel"] == launch["kernel"]),
                "note": f"achieved / frac: physical bytes through the fabric per launch / kernel time / 8 TB/s; *_credit_8d: SURVEY 8d credit, {per_pt} B per "
                        f"point-evaluation x {B}x{N} points x {passes} passes per launch (counts J bytes a fused kernel never moves: NOT a bandwidth fraction); "
                        f"*_must_move: {mm} B per point-evaluation (point constants + taps only)"})
            launch_digest = {k: launch[k] for k in ("workgroups", "span_us", "mean_workgroup_us", "covered", "tail_idle_us")}
        # north_star's two streaming kernels, each on its own and COLD (1 GiB streamed through the caches in front of every repetition:
        # eds_trk_bench_kernel_cold) — back to back the reduction reads planes the previous kernel just left in the Infinity Cache
        rj_ms = h.bench_kernel_cold(0, B, ncols=6, which=0, reps=10)
        red_ms = h.bench_kernel_cold(0, B, ncols=6, which=1, reps=10)
        rj_warm_ms = h.bench_eval(0, B, ncols=6, with_reduction=False, reps=20)
        both_ms = h.bench_eval(0, B, ncols=6, with_reduction=True, reps=20)
        roof_rj = physical_roofline("eds_resjac_kernel" + ("<0" if a.sampling == "bicubic" else "<1"), rj_ms, B * N, BYTES_RESJAC[a.sampling],
                                    BYTES_RESJAC[a.sampling], a, extra={"point_evals_per_s": B * N / (rj_ms * 1e-3), "kernel_ms_back_to_back": rj_warm_ms,
                                                                         "resjac_plus_reduce_ms_back_to_back": both_ms,
                                                                         "note": f"stand-alone residual/Jacobian pass, cold; credit = must-move = {BYTES_RESJAC[a.sampling]} B per point "
                                                                                 "(it does write r and J)"})
        roof_rj["kernel"] = "eds_resjac_kernel"
        roof_red = physical_roofline("eds_reduce_kernel", red_ms, B * N, BYTES_REDUCE, BYTES_REDUCE, a, extra={
            "kernel_ms_behind_resjac": max(both_ms - rj_warm_ms, 1e-6),
            "note": f"{BYTES_REDUCE} B per point read once (r + six Jacobian planes, 16-byte loads, four points per lane), COLD: the planes are evicted in front "
                    "of every repetition; kernel_ms_behind_resjac = (resjac + reduce) - resjac back to back, where the Infinity Cache still holds what the "
                    "first kernel wrote"})
        roof_red["kernel"] = "eds_reduce_kernel<6, 4>"
        # what this box's HBM really streams ("confirm on the box ... and report the measured peak beside the nominal"): the
        # library's own plain stream kernel over 1 GiB (16 bytes per lane, grid-stride; csrc/eds_capi_solve.hip: eds_probe_kernel)
        try:
            hbm_probe = dict(h.hbm_probe(1 << 30, 10), note="library's own stream kernel over 1 GiB, 10 repetitions under HIP events on its stream: read-only pass, "
                                                            "and copy with read + write counted; nominal peak 8 000 GB/s, the guide's achievable ~6 300")
        except Exception as ex_:                     # never fail the bench over the probe
            hbm_probe = {"error": str(ex_)}
        if roof is None:
            roof = roof_rj
        pose_err = float(np.median([np.linalg.norm(table[b, 0:3] - als[b % distinct].p_true) for b in range(min(B, distinct))]))
        out = {
            "metric": "tracker_iterations_per_sec", "value": value, "unit": "iterations/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (projection + accumulation f64)", "data": "synthetic",
            "config": {"workload": f"{B} alignments/GPU x {N} points on {W}x{H} event frames, {a.iters} {a.solver.upper()} "
                                   f"iterations each ({a.sampling}), exec={a.exec_}; BASELINE.json configs[1] batched",
                       "alignments_per_gpu": B, "points": N, "frame": [H, W], "iterations": a.iters, "solver": a.solver,
                       "sampling": a.sampling, "exec": a.exec_, "parallelism": f"alignments sharded x{world}, all-gather of 16 doubles/alignment"},
            "alignments_per_s": total / (ms_per_step * 1e-3),
            "point_evals_per_s_in_solver": total * N * passes / (ms_per_step * 1e-3),
            "iterations_per_alignment": iters_done, "success_fraction": float(np.mean(table[:, 15])),
            "median_translation_error": pose_err,
            "roofline": roof, "roofline_resjac": roof_rj, "roofline_reduce": roof_red, "hbm_probe": hbm_probe,
        }
        h.set_knob("EDS_FUSED_LAYOUT", None)             # the library's own layout rule again from here on
        out["config"]["workload"] += ("; every timed solve samples its frame as a frame's FIRST solve does (4x4 tiles: the library's rule for a frame it has not "
                                      "solved before; Tracker.cpp:104 is one optimize per frame)")
        out["config"]["frame_regime"] = "new frame per solve (first-solve kernel)"
        if a.exec_ == "device":
            # RE-solved frames (round 4's headline regime): the strip copies of the frames are made (outside any timed region: their
            # cost is reported) and the same batch is solved from them
            h.prepare_frames(0, B)                          # (allocates the copies at its first call: not part of the conversion's cost)
            prep_ms = h.prepare_frames(0, B, force=True)     # the conversion of all B frames again, under HIP events
            r_ms, r_dev = [], []
            for k in range(6):
                h.set_states(0, p0, q0, v0)
                t1 = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); r_ms.append(1e3 * (time.perf_counter() - t1))
                r_dev.append(h.info(0)["device_time_us"] * 1e-3)
            rtab = h.results(0, B); rl = h.last_launch()
            its_r = float(np.mean(rtab[:, 14])); rk_ms_ = float(np.median(r_dev[1:])); rw_ms = float(np.median(r_ms[1:]))
            out["value_resident_frames"] = B * its_r / (rw_ms * 1e-3)
            out["resident_frames"] = {
                "iterations_per_s": B * its_r / (rw_ms * 1e-3), "ms_per_step": rw_ms, "kernel": rl["kernel"],
                "roofline": physical_roofline(rl["kernel"], rk_ms_, B * N * passes, per_pt, mm, a,
                                              extra={"frame_layout": {0: "row-major", 1: "4x4 tiles", 2: "strips"}.get(rl["layout"], "?")}),
                "max_abs_pose_difference_to_the_timed_kernel": float(np.abs(rtab[:, :7] - table[:B, :7]).max()) if table.shape[0] >= B else None,
                "frame_layout_prep": {"ms_for_batch": prep_ms, "us_per_frame": 1e3 * prep_ms / B},
                "with_strip_copies_made_for_every_frame_iterations_per_s": B * its_r / ((rk_ms_ + prep_ms) * 1e-3),
                "note": "NOT `value`: the same batch on frames that were solved before — the library has made their strip copies (one 128-byte line per "
                        "bicubic patch; csrc/eds_layout.hpp) by one conversion launch per frame set, which costs more than one solve gains and pays from "
                        "about the 12th solve of a frame: several keyframes / hypotheses against one frame, not the reference's call pattern"}
        if world == 1 and a.exec_ == "device" and a.solver == "lm6" and not a.no_configs:
            # The boundary takes HOST buffers (the reference hands `optimize` a std::vector<double>): what a batch costs when its frames
            # cross PCIe inside the timed region — never `value` (the contract's inputs are resident).  256 alignments, frames new for
            # the solve (tiles), fp64 host frames as the reference holds them, then fp32 ones (eds_trk_set_event_frame_f32).
            nhb = min(B, 256)
            hb = {}
            for nm, fr in (("fp64", [frames32[b % distinct].astype(np.float64) for b in range(nhb)]), ("fp32", [frames32[b % distinct] for b in range(nhb)])):
                for mode in ("batch_call", "one_call_per_frame"):
                    ts = []
                    for k in range(3):
                        t1 = time.perf_counter()
                        if mode == "batch_call":
                            h.set_event_frames(0, fr)                 # ABI 5: one call, narrowed on a few host threads, PCIe-bound
                        else:
                            for b in range(nhb):
                                h.set_event_frame(b, fr[b])
                        h.set_states(0, p0[:nhb], q0[:nhb], v0[:nhb])
                        h.optimize_batch(0, 0, nhb, sync=True)
                        ts.append(time.perf_counter() - t1)
                    its_hb = float(np.mean(h.results(0, nhb)[:, 14]))
                    hb[nm if mode == "batch_call" else nm + "_one_call_per_frame"] = {
                        "iterations_per_s": nhb * its_hb / float(np.median(ts)), "ms_per_batch": 1e3 * float(np.median(ts)),
                        "host_GB_per_s": nhb * fr[0].nbytes / float(np.median(ts)) / 1e9, "kernel": h.last_launch()["kernel"]}
                del fr
            hb["alignments"] = nhb
            hb["note"] = ("NOT `value`: every frame is handed over as a host buffer inside the timed region (eds_trk_set_event_frames: one call for all of them — or one "
                          "eds_trk_set_event_frame per alignment —, then one launch): the upload, not the solve, is what this measures")
            out["host_buffers_inclusive"] = hb
            h.prepare_frames(0, B)                        # (the legs below solve resident frames)
        if strong is not None:
            out["strong_scaling_config4"] = strong
        out["input_generation_s"] = t_gen
        if a.exec_ == "device":
            out["launch_digest"] = dict(launch_digest, note="the timed kernel's last launch, from its workgroups' own begin / end stamps: covered = sum of "
                                        "workgroup durations / (256 CUs x span); tail_idle_us = mean idle time of a CU behind its last workgroup")
        out["config"]["distinct_alignments"] = distinct
        if forced:
            out["forced_dist"] = True
        if world == 1 and a.exec_ == "device" and a.sampling == "bicubic" and not a.no_ref12:
            # the sampler north_star names (bilinear, 2x2 taps; the reference itself samples bicubically): informational
            h.set_config(capi.default_config(device=0, sampling=capi.SAMPLE_BILINEAR, solver=cfg.solver, exec=capi.EXEC_DEVICE,
                                             max_num_iterations=a.iters, lambda0=a.lambda0))
            b_ms, b_dev = [], []
            for k in range(4):
                h.set_states(0, p0, q0, v0)
                t1 = time.perf_counter()
                h.optimize_batch(0, 0, B, sync=True)
                b_ms.append(1e3 * (time.perf_counter() - t1))
                b_dev.append(h.info(0)["device_time_us"] * 1e-3)
            bt = h.results(0, B)
            per_pt_b = BYTES_RESJAC["bilinear"] + BYTES_REDUCE
            out["bilinear_sampling"] = {"iterations_per_s": B * float(np.mean(bt[:, 14])) / (float(np.median(b_ms[1:])) * 1e-3),
                                        "kernel_ms": float(np.median(b_dev[1:])),
                                        "roofline_frac": B * N * passes * per_pt_b / (float(np.median(b_dev[1:])) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "success_fraction": float(np.mean(bt[:, 15])), "kernel": h.last_launch()["kernel"]}
            h.set_config(cfg)
        if world == 1 and a.exec_ == "device" and N <= 2048 and not a.no_ref12:
            # the reference's own problem on the same batch (12 local parameters, Ceres-LM rules; one residual block, no loss):
            # informational, outside the timed region.  Main figures: frames new for the solve (what a first solve launches), as for
            # `value`; `resident_frames`: the same on frames whose strip copies exist.
            h.set_config(capi.default_config(device=0, sampling=cfg.sampling, solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE,
                                             max_num_iterations=a.iters, num_blocks=1))
            r_cred, r_mm = BYTES_REF12["credited"][a.sampling], BYTES_REF12["must_move"][a.sampling]

            def ref12_leg():
                w_ms, d_ms = [], []
                for k in range(4):
                    h.set_states(0, p0, q0, v0)
                    t1 = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); w_ms.append(1e3 * (time.perf_counter() - t1))
                    d_ms.append(h.info(0)["device_time_us"] * 1e-3)
                tab_ = h.results(0, B); it_ = float(np.mean(tab_[:, 14])); kms = float(np.median(d_ms[1:])); kern = h.last_launch()["kernel"]
                ev_ = it_ + 1.0              # evaluations per solve: the initial one + one per LM iteration (residuals kept as it goes)
                return {"lm_iterations_per_s": B * it_ / (float(np.median(w_ms[1:])) * 1e-3), "ms_per_step": float(np.median(w_ms[1:])), "kernel": kern,
                        "kernel_ms": kms, "iterations_per_alignment": it_, "success_fraction": float(np.mean(tab_[:, 15])),
                        "roofline": physical_roofline(kern, kms, B * N * ev_, r_cred, r_mm, a, extra={
                            "note": f"{r_cred} B credited / {r_mm} B must-move per point-evaluation x {B}x{N} points x {ev_:.2f} evaluations per solve"})}

            h.set_knob("EDS_FUSED_LAYOUT", "tiles")
            try:
                out["reference_problem"] = dict(ref12_leg(), solver="ref12", frame_regime="new frame per solve (first-solve kernel, 4x4 tiles)")
            finally:
                h.set_knob("EDS_FUSED_LAYOUT", None)
            out["reference_problem"]["resident_frames"] = ref12_leg()
            h.set_config(cfg)
        if world == 1 and a.exec_ == "device":
            out["latency"] = latency_block(capi, synth, als[0], a)
        if world == 1 and a.exec_ == "device" and a.sampling == "bicubic" and a.solver == "lm6" and not a.no_configs:
            out["configs"] = configs_block(capi, synth, a)
        if world == 1 and a.exec_ == "device" and not a.no_shared:
            # Informational: TWO batches in flight — a second handle (own stream, own copy of every frame) takes step k + 1 while step k
            # runs.  The host's work per step and, more, the TAIL of a launch (its last workgroups end up to one alignment's duration
            # apart: ~110 us of idle per CU in a 2.7 ms launch, 4-5 %) disappear under the other batch's kernel.  Not the headline: the
            # contract's step is one batch on one stream, and two overlapping launches stretch each other's event-measured duration.
            h2 = capi.Handle(cfg, B, N, H, W)
            for b in range(B):
                x = als[b % distinct]
                h2.set_keyframe(b, x.norm_coord, x.grad, x.idp, x.weights, x.fx, x.fy, x.cx, x.cy)
                h2.set_event_frame(b, frames32[b % distinct])
            h2.prepare_frames(0, B)
            hs = [h, h2]
            for hh in hs:                                  # warm-up, one at a time
                hh.set_states(0, p0, q0, v0); hh.optimize_batch(0, 0, B, sync=True)
            nsteps = 8
            t1 = time.perf_counter()
            hs[0].set_states(0, p0, q0, v0); hs[0].optimize_batch(0, 0, B, sync=False)
            same = True
            for k in range(nsteps):
                cur, nxt = hs[k % 2], hs[(k + 1) % 2]
                if k + 1 < nsteps:
                    nxt.set_states(0, p0, q0, v0); nxt.optimize_batch(0, 0, B, sync=False)
                cur.sync()
                tabk = cur.results(0, B)
                same = same and (bool(np.array_equal(tabk, table)) if a.solver == "lm6" else True)
            el = time.perf_counter() - t1
            out["two_batches_in_flight"] = {"iterations_per_s": nsteps * B * float(np.mean(tabk[:, 14])) / el, "ms_per_step": 1e3 * el / nsteps,
                                            "steps": nsteps, "identical_to_single_batch": same,
                                            "note": "NOT the headline: two handles / streams alternate, step k + 1 is launched while step k runs"}
            h2.close()
        if world == 1 and a.exec_ == "device" and not a.no_shared and B > 32:
            # A DIFFERENT workload, informational: the batch shape with 32 distinct alignments whose replicas SHARE their event frame
            # (eds_trk_share_event_frame: slot b holds alignment b % 32 and samples slot b % 32's storage) — several keyframes / pose
            # hypotheses against one frame.  The frames in flight then fit the L2s (TCC hit 0.97 against 0.08, profiles/r02_shared_frames_l2.txt)
            # and the same kernel runs without the fabric-bound gather: what is left is its instruction stream.  Last leg on this handle.
            nsh = min(32, distinct)
            for b in range(B):
                x = als[b % nsh]
                h.set_keyframe(b, x.norm_coord, x.grad, x.idp, x.weights, x.fx, x.fy, x.cx, x.cy)
                if b < nsh:
                    h.set_event_frame(b, frames32[b])
                else:
                    h.share_event_frame(b, b % nsh)
            h.prepare_frames(0, B)
            ps, qs, vs = (np.stack([getattr(als[b % nsh], k) for b in range(B)]) for k in ("p0", "q0", "v0"))
            s_ms, s_dev = [], []
            for k in range(5):
                h.set_states(0, ps, qs, vs)
                t1 = time.perf_counter()
                h.optimize_batch(0, 0, B, sync=True)
                s_ms.append(1e3 * (time.perf_counter() - t1))
                s_dev.append(h.info(0)["device_time_us"] * 1e-3)
            stab = h.results(0, B)
            out["shared_frames"] = {"iterations_per_s": B * float(np.mean(stab[:, 14])) / (float(np.median(s_ms[1:])) * 1e-3),
                                    "kernel_ms": float(np.median(s_dev[1:])), "distinct_frames": nsh, "kernel": h.last_launch()["kernel"],
                                    "replicas_bit_identical": bool(all(np.array_equal(stab[b], stab[b % nsh]) for b in range(nsh, B, 97))),
                                    "roofline_frac": B * N * passes * per_pt / (float(np.median(s_dev[1:])) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "NOT the headline workload: 32 distinct alignments, replicas of an alignment sample one shared frame instead of a copy each"}
    h.close()
    if rank == 0:
        # ---- parity of the timed batch against the CPU oracle (the checker, outside every timed region) ---------------------------
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        samp = po.BICUBIC if a.sampling == "bicubic" else po.BILINEAR
        nchk = min(distinct, 64 if distinct >= B else 32)      # every alignment distinct: 64 of them, one row each; replicated: 32, two replicas each
        worst, worst_row, acc_mismatch, nrows = 0.0, -1, 0, 0
        for d in range(nchk):
            x = als[d]
            x32 = synth.Alignment(**{**x.__dict__, "frame": frames32[d].astype(np.float64)})      # the frame as handed to the library
            o = po.Oracle(x32, sampling=samp)
            ref = o.pose6_lm(x.p0, x.q0, x.v0, iters=a.iters, lambda0=a.lambda0) if a.solver == "lm6" else o.pose6_gn(x.p0, x.q0, x.v0, iters=a.iters)
            rows = [b for b in range(d, B, distinct)][:2]                                         # first two replicas of this alignment (rank 0's shard)
            for b in rows:
                dist_se3 = po.se3_distance(table[b, 0:3], table[b, 3:7], ref["p"], ref["q"])
                if dist_se3 > worst:
                    worst, worst_row = dist_se3, b
                acc_mismatch += int(table[b, 14] != ref["iterations"])
                nrows += 1
        out["parity"] = {"parity_max_se3": worst, "rows_checked": nrows, "distinct_alignments": nchk, "tolerance": PARITY_TOL,
                         "worst_row": worst_row, "iteration_count_mismatches": acc_mismatch,
                         "against": "oracle pose6_lm/pose6_gn on the fp32-rounded frame, same start, same iteration budget"}
        out["parity_max_se3"] = worst
        if world == 1 and not a.no_cpu:
            out.update(cpu_baselines(als[:min(8, distinct)], a.iters, a.sampling, a.cpu_seconds))
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            if "reference_problem" in out:
                out["reference_problem"]["speedup_vs_cpu_ref12_all_threads"] = \
                    out["reference_problem"]["lm_iterations_per_s"] / max(v["lm_iterations_per_s"] for k, v in out["cpu_baseline_ref12"].items() if k.startswith("T"))
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed when the process exits — behind anything Python printed.
        # The JSON line has to be the LAST line on stdout: empty the C buffers first.
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
        if not (out["parity_max_se3"] <= PARITY_TOL) or out["parity"]["iteration_count_mismatches"]:
            sys.stderr.write(f"bench.py: PARITY FAILURE: max SE(3) distance to the oracle {out['parity_max_se3']:.3e} (tolerance {PARITY_TOL}), "
                             f"{out['parity']['iteration_count_mismatches']} iteration-count mismatches\n")
            sys.exit(3)


if __name__ == "__main__":
    main()
