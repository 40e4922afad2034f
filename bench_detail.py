#!/usr/bin/env python3
"""Everything bench.py measures BESIDE its headline, and the helpers both share (roofline arithmetic, CPU detection).

bench.py times the headline (4 096 alignments per GPU, frames new for the solve), builds the compact record the driver parses and
calls `detail_legs` here for the rest: re-solved frames, host-buffer-inclusive batches, the bilinear sampler, the reference's own
12-parameter problem, the latency regime, the other BASELINE.json configs, two batches in flight, shared frames.  All of it lands in
`bench_detail.json` next to bench.py (and in gpurun_out/ when that directory exists); only a digest of a few numbers rides in the
compact line.  The oracle (oracle/pyoracle.py) is imported here only as the checker and as the timed CPU baseline.
"""
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E nominal (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 achievable)
BYTES_RESJAC = {"bicubic": 112, "bilinear": 64}      # SURVEY §8d, per point-evaluation
BYTES_REDUCE = 28
BYTES_MUST_MOVE = {"bicubic": 84, "bilinear": 36}    # a fused kernel: 20 B of point constants + the taps; no J planes written or re-read
BYTES_REF12 = {"credited": {"bicubic": 196, "bilinear": 148},      # §8d 12-DoF: reads 28 B + taps, writes r + J[12] = 52 B, reduction reads 52 B
               "must_move": {"bicubic": 92, "bilinear": 44}}       # 28 B of point constants + the taps
PARITY_TOL = 1e-4                                    # SE(3) distance to the oracle's solved pose


def _match_kernel(kernels, kernel_prefix):
    """(the profiler prints every template argument — "eds_fused6_kernel<0, 4, 512, 1, 1, 1>" with the defaulted GROUPS —, the library's
    eds_trk_last_launch only those that differ from the default: match the name, or the name continued by further arguments)"""
    stem = kernel_prefix[:-1] if kernel_prefix.endswith(">") else kernel_prefix
    best = None
    for k, v in kernels.items():
        if (k == kernel_prefix or k.startswith(stem + ",") or (not kernel_prefix.endswith(">") and k.startswith(stem))) \
                and v.get("fetch_kb") is not None and v.get("write_kb") is not None:
            best = v
    return best


def pmc_traffic(kernel_prefix, a, workload=None):
    """Bytes through the fabric per launch from the committed rocprofv3 PMC passes (profiles/traffic_*.json, produced by
    tools/profile.sh + tools/summarise_profile.py in separate --pmc runs) — only when they were taken on exactly this workload;
    otherwise None.  Corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies a 128-byte request at 64 bytes, so
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  `workload` = None: the headline's bench_config must equal `a`; a name: the
    file's "workloads" section of that name (the config / latency legs, profiled by tools/profile_legs.sh with the leg's own
    fixed shape) — the newest file that holds the kernel wins."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if workload is None:
            c = t.get("bench_config", {})
            if (c.get("alignments_per_gpu"), c.get("points"), c.get("iterations"), c.get("solver"), c.get("sampling"), c.get("exec")) != \
                    (a.batch, a.points, a.iters, a.solver, a.sampling, a.exec_) or c.get("frame") != [a.height, a.width]:
                continue
            kernels = t.get("kernels", {})
        else:
            w = t.get("workloads", {}).get(workload)
            if not w or w.get("iterations") != a.iters or w.get("sampling") != a.sampling:
                continue
            kernels = w.get("kernels", {})
        v = _match_kernel(kernels, kernel_prefix)
        if v:
            best = {"bytes": (2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0, "raw_bytes": (v["fetch_kb"] + v["write_kb"]) * 1024.0,
                    "read_requests": (v.get("l2") or {}).get("TCC_EA0_RDREQ_sum"), "profiled_avg_us": v.get("avg_us"),
                    "source": os.path.relpath(f, ROOT)}
    return best


def roofline_block(kernel, k_ms, units, per_unit_credit, per_unit_must_move, a, workload=None, extra=None):
    """The roofline block of one kernel.  units = point-evaluations per launch.
      achieved / frac        the contract's figure: ALGORITHMIC bytes per launch (SURVEY 8d's per-unit credit x units) / the kernel
                             time measured live / 8 TB/s.  The credit counts J bytes a fused kernel never moves, so this is a
                             crediting convention, not a bandwidth (it may exceed 1 for a kernel that moves less than it is credited);
      frac_must_move         what a fused kernel has to move (point constants + taps);
      traffic, frac_physical bytes through the fabric per launch from the committed PMC passes of this workload, and that / time /
                             8 TB/s (never above 1); null when no pass of the workload is committed."""
    t = pmc_traffic(kernel, a, workload)
    cred = units * per_unit_credit / (k_ms * 1e-3) / 1e9
    mm = units * per_unit_must_move / (k_ms * 1e-3) / 1e9
    r = {"kernel": kernel, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": k_ms,
         "achieved": cred, "frac": cred / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": units * per_unit_credit,
         "achieved_must_move": mm, "frac_must_move": mm / HBM_PEAK_GBS, "must_move_bytes_per_launch": units * per_unit_must_move,
         "traffic": None, "achieved_physical": None, "frac_physical": None, "traffic_source": None}
    if t:
        ach = t["bytes"] / (k_ms * 1e-3) / 1e9
        r.update({"traffic": t["bytes"], "achieved_physical": ach, "frac_physical": ach / HBM_PEAK_GBS, "traffic_source": t["source"],
                  "traffic_read_requests": t["read_requests"], "traffic_profiled_kernel_us": t["profiled_avg_us"],
                  "traffic_over_algorithmic": t["bytes"] / (units * per_unit_credit), "traffic_over_must_move": t["bytes"] / (units * per_unit_must_move)})
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
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")       # the reference's pattern is a NEW frame per call: sample it as a frame's first solve does (no strip copies)
    per_pt, mm = BYTES_RESJAC[a.sampling] + BYTES_REDUCE, BYTES_MUST_MOVE[a.sampling]
    out["B1_lm6_ms"] = med(lambda: h.optimize(0, p=al.p0, q=al.q0, v=al.v0))
    out["B1_lm6_kernel_ms"] = h.info(0)["device_time_us"] * 1e-3
    out["B1_lm6_roofline"] = roofline_block(h.last_launch()["kernel"], out["B1_lm6_kernel_ms"], al.N * (a.iters + 1), per_pt, mm, a, workload="b1_lm6",
                                            extra={"note": "credit / must-move: the sequential solver's 11 passes; the candidate groups evaluate more poses than that (physical traffic shows it)"})
    # the same calls looped INSIDE the library (eds_trk_bench_live): what a C++ caller pays — the figures above include the Python
    # binding's own work around every call (argument conversion, ~10 us)
    out["B1_lm6_c_ms"] = h.bench_live(0, al.p0, al.q0, al.v0, reps=100)["optimize_us"] * 1e-3
    h.set_config(capi.default_config(sampling=samp, solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, num_blocks=1))
    out["B1_ref12_ms"] = med(lambda: h.optimize(0, p=al.p0, q=al.q0, v=al.v0))
    out["B1_ref12_kernel_ms"] = h.info(0)["device_time_us"] * 1e-3
    out["B1_ref12_roofline"] = roofline_block(h.last_launch()["kernel"], out["B1_ref12_kernel_ms"], al.N * (h.info(0)["num_iterations"] + 1),
                                              BYTES_REF12["credited"][a.sampling], BYTES_REF12["must_move"][a.sampling], a, workload="b1_ref12")
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
    bb = h.bench_batch(P0, Q0, V0, reps=100)                 # the same step looped inside ONE C call (what a C++ caller pays)
    out["B64_c_ms"] = bb["step_us"] * 1e-3
    out["B64_c_steps_us"] = {k: round(v, 1) for k, v in bb.items()}

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


def _rounded(synth, x):
    """The alignment with its frame as the library holds it (fp32)."""
    return synth.Alignment(**{**x.__dict__, "frame": np.ascontiguousarray(x.frame, dtype=np.float32).astype(np.float64)})


class Leg:
    """One measured workload beside the headline: `step()` is one pass of it (what tools/run_leg.py repeats under the profiler's
    PMC passes, so that profiles/traffic_*.json holds physical bytes of exactly this shape), `finish()` tears it down."""

    def __init__(self, name, step, finish, **kw):
        self.name, self.step, self.finish = name, step, finish
        self.__dict__.update(kw)


def setup_config2(capi, synth, a, resident=False):
    """configs[2] batched: 256 alignments of 1280x720 / 8 000 points, per-point Huber at 1.345 MAD of the start residuals."""
    from concurrent.futures import ThreadPoolExecutor
    B2, N2, H2, W2, D2 = 256, 8000, 720, 1280, 8
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
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

    def step():
        h.set_states(0, P0, Q0, V0); h.optimize_batch(0, 0, B2, sync=True)

    def to_resident():
        h.set_knob("EDS_FUSED_LAYOUT", None)
        h.prepare_frames(0, B2)                     # frames whose strip copies exist (re-solved frames)
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")         # as for the headline: every solve as a frame's first solve (the library's rule for a new frame)
    if resident:
        to_resident()
    return Leg("config2_resident" if resident else "config2", step, h.close, h=h, als=als, tau=tau, B=B2, N=N2, H=H2, W=W2, D=D2, to_resident=to_resident)


def setup_config3(capi, synth, a):
    """configs[3] batched: 64 four-level pyramids, 2 000 .. 16 000 points, one launch per level for all of them."""
    from concurrent.futures import ThreadPoolExecutor
    counts, B3, D3 = [16000, 8000, 4000, 2000], 64, 8
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
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
    return Leg("config3", lambda: pyr.optimize_batch(P0, Q0, V0), pyr.close, pyr=pyr, als=als, counts=counts, B=B3, D=D3)


def setup_config4(capi, synth, a):
    """configs[4] on ONE GPU: 64 alignments of configs[1] (seeds 5000 + b), one launch."""
    from concurrent.futures import ThreadPoolExecutor
    B4 = 64
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
        als = list(pool.map(lambda b: synth.make_alignment(5000 + b, H=a.height, W=a.width, N=a.points), range(B4)))
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0), B4, a.points, a.height, a.width)
    for b in range(B4):
        h.set_alignment(b, als[b])
    P0 = np.stack([x.p0 for x in als]); Q0 = np.stack([x.q0 for x in als]); V0 = np.stack([x.v0 for x in als])

    def step():
        h.set_states(0, P0, Q0, V0); h.optimize_batch(0, 0, B4, sync=True)
    return Leg("config4_one_gpu", step, h.close, h=h, als=als, B=B4)


def setup_b1(capi, synth, a, solver, al=None, num_blocks=1):
    """The reference's call pattern: ONE alignment per call (Tracker.cpp:104), frames new for the solve."""
    al = al if al is not None else synth.make_alignment(5000, H=a.height, W=a.width, N=a.points)
    samp = capi.SAMPLE_BICUBIC if a.sampling == "bicubic" else capi.SAMPLE_BILINEAR
    kw = dict(lambda0=a.lambda0) if solver == "lm6" else dict(num_blocks=num_blocks)
    h = capi.Handle(capi.default_config(sampling=samp, solver=capi.SOLVER_LM6 if solver == "lm6" else capi.SOLVER_REF12, exec=capi.EXEC_DEVICE,
                                        max_num_iterations=a.iters, **kw), 1, al.N, al.H, al.W)
    h.set_alignment(0, al)
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")
    return Leg("b1_" + solver, lambda: h.optimize(0, p=al.p0, q=al.q0, v=al.v0), h.close, h=h, al=al)


def setup_distribution(capi, synth, a, layout):
    """The headline's kernel on another POINT DISTRIBUTION (every figure was for uniformly random points): 1 024
    alignments of 640x480 / 2 000 points, 8 distinct, every slot its own frame, LM6 on new frames.  layout = "uniform"  or
    "edges" (synth.make_alignment: points strung along contours, as a gradient-selected keyframe of a real scene has them)."""
    from concurrent.futures import ThreadPoolExecutor
    Bd, Dd = 1024, 8
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
        als = list(pool.map(lambda i: synth.make_alignment(6200 + i, H=a.height, W=a.width, N=a.points, layout=layout), range(Dd)))
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, lambda0=a.lambda0), Bd, a.points, a.height, a.width)
    fr = [np.ascontiguousarray(x.frame, dtype=np.float32) for x in als]
    for b in range(Bd):
        x = als[b % Dd]
        h.set_keyframe(b, x.norm_coord, x.grad, x.idp, x.weights, x.fx, x.fy, x.cx, x.cy); h.set_event_frame(b, fr[b % Dd])
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")
    P0 = np.stack([als[b % Dd].p0 for b in range(Bd)]); Q0 = np.stack([als[b % Dd].q0 for b in range(Bd)]); V0 = np.stack([als[b % Dd].v0 for b in range(Bd)])

    def step():
        h.set_states(0, P0, Q0, V0); h.optimize_batch(0, 0, Bd, sync=True)
    return Leg("dist_" + layout, step, h.close, h=h, als=als, B=Bd, D=Dd, frames32=fr)


def distribution_block(capi, synth, a):
    """Uniform against edge-like point positions, same box, same batch shape: rate, kernel time, credited / must-move / physical fractions,
    8 rows of each against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    per_pt, mm = BYTES_RESJAC[a.sampling] + BYTES_REDUCE, BYTES_MUST_MOVE[a.sampling]
    out = {}
    for layout in ("uniform", "edges"):
        L = setup_distribution(capi, synth, a, layout)
        L.step(); L.step()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); L.step(); ts.append(time.perf_counter() - t0)
        tab = L.h.results(0, L.B); launch = L.h.last_launch(); k_ms = L.h.info(0)["device_time_us"] * 1e-3
        worst, mism = 0.0, 0
        for d in range(L.D):
            x = L.als[d]
            x32 = synth.Alignment(**{**x.__dict__, "frame": L.frames32[d].astype(np.float64)})
            ref = po.Oracle(x32).pose6_lm(x.p0, x.q0, x.v0, iters=a.iters, lambda0=a.lambda0)
            worst = max(worst, po.se3_distance(tab[d, 0:3], tab[d, 3:7], ref["p"], ref["q"])); mism += int(tab[d, 14] != ref["iterations"])
        wall = float(np.median(ts))
        out[layout] = {"iterations_per_s": L.B * float(np.mean(tab[:, 14])) / wall, "ms_per_step": 1e3 * wall, "kernel": launch["kernel"],
                       "roofline": roofline_block(launch["kernel"], k_ms, L.B * a.points * (a.iters + 1), per_pt, mm, a, workload="dist_" + layout),
                       "success_fraction": float(np.mean(tab[:, 15])),
                       "parity": {"rows_checked": L.D, "parity_max_se3": worst, "iteration_count_mismatches": mism, "tolerance": PARITY_TOL}}
        L.finish()
    out["edges_over_uniform"] = out["edges"]["iterations_per_s"] / out["uniform"]["iterations_per_s"]
    out["note"] = ("1 024 alignments x 2 000 points, LM6 on new frames, the headline's kernel: 'uniform' = SURVEY 8d's point positions, 'edges' = the same number of "
                   "points strung along contours (what a gradient-selected keyframe looks like): neighbouring patches share 128-byte lines, so the gather moves less")
    return out


LEGS = {"dist_uniform": lambda c, s, a: setup_distribution(c, s, a, "uniform"), "dist_edges": lambda c, s, a: setup_distribution(c, s, a, "edges"),
        "config2": lambda c, s, a: setup_config2(c, s, a), "config2_resident": lambda c, s, a: setup_config2(c, s, a, resident=True),
        "config3": setup_config3, "config4_one_gpu": setup_config4,
        "b1_lm6": lambda c, s, a: setup_b1(c, s, a, "lm6"), "b1_ref12": lambda c, s, a: setup_b1(c, s, a, "ref12")}


def _step_traffic(workload, a):
    """All solver launches of one step of a leg, from the committed PMC passes: (bytes per step, read requests per step, source)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json"))):
        try:
            w = json.load(open(f)).get("workloads", {}).get(workload)
        except Exception:
            continue
        if w and w.get("iterations") == a.iters and w.get("sampling") == a.sampling and w.get("bytes_per_step") is not None:
            best = (w["bytes_per_step"], w.get("read_requests_per_step"), os.path.relpath(f, ROOT))
    return best


def configs_block(capi, synth, a):
    """The other BASELINE.json configs as throughput + roofline + parity in the same record (each outside the headline's timed region):
    configs[2] batched (256 x 1280x720 / 8 000 points, per-point Huber at 1.345 MAD), configs[3] batched (64 four-level pyramids,
    2 000 .. 16 000 points, one launch per level for all of them), configs[4] (64 alignments of configs[1], one launch).  LM6,
    `--iters` iterations (per level), bicubic; >= 8 result rows of each against the CPU oracle (the checker: never inside a timing).
    Rooflines: credit / must-move arithmetic on the kernel time measured here + the PHYSICAL bytes of the same leg from the committed
    PMC passes (profiles/traffic_*.json "workloads", taken by tools/profile_legs.sh running exactly these set-ups)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    import np_pyramid_oracle as pyo
    per_pt, mm = BYTES_RESJAC["bicubic"] + BYTES_REDUCE, BYTES_MUST_MOVE["bicubic"]
    out = {}

    def timed(f, reps=5):
        f(); f()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter(); r = f(); t.append(time.perf_counter() - t0)
        return float(np.median(t)), r

    def roof(workload, kernel, points_passes, k_ms):          # points_passes = sum over the step's launches of alignments x points x passes
        r = roofline_block(kernel, k_ms, points_passes, per_pt, mm, a, workload=workload)
        if r["traffic"] is None:                               # several kernels per step (config3) or a name the profile lists otherwise: the step's total
            t = _step_traffic(workload, a)
            if t:
                ach = t[0] / (k_ms * 1e-3) / 1e9
                r.update({"traffic": t[0], "achieved_physical": ach, "frac_physical": ach / HBM_PEAK_GBS, "traffic_source": t[2], "traffic_read_requests": t[1],
                          "traffic_over_algorithmic": t[0] / (points_passes * per_pt), "traffic_over_must_move": t[0] / (points_passes * mm)})
        return r

    # ---- configs[2] ------------------------------------------------------------------------------------------------------------
    L = setup_config2(capi, synth, a)
    h, B2, N2, D2, tau = L.h, L.B, L.N, L.D, L.tau
    wall, _ = timed(L.step)
    tab = h.results(0, B2); launch = h.last_launch(); k_ms = h.info(0)["device_time_us"] * 1e-3
    L.to_resident()
    wall_r, _ = timed(L.step)
    launch_r = h.last_launch(); k_ms_r = h.info(0)["device_time_us"] * 1e-3
    its = float(np.mean(tab[:, 14])); passes = a.iters + 1
    worst, mism = 0.0, 0
    for d in range(D2):
        ref = po.Oracle(_rounded(synth, L.als[d])).pose6_lm(L.als[d].p0, L.als[d].q0, L.als[d].v0, iters=a.iters, lambda0=a.lambda0, huber_tau=tau)
        worst = max(worst, po.se3_distance(tab[d, 0:3], tab[d, 3:7], ref["p"], ref["q"])); mism += int(tab[d, 14] != ref["iterations"])
    out["config2"] = {"workload": f"{B2} alignments x {N2} points on {L.W}x{L.H}, {a.iters} LM6 iterations, per-point Huber tau = 1.345 MAD = {tau:.4g}; frames new for the solve",
                      "iterations_per_s": B2 * its / wall, "ms_per_step": 1e3 * wall, "kernel": launch["kernel"], "cus_per_alignment": launch["cus_per_alignment"],
                      "roofline": roof("config2", launch["kernel"], B2 * N2 * passes, k_ms), "success_fraction": float(np.mean(tab[:, 15])),
                      "resident_frames": {"iterations_per_s": B2 * its / wall_r, "ms_per_step": 1e3 * wall_r, "kernel": launch_r["kernel"],
                                          "roofline": roof("config2_resident", launch_r["kernel"], B2 * N2 * passes, k_ms_r)},
                      "parity": {"rows_checked": D2, "parity_max_se3": worst, "iteration_count_mismatches": mism, "tolerance": PARITY_TOL}}
    L.finish()

    # ---- configs[3] ------------------------------------------------------------------------------------------------------------
    L = setup_config3(capi, synth, a)
    counts, B3, D3 = L.counts, L.B, L.D
    wall, (P, Q, V, infos) = timed(L.step)
    its_l = [float(np.mean([infos[l][k]["num_iterations"] for k in range(B3)])) for l in range(len(counts))]
    k_ms = sum(infos[l][0]["device_time_us"] for l in range(len(counts))) * 1e-3
    worst, mism = 0.0, 0
    for d in range(D3):
        rp, rq, rv, per_level = pyo.track(po, synth, L.als[d], counts, [a.iters] * len(counts), solver="lm6")
        worst = max(worst, po.se3_distance(P[d], Q[d], rp, rq))
        mism += sum(int(infos[l][d]["num_iterations"] != per_level[l]["iterations"]) for l in range(len(counts)))
    out["config3"] = {"workload": f"{B3} coarse-to-fine pyramids, levels 80x60 .. 640x480 with {counts[::-1]} points, {a.iters} LM6 iterations per level, "
                                  f"one launch per level for all pyramids; frames new for the solve",
                      "iterations_per_s": B3 * sum(its_l) / wall, "pyramids_per_s": B3 / wall, "ms_per_step": 1e3 * wall,
                      "iterations_per_level_finest_first": its_l,
                      "kernel_ms_per_level_finest_first": [infos[l][0]["device_time_us"] * 1e-3 for l in range(len(counts))],
                      "roofline": roof("config3", "eds_fused6_kernel (one launch per level; the step's launches summed)", sum(B3 * n * (a.iters + 1) for n in counts), k_ms),
                      "parity": {"rows_checked": D3, "parity_max_se3": worst, "iteration_count_mismatches": mism, "tolerance": PARITY_TOL}}
    L.finish()

    # ---- configs[4] on one GPU -------------------------------------------------------------------------------------------------
    L = setup_config4(capi, synth, a)
    h, B4 = L.h, L.B
    wall, _ = timed(L.step, reps=20)
    tab = h.results(0, B4); launch = h.last_launch(); k_ms = h.info(0)["device_time_us"] * 1e-3
    worst, mism = 0.0, 0
    for d in range(0, B4, 8):
        ref = po.Oracle(_rounded(synth, L.als[d])).pose6_lm(L.als[d].p0, L.als[d].q0, L.als[d].v0, iters=a.iters, lambda0=a.lambda0)
        worst = max(worst, po.se3_distance(tab[d, 0:3], tab[d, 3:7], ref["p"], ref["q"])); mism += int(tab[d, 14] != ref["iterations"])
    out["config4_one_gpu"] = {"workload": f"{B4} alignments (seeds 5000..5063) x {a.points} points on {a.width}x{a.height}, {a.iters} LM6 iterations, ONE launch "
                                          f"(the 8-GPU config shards them 8 per GPU; this is all 64 on one)",
                              "iterations_per_s": B4 * float(np.mean(tab[:, 14])) / wall, "ms_per_step": 1e3 * wall, "kernel": launch["kernel"],
                              "cus_per_alignment": launch["cus_per_alignment"], "roofline": roof("config4_one_gpu", launch["kernel"], B4 * a.points * (a.iters + 1), k_ms),
                              "parity": {"rows_checked": len(range(0, B4, 8)), "parity_max_se3": worst, "iteration_count_mismatches": mism, "tolerance": PARITY_TOL}}
    L.finish()
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
                    "over the steps (MAX over ranks); a team of CUs that did not assemble within 5 ms is re-run on one CU per alignment and counted here"}


def ref12_leg(capi, c, layout):
    """The reference's own problem on the headline batch (12 local parameters, Ceres-LM rules; one residual block, no loss).
    layout = "tiles": frames new for the solve (what a first solve launches, as for `value`); None: frames whose strip copies exist."""
    h, a, B, N = c.h, c.a, c.B, c.N
    r_cred, r_mm = BYTES_REF12["credited"][a.sampling], BYTES_REF12["must_move"][a.sampling]
    h.set_config(capi.default_config(device=0, sampling=c.cfg.sampling, solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=a.iters, num_blocks=1))
    h.set_knob("EDS_FUSED_LAYOUT", layout)
    try:
        w_ms, d_ms = [], []
        for k in range(4):
            h.set_states(0, c.p0, c.q0, c.v0)
            t1 = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); w_ms.append(1e3 * (time.perf_counter() - t1))
            d_ms.append(h.info(0)["device_time_us"] * 1e-3)
        tab_ = h.results(0, B); it_ = float(np.mean(tab_[:, 14])); kms = float(np.median(d_ms[1:])); kern = h.last_launch()["kernel"]
        ev_ = it_ + 1.0              # evaluations per solve: the initial one + one per LM iteration (residuals kept as it goes)
        return {"lm_iterations_per_s": B * it_ / (float(np.median(w_ms[1:])) * 1e-3), "ms_per_step": float(np.median(w_ms[1:])), "kernel": kern,
                "kernel_ms": kms, "iterations_per_alignment": it_, "success_fraction": float(np.mean(tab_[:, 15])), "solver": "ref12",
                "frame_regime": "new frame per solve (first-solve kernel, 4x4 tiles)" if layout == "tiles" else "frames solved before (strip copies)",
                "roofline": roofline_block(kern, kms, B * N * ev_, r_cred, r_mm, a, extra={
                    "note": f"{r_cred} B credited / {r_mm} B must-move per point-evaluation x {B}x{N} points x {ev_:.2f} evaluations per solve"})}
    finally:
        h.set_knob("EDS_FUSED_LAYOUT", None)
        h.set_config(c.cfg)


def detail_legs(capi, synth, c, out):
    """Everything beside the headline that needs the headline's handle and inputs (rank 0, one GPU).  `c` carries them:
    h, a, cfg, B, N, H, W, p0, q0, v0, als, frames32, distinct, table, passes.  Fills `out` (the detail record) in place.
    Order matters: the legs leave the handle in the state the next one expects (strip copies made from `resident_frames` on)."""
    h, a, B, N, H, W, p0, q0, v0 = c.h, c.a, c.B, c.N, c.H, c.W, c.p0, c.q0, c.v0
    als, frames32, distinct, table, passes, cfg = c.als, c.frames32, c.distinct, c.table, c.passes, c.cfg
    per_pt, mm = BYTES_RESJAC[a.sampling] + BYTES_REDUCE, BYTES_MUST_MOVE[a.sampling]
    device = a.exec_ == "device"
    if device:
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
            "roofline": roofline_block(rl["kernel"], rk_ms_, B * N * passes, per_pt, mm, a,
                                       extra={"frame_layout": {0: "row-major", 1: "4x4 tiles", 2: "strips"}.get(rl["layout"], "?")}),
            "max_abs_pose_difference_to_the_timed_kernel": float(np.abs(rtab[:, :7] - table[:B, :7]).max()) if table.shape[0] >= B else None,
            "frame_layout_prep": {"ms_for_batch": prep_ms, "us_per_frame": 1e3 * prep_ms / B},
            "with_strip_copies_made_for_every_frame_iterations_per_s": B * its_r / ((rk_ms_ + prep_ms) * 1e-3),
            "note": "NOT `value`: the same batch on frames that were solved before — the library has made their strip copies (one 128-byte line per "
                    "bicubic patch; csrc/eds_layout.hpp) by one conversion launch per frame set, which costs more than one solve gains and pays from "
                    "about the 12th solve of a frame: several keyframes / hypotheses against one frame, not the reference's call pattern"}
    if device and a.solver == "lm6" and not a.no_configs:
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
    if device and a.sampling == "bicubic" and not a.no_ref12:
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
    if device and N <= 2048 and not a.no_ref12:
        out["reference_problem"]["resident_frames"] = ref12_leg(capi, c, None)
    if device:
        out["latency"] = latency_block(capi, synth, als[0], a)
    if device and a.sampling == "bicubic" and a.solver == "lm6" and not a.no_configs:
        out["configs"] = configs_block(capi, synth, a)
        out["point_distribution"] = distribution_block(capi, synth, a)
    if device and not a.no_shared:
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
        ref_tab = h.results(0, B).copy()
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
            same = same and (bool(np.array_equal(tabk, ref_tab)) if a.solver == "lm6" else True)
        el = time.perf_counter() - t1
        out["two_batches_in_flight"] = {"iterations_per_s": nsteps * B * float(np.mean(tabk[:, 14])) / el, "ms_per_step": 1e3 * el / nsteps,
                                        "steps": nsteps, "identical_to_single_batch": same,
                                        "note": "NOT the headline: two handles / streams alternate, step k + 1 is launched while step k runs"}
        h2.close()
    if device and not a.no_shared and B > 32:
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


def write_detail(out):
    """bench_detail.json next to bench.py, and a copy under gpurun_out/ when that scratch directory exists (it travels back from the GPU box)."""
    name = "bench_detail_profiled.json" if _under_profiler() else "bench_detail.json"      # (a profiled run never overwrites the plain run's record)
    paths = [os.path.join(ROOT, name)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", name))
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
            written.append(os.path.relpath(p, ROOT))
        except OSError:
            pass
    return written
