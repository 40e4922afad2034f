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

The timed steps solve every frame THE WAY A FRAME IS SOLVED THE FIRST TIME — the reference's call pattern is one optimize per event
frame (Tracker.cpp:104), and for such a solve the library samples the 4x4 tiles the frame was written in (its rule:
csrc/eds_strips.hip; the strip copies that make RE-solves of a frame faster cost more than one solve gains).  The bench lets the
library pick that kernel by itself on the frames' real first solve (an untimed step), checks that the timed steps launch the same
one, and pins it with the handle's EDS_FUSED_LAYOUT=tiles knob only because the SAME frames are solved again every step.

OUTPUT (round 6).  The LAST stdout line is a COMPACT record (< 4 KB: `compact_record`, size-tested on the CPU) — the contract's
keys plus
  roofline      the kernel of the timed region.  `achieved` / `frac` follow the contract: ALGORITHMIC bytes per launch (SURVEY 8d:
                140 B per point-evaluation bicubic = 112 B residual/Jacobian + 28 B reduction, x alignments x points x passes) / the
                kernel's duration measured live (HIP events on the library's own stream) / 8 TB/s.  `traffic` = bytes through the
                fabric per launch from the committed rocprofv3 PMC passes of this workload (profiles/traffic_*.json; gfx950
                correction of MI355X_MICROARCH.md: 2 x FETCH_SIZE + WRITE_SIZE) and `frac_physical` = that / time / 8 TB/s;
                `frac_must_move` counts what a fused kernel has to move (84 B per point-evaluation: constants + taps)
  cpu_baseline  the CPU oracle (a port: the reference needs Ceres and cannot be built here) running the SAME iterations on this
                host's cores, on a bounded sample
  reference_problem, roofline_resjac, roofline_reduce, latency, configs    digests of a few numbers each
Everything else — every leg in full (bench_detail.py) — goes to bench_detail.json next to this file (and to gpurun_out/).
"""
import argparse
import gc
import importlib
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_detail import (BYTES_MUST_MOVE, BYTES_REDUCE, BYTES_REF12, BYTES_RESJAC, HBM_PEAK_GBS, PARITY_TOL, _cpu_info, _under_profiler,  # noqa: E402,F401
                          _usable_cpus, cpu_baselines, detail_legs, pmc_traffic, ref12_leg, roofline_block, strong_scaling_config4, write_detail)

COMPACT_LIMIT = 4096             # bytes: the driver keeps an 8 KB tail of stdout; the record has to fit it with room to spare


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



def _sig(x, n=5):
    """Numbers of the compact record carry n significant digits (the full precision is in bench_detail.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d and k in d}


def compact_record(full):
    """The one line the driver parses, built from the detail record: the contract's keys, the roofline of the timed kernel, the CPU
    baseline, parity, and a few-number digest of each other leg.  Pure function of `full` (tests/test_bench_cli.py builds it from a
    canned record and checks the size)."""
    roof_keys = ("kernel", "bound", "peak", "unit", "kernel_ms", "achieved", "frac", "frac_physical", "frac_must_move", "traffic",
                 "algorithmic_bytes_per_launch", "traffic_source", "traffic_read_requests")
    rec = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    rec["config"] = _pick(full.get("config", {}), ("workload", "alignments_per_gpu", "points", "frame", "iterations", "solver", "sampling", "exec",
                                                    "parallelism", "frame_regime", "distinct_alignments"))
    if full.get("roofline"):
        rec["roofline"] = _pick(full["roofline"], roof_keys)
        rec["roofline"]["note"] = "frac = SURVEY 8d algorithmic bytes/time/peak; frac_physical = PMC traffic/time/peak"
    if full.get("cpu_baseline"):
        rec["cpu_baseline"] = _pick(full["cpu_baseline"], ("value", "unit", "cores", "kind", "one_core_value", "cpu_model"))
        rec["cpu_baseline"]["sample"] = full["cpu_baseline"].get("sample", "")[:160]
    for k in ("parity_max_se3", "value_resident_frames", "forced_dist"):
        if k in full:
            rec[k] = full[k]
    if full.get("parity"):
        rec["parity"] = _pick(full["parity"], ("rows_checked", "iteration_count_mismatches", "tolerance"))
    rp = full.get("reference_problem")
    if rp:
        rec["reference_problem"] = dict(_pick(rp, ("lm_iterations_per_s", "kernel_ms", "kernel")),
                                        **_pick(rp.get("roofline", {}), ("frac", "frac_physical", "frac_must_move", "traffic_read_requests")))
        if rp.get("resident_frames"):
            rec["reference_problem"]["resident_lm_iterations_per_s"] = rp["resident_frames"]["lm_iterations_per_s"]
    for k in ("roofline_resjac", "roofline_reduce"):
        if full.get(k):
            rec[k] = _pick(full[k], ("kernel", "kernel_ms", "frac", "frac_physical", "point_evals_per_s"))
    if full.get("cpu_baseline_fast"):
        rec["cpu_baseline_fast"] = _pick(full["cpu_baseline_fast"], ("value", "cores", "one_core_value"))
    if full.get("cpu_baseline_ref12"):
        rec["cpu_baseline_ref12"] = {k: v["lm_iterations_per_s"] for k, v in full["cpu_baseline_ref12"].items() if k.startswith("T") and isinstance(v, dict)}
    if full.get("latency"):
        rec["latency"] = _pick(full["latency"], ("B1_lm6_kernel_ms", "B1_lm6_c_ms", "B1_ref12_kernel_ms", "B1_ref12_c_ms", "live_call_ref12_c_ms",
                                                 "B64_kernel_ms", "B64_ms", "B64_c_ms", "B64_step_ms", "slice_ms"))
        sup = (full["latency"].get("live_call_ref12_c_steps_us") or {}).get("set_event_frame")
        if sup is not None:
            rec["latency"]["set_event_frame_us"] = sup
    if full.get("configs"):
        rec["configs"] = {}
        for name, c in full["configs"].items():
            r = c.get("roofline", {})
            rec["configs"][name] = dict(_pick(c, ("iterations_per_s", "ms_per_step")), kernel_ms=r.get("kernel_ms"), frac=r.get("frac"),
                                        frac_physical=r.get("frac_physical"), frac_must_move=r.get("frac_must_move"),
                                        parity_max_se3=(c.get("parity") or {}).get("parity_max_se3"))
    pd = full.get("point_distribution")
    if pd:
        rec["point_distribution"] = {k: dict(_pick(pd[k], ("iterations_per_s",)), kernel_ms=pd[k]["roofline"].get("kernel_ms"), frac=pd[k]["roofline"].get("frac"),
                                             frac_physical=pd[k]["roofline"].get("frac_physical")) for k in ("uniform", "edges") if k in pd}
    if full.get("strong_scaling_config4"):
        rec["strong_scaling_config4"] = _pick(full["strong_scaling_config4"], ("ms_per_step", "iterations_per_s", "alignments_per_gpu", "kernel_ms_rank0"))
    if full.get("hbm_probe"):
        rec["hbm_probe_GBps"] = _pick(full["hbm_probe"], ("read_GBps", "copy_GBps"))
    rec["detail"] = full.get("detail_files") or ["bench_detail.json"]
    rec = _sig(rec)
    line = json.dumps(rec, separators=(",", ":"))
    # never let an unforeseen string push the record past what the driver keeps: drop digests, least important first
    for k in ("hbm_probe_GBps", "strong_scaling_config4", "point_distribution", "cpu_baseline_ref12", "cpu_baseline_fast", "configs", "latency", "roofline_reduce", "roofline_resjac"):
        if len(line) <= COMPACT_LIMIT:
            break
        rec.pop(k, None)
        line = json.dumps(rec, separators=(",", ":"))
    return rec, line


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
    # the module docstring; the strip copies are NOT made (the resident-frames leg of bench_detail makes them, outside the timed region).
    # The timed region measures the library, not the interpreter: with torch imported a full (generation-2) collection of CPython's cyclic
    # garbage collector walks several 10^5 objects — 40-150 ms in this process, dozens of headline steps — whenever its allocation counters
    # happen to trip.  Collect now — BEFORE the warm-up — and keep the collector off until the clock stops.
    gc.collect(); gc.disable()
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
            roof = roofline_block(launch["kernel"], k_ms, B * N * passes, per_pt, mm, a, extra={
                "frame_layout": {0: "row-major", 1: "4x4 tiles", 2: "strips"}.get(launch["layout"], "?"),
                "first_solve_kernel": first_solve["kernel"], "timed_kernel_is_first_solve_kernel": bool(first_solve["kernel"] == launch["kernel"]),
                "note": f"achieved / frac: SURVEY 8d credit, {per_pt} B per point-evaluation x {B}x{N} points x {passes} passes per launch / kernel time / 8 TB/s (counts J "
                        f"bytes a fused kernel never moves); frac_physical: bytes through the fabric per launch (PMC passes) / kernel time / 8 TB/s; "
                        f"frac_must_move: {mm} B per point-evaluation (point constants + taps only)"})
            launch_digest = {k: launch[k] for k in ("workgroups", "span_us", "mean_workgroup_us", "covered", "tail_idle_us")}
        # north_star's two streaming kernels, each on its own and COLD (1 GiB streamed through the caches in front of every repetition:
        # eds_trk_bench_kernel_cold) — back to back the reduction reads planes the previous kernel just left in the Infinity Cache
        rj_ms = h.bench_kernel_cold(0, B, ncols=6, which=0, reps=10)
        red_ms = h.bench_kernel_cold(0, B, ncols=6, which=1, reps=10)
        rj_warm_ms = h.bench_eval(0, B, ncols=6, with_reduction=False, reps=20)
        both_ms = h.bench_eval(0, B, ncols=6, with_reduction=True, reps=20)
        roof_rj = roofline_block("eds_resjac_kernel" + ("<0" if a.sampling == "bicubic" else "<1"), rj_ms, B * N, BYTES_RESJAC[a.sampling],
                                 BYTES_RESJAC[a.sampling], a, extra={"point_evals_per_s": B * N / (rj_ms * 1e-3), "kernel_ms_back_to_back": rj_warm_ms,
                                                                      "resjac_plus_reduce_ms_back_to_back": both_ms,
                                                                      "note": f"stand-alone residual/Jacobian pass, cold; credit = must-move = {BYTES_RESJAC[a.sampling]} B per point "
                                                                              "(it does write r and J)"})
        roof_rj["kernel"] = "eds_resjac_kernel"
        roof_red = roofline_block("eds_reduce_kernel", red_ms, B * N, BYTES_REDUCE, BYTES_REDUCE, a, extra={
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
            "config": {"workload": f"{B} alignments/GPU x {N} points on {W}x{H} event frames, {a.iters} {a.solver.upper()} iterations each ({a.sampling}), "
                                   f"exec={a.exec_}; BASELINE.json configs[1] batched; every frame sampled as on its FIRST solve (4x4 tiles; Tracker.cpp:104)",
                       "alignments_per_gpu": B, "points": N, "frame": [H, W], "iterations": a.iters, "solver": a.solver,
                       "sampling": a.sampling, "exec": a.exec_, "parallelism": f"alignments sharded x{world}, all-gather of 16 doubles/alignment",
                       "frame_regime": "new frame per solve (first-solve kernel)", "distinct_alignments": distinct},
            "alignments_per_s": total / (ms_per_step * 1e-3),
            "point_evals_per_s_in_solver": total * N * passes / (ms_per_step * 1e-3),
            "iterations_per_alignment": iters_done, "success_fraction": float(np.mean(table[:, 15])),
            "median_translation_error": pose_err,
            "roofline": roof, "roofline_resjac": roof_rj, "roofline_reduce": roof_red, "hbm_probe": hbm_probe,
            "input_generation_s": t_gen,
        }
        h.set_knob("EDS_FUSED_LAYOUT", None)             # the library's own layout rule again from here on
        if a.exec_ == "device":
            out["launch_digest"] = dict(launch_digest, note="the timed kernel's last launch, from its workgroups' own begin / end stamps: covered = sum of "
                                        "workgroup durations / (256 CUs x span); tail_idle_us = mean idle time of a CU behind its last workgroup")
        if strong is not None:
            out["strong_scaling_config4"] = strong
        if forced:
            out["forced_dist"] = True
        if world == 1:
            ctx = types.SimpleNamespace(h=h, a=a, cfg=cfg, B=B, N=N, H=H, W=W, p0=p0, q0=q0, v0=v0, als=als, frames32=frames32, distinct=distinct,
                                        table=table, passes=passes)
            if a.exec_ == "device" and N <= 2048 and not a.no_ref12:
                # the reference's own problem on the same batch, frames new for the solve (before any strip copy exists), outside the timed region
                out["reference_problem"] = ref12_leg(capi, ctx, "tiles")
            detail_legs(capi, synth, ctx, out)
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
        out["detail_files"] = write_detail(out) or ["(bench_detail.json could not be written)"]
        rec, line = compact_record(out)
        # RCCL prints its version banner through C stdio, which is flushed when the process exits — behind anything Python printed.
        # The JSON line has to be the LAST line on stdout: empty the C buffers first.
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        sys.stderr.write(f"[bench] full record: {', '.join(out['detail_files'])} ({len(json.dumps(out))} bytes); compact line {len(line)} bytes\n")
        sys.stderr.flush()
        print(line, flush=True)
        if not (out["parity_max_se3"] <= PARITY_TOL) or out["parity"]["iteration_count_mismatches"]:
            sys.stderr.write(f"bench.py: PARITY FAILURE: max SE(3) distance to the oracle {out['parity_max_se3']:.3e} (tolerance {PARITY_TOL}), "
                             f"{out['parity']['iteration_count_mismatches']} iteration-count mismatches\n")
            sys.exit(3)


if __name__ == "__main__":
    main()
