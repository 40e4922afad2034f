"""Over-subscription stress of the team / candidate-group launches: T host threads, each with its own handle and stream, launch small
batches (team x group shapes: up to 128 workgroups per launch) at the same time, so the chip's 256 CUs are asked for up to T x 128
co-resident workgroups.  A team whose members cannot all become resident must time out (5 ms bound), fall back to one CU per
alignment and still return the result the same sequence gives when it runs alone: LM6 bit for bit, REF12 within 1e-9 (its fp64 LDS
atomics are not order-deterministic).  Prints the time-out count (allowed) and the disagreement count (must be 0).

    python tools/stress_oversubscribe.py [threads=8] [reps=40] [seed=77]"""
import importlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 77
H, W = 240, 320


def sequence(tid, reps, out, barrier=None):
    try:
        rng = np.random.default_rng(SEED + tid)
        solver = capi.SOLVER_LM6 if tid % 2 == 0 else capi.SOLVER_REF12
        B = int(rng.integers(1, 5))
        N = int(rng.integers(900, 2000))
        cfg = capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, num_blocks=1 + (tid % 3), max_num_iterations=8)
        h = capi.Handle(cfg, B, N, H, W)
        als = [synth.make_alignment(SEED * 10 + tid * 8 + b, H=H, W=W, N=N) for b in range(B)]
        for b, a in enumerate(als):
            h.set_alignment(b, a)
        p0 = np.stack([a.p0 for a in als]); q0 = np.stack([a.q0 for a in als]); v0 = np.stack([a.v0 for a in als])
        if barrier is not None:
            barrier.wait()
        res, flags, kernels = [], 0, set()
        for k in range(reps):
            h.set_states(0, p0, q0, v0)
            h.optimize_batch(0, 0, B)
            tab = np.array(h.results(0, B))
            flags += sum(1 for b in range(B) if h.info(b)["flags"] != 0)
            kernels.add(h.last_launch()["kernel"])
            res.append((tab, h.residuals(0).copy()))
        h.close()
        out[tid] = (solver, res, flags, kernels)
    except BaseException as e:
        out[tid] = e


def main():
    alone = {}
    for t in range(T):
        sequence(t, 1, alone)
        if isinstance(alone[t], BaseException):
            raise alone[t]
    together = {}
    barrier = threading.Barrier(T)
    th = [threading.Thread(target=sequence, args=(t, REPS, together, barrier)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
        if x.is_alive():
            print("A THREAD DID NOT FINISH"); os._exit(2)
    wall = time.perf_counter() - t0
    bad = timeouts = 0
    for t in range(T):
        if isinstance(together[t], BaseException):
            raise together[t]
        solver, res, flags, kernels = together[t]
        timeouts += flags
        ref_tab, ref_r = alone[t][1][0]
        for tab, r in res:
            if solver == capi.SOLVER_LM6:
                ok = np.array_equal(tab, ref_tab) and np.array_equal(r, ref_r)
            else:
                ok = np.allclose(tab[:, :13], ref_tab[:, :13], rtol=0, atol=1e-9) and np.array_equal(tab[:, 14:], ref_tab[:, 14:]) and np.allclose(r, ref_r, rtol=0, atol=1e-9)
            bad += 0 if ok else 1
        print(f"thread {t}: {'LM6' if solver == capi.SOLVER_LM6 else 'REF12'} B={ref_tab.shape[0]} kernels {sorted(kernels)} flagged alignments {flags}")
    print(f"{T} threads x {REPS} launches in {wall:.2f} s; alignments reporting a team time-out / pause: {timeouts}; DISAGREEMENTS: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
