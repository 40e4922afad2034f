"""One alignment per launch (the reference's operating point, Tracker.cpp:104): kernel and wall time of `optimize`, median of many
solves from the same start.  python tools/bench_b1.py [ref12|lm6] [reps] [points] [iterations]   (EDS_HIP_LIB selects the build: tools/ab_latency.sh)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
which = sys.argv[1] if len(sys.argv) > 1 else "ref12"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
al = synth.make_alignment(5000, N=N)
solver = capi.SOLVER_REF12 if which == "ref12" else capi.SOLVER_LM6
h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=iters), 1, N, 480, 640)
h.set_alignment(0, al)
ker, wall = [], []
for rep in range(reps + 20):
    t0 = time.perf_counter()
    p, q, v, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    wall.append((time.perf_counter() - t0) * 1e6); ker.append(info["device_time_us"])
ker, wall = np.array(ker[20:]), np.array(wall[20:])
print(f"{which} B1 N={N}: kernel median {np.median(ker):.1f} us (p10 {np.percentile(ker, 10):.1f}, p90 {np.percentile(ker, 90):.1f})  "
      f"wall median {np.median(wall):.1f} us  iterations {info['num_iterations']}  {h.last_launch()['kernel']}")
# where the host's share goes: launch side (state -> pinned block, event/launch calls) and wait side (poll + collect), separately
ta, tb = [], []
for rep in range(120):
    h.set_state(0, al.p0, al.q0, al.v0)
    t0 = time.perf_counter(); h.optimize_batch(0, 0, 1, sync=False)
    t1 = time.perf_counter(); h.sync()
    t2 = time.perf_counter()
    ta.append((t1 - t0) * 1e6); tb.append((t2 - t1) * 1e6)
print(f"   split: launch call {np.median(ta[20:]):.1f} us, wait + collect {np.median(tb[20:]):.1f} us (kernel {h.info(0)['device_time_us']:.1f})")
# the same through ONE C call (eds_trk_bench_live): what a C++ caller pays, without the interpreter between the calls
r = h.bench_live(0, al.p0, al.q0, al.v0, reps=200)
print(f"   C loop: optimize {r['optimize_us']:.1f} us (kernel {r['kernel_us']:.1f})")
frame = np.ascontiguousarray(al.frame, dtype=np.float64); idp = np.ascontiguousarray(al.idp, dtype=np.float64)
r = h.bench_live(0, al.p0, al.q0, al.v0, idp=idp, frame=frame, method=capi.LP_MAD, reps=400)
print("   C loop, live sequence: " + "  ".join(f"{k[:-3]} {v:.1f}" for k, v in r.items()))
