"""One alignment per launch (the reference's operating point, Tracker.cpp:104): kernel and wall time of `optimize`, median of many
solves from the same start.  python tools/bench_b1.py [ref12|lm6] [reps] [points]   (EDS_HIP_LIB selects the build: tools/ab_latency.sh)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
which = sys.argv[1] if len(sys.argv) > 1 else "ref12"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
al = synth.make_alignment(5000, N=N)
solver = capi.SOLVER_REF12 if which == "ref12" else capi.SOLVER_LM6
h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10), 1, N, 480, 640)
h.set_alignment(0, al)
ker, wall = [], []
for rep in range(reps + 20):
    t0 = time.perf_counter()
    p, q, v, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    wall.append((time.perf_counter() - t0) * 1e6); ker.append(info["device_time_us"])
ker, wall = np.array(ker[20:]), np.array(wall[20:])
print(f"{which} B1 N={N}: kernel median {np.median(ker):.1f} us (p10 {np.percentile(ker, 10):.1f}, p90 {np.percentile(ker, 90):.1f})  "
      f"wall median {np.median(wall):.1f} us  iterations {info['num_iterations']}  {h.last_launch()['kernel']}")
