"""One alignment per launch at the sizes of configs[0..3]: wall time of eds_trk_optimize and kernel time (HIP events)."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
for solver, name in ((capi.SOLVER_LM6, "lm6"), (capi.SOLVER_REF12, "ref12")):
    for (H, W, N) in ((480, 640, 2000), (480, 640, 4000), (720, 1280, 8000), (480, 640, 16000)):
        al = synth.make_alignment(2234, H, W, N)
        h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10), 1, N, H, W)
        h.set_alignment(0, al)
        ts = []
        for _ in range(8):
            t = time.perf_counter(); h.optimize(0, p=al.p0, q=al.q0, v=al.v0); ts.append(time.perf_counter() - t)
        print("%-5s %4dx%-4d N=%5d  wall %7.1f us  kernel %7.1f us  iterations %d" % (name, W, H, N, np.median(ts[2:]) * 1e6, h.info(0)["device_time_us"], h.info(0)["num_iterations"]))
        h.close()
