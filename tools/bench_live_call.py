"""Where a live Tracker::optimize call (one event slice, Tracker.cpp:104-241) spends its time through the C ABI:
idepth re-upload (Tracker.cpp:167) -> event frame -> solve -> residuals (Tracker.cpp:223-230) -> loss scale (Tracker.cpp:233,
MAD, which reorders them: one call since round 3).  Wall time per step, median of the repeats."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(5000)
solver = capi.SOLVER_REF12 if (len(sys.argv) > 1 and sys.argv[1] == "ref12") else capi.SOLVER_LM6
h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10), 1, 2000, 480, 640)
h.set_alignment(0, al)
frame = np.ascontiguousarray(al.frame, dtype=np.float64)
idp = np.ascontiguousarray(al.idp, dtype=np.float64)
steps = ["set_idepth", "set_event_frame", "optimize", "residuals_and_loss"]
T = {k: [] for k in steps}
tot = []
for rep in range(30):
    t0 = time.perf_counter(); h.set_idepth(0, idp)
    t1 = time.perf_counter(); h.set_event_frame(0, frame)
    t2 = time.perf_counter(); h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    t3 = time.perf_counter(); h.residuals_and_loss(0, capi.LP_MAD, 0.0)
    t4 = time.perf_counter()
    for k, a, b in zip(steps, (t0, t1, t2, t3), (t1, t2, t3, t4)): T[k].append((b - a) * 1e6)
    tot.append((t4 - t0) * 1e6)
print("live call, %s: total %.1f us  kernel %.1f us" % ("ref12" if solver == capi.SOLVER_REF12 else "lm6", np.median(tot[5:]), h.info(0)["device_time_us"]))
print("  " + "  ".join("%s %.1f" % (k, np.median(T[k][5:])) for k in steps))
