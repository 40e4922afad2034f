import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(5000)
solver = capi.SOLVER_REF12 if (len(sys.argv) > 1 and sys.argv[1] == "ref12") else capi.SOLVER_LM6
h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10), 1, 2000, 480, 640)
h.set_alignment(0, al)
ts=[]
for _ in range(6):
    t=time.perf_counter(); h.optimize(0, p=al.p0, q=al.q0, v=al.v0); ts.append(time.perf_counter()-t)
print("wall %.1f us kernel %.1f us" % (np.median(ts[2:])*1e6, h.info(0)["device_time_us"]))
