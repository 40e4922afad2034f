"""Device loss scale (MAD) of a whole batch: time per call and per alignment."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
als = [synth.make_alignment(5000 + i) for i in range(8)]
for B in (64, 256, 4096):
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=3), B, 2000, 480, 640)
    for b in range(B): h.set_alignment(b, als[b % 8])
    h.optimize_batch(0, 0, B)
    h.loss_param_batch(capi.LP_MAD, 0, B)
    t = time.perf_counter()
    for _ in range(10): h.loss_param_batch(capi.LP_MAD, 0, B)
    dt = (time.perf_counter() - t) / 10
    print(f"MAD of {B:5d} x 2000 residuals: {dt*1e6:8.1f} us per call = {dt/B*1e6:6.3f} us per alignment", flush=True)
    h.close()
