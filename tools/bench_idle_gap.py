"""The live tracker's real call pattern has IDLE GAPS between calls (one optimize per event slice, slices every few milliseconds): wall
time and kernel time of one alignment's solve after sleeping 0 / 1 / 5 / 20 / 100 ms, LM6 and REF12 (4 blocks + Huber).  The back-to-back
figures of bench.py's `latency` block are the 0 ms row."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(1234)
for name, cfg in (("LM6", dict(solver=capi.SOLVER_LM6)), ("REF12", dict(solver=capi.SOLVER_REF12, num_blocks=4, loss_type=capi.LOSS_HUBER if hasattr(capi, "LOSS_HUBER") else 1, loss_param=0.5))):
    h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, max_num_iterations=10, **cfg), 1, 2000, 480, 640)
    h.set_alignment(0, al)
    for _ in range(20): h.optimize(0, 0, al.p0, al.q0, al.v0)
    for gap_ms in (0, 1, 5, 20, 100):
        w, k = [], []
        for _ in range(30 if gap_ms < 100 else 12):
            if gap_ms: time.sleep(gap_ms * 1e-3)
            t = time.perf_counter(); h.optimize(0, 0, al.p0, al.q0, al.v0); w.append(time.perf_counter() - t); k.append(h.info(0)["device_time_us"])
        print(f"{name:6s} idle gap {gap_ms:4d} ms: call {1e6*np.median(w):7.1f} us (max {1e6*np.max(w):7.1f})  kernel {np.median(k):7.1f} us (max {np.max(k):7.1f})", flush=True)
    h.close()
