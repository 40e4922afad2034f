"""Experiment: how fast are the kernels when the frame gather is cache-resident (points squeezed into a small window)?"""
import importlib, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for squeeze in (1.0, 0.1):
  for iters in (10, 40):
    als = [synth.make_alignment(5000 + i) for i in range(8)]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=iters)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in range(B):
        a = als[b % len(als)]
        h.set_keyframe(b, a.norm_coord * squeeze, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % len(als)])
    p0 = np.stack([als[b % len(als)].p0 for b in range(B)]); q0 = np.stack([als[b % len(als)].q0 for b in range(B)]); v0 = np.stack([als[b % len(als)].v0 for b in range(B)])
    ts = []
    for _ in range(5):
        h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B); ts.append(h.info(0)["device_time_us"])
    rj = h.bench_eval(0, B, 6, False, 20)
    acc = [int(h.trace(b)["accepted"].sum()) for b in range(4)]
    us = float(np.median(ts[2:])); passes = iters + 2
    print(f"squeeze={squeeze} iters={iters}: fused {us:8.1f} us = {B*iters/us:6.2f} M it/s ({us/passes*256/B:6.2f} us per pass per CU-slot) | resjac {rj*1e3:7.1f} us = {B*2000/rj/1e6:6.2f} G pt/s | accepted {acc}", flush=True)
    h.close()
