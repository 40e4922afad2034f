"""Experiment: does the spatial order of the points matter for the gather-bound kernels?"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def morton(x, y):
    def part(v):
        v = v.astype(np.uint32) & 0xffff
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    return part(x) | (part(y) << 1)
for order in ("random", "raster20", "morton"):
    als = [synth.make_alignment(5000 + i) for i in range(8)]
    for a in als:
        if order == "raster20": key = (a.coord[:, 1] // 20) * 1000 + (a.coord[:, 0] // 20)
        elif order == "morton": key = morton(a.coord[:, 0] // 4, a.coord[:, 1] // 4)
        else: key = None
        if key is not None:
            idx = np.argsort(key, kind="stable")
            for f in ("norm_coord", "grad", "idp", "weights", "coord"): setattr(a, f, np.ascontiguousarray(getattr(a, f)[idx]))
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in range(B):
        a = als[b % 8]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 8])
    p0 = np.stack([als[b % 8].p0 for b in range(B)]); q0 = np.stack([als[b % 8].q0 for b in range(B)]); v0 = np.stack([als[b % 8].v0 for b in range(B)])
    ts = []
    for _ in range(5):
        h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B); ts.append(h.info(0)["device_time_us"])
    rj = h.bench_eval(0, B, 6, False, 20); us = float(np.median(ts[2:]))
    print(f"order={order:9s}: fused {us:8.1f} us = {B*10/us:6.2f} M it/s | resjac {rj*1e3:7.1f} us = {B*2000/rj/1e6:6.2f} G pt/s", flush=True)
    h.close()
