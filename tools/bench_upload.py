"""Cost of handing one 640x480 fp64 event frame / one keyframe to the library (the live-call path of the C++ shim)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(1234)
h = capi.Handle(capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE), 1, al.N, al.H, al.W)
f64 = np.ascontiguousarray(al.frame); f32 = f64.astype(np.float32)
def med(f, n=30):
    for _ in range(3): f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return np.median(t) * 1e6
print(f"set_event_frame (fp64 {f64.nbytes/1e6:.2f} MB): {med(lambda: h.set_event_frame(0, f64)):8.1f} us")
print(f"set_event_frame_f32 ({f32.nbytes/1e6:.2f} MB):   {med(lambda: h.set_event_frame(0, f32)):8.1f} us")
print(f"set_keyframe ({al.N} points):              {med(lambda: h.set_keyframe(0, al.norm_coord, al.grad, al.idp, al.weights, al.fx, al.fy, al.cx, al.cy)):8.1f} us")
print(f"set_idepth:                                {med(lambda: h.set_idepth(0, al.idp)):8.1f} us")
print(f"optimize (REF12):                          {med(lambda: h.optimize(0, p=al.p0, q=al.q0, v=al.v0)):8.1f} us")
print(f"residuals + loss_param(MAD):               {med(lambda: (h.residuals(0), h.loss_param(0, capi.LP_MAD))):8.1f} us")
