"""LM6 batched throughput of the persistent kernels at several batch sizes (EDS_LM6_KERNEL=stream|resident)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
als = [synth.make_alignment(5000 + i) for i in range(8)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
for B in (int(x) for x in (sys.argv[1:] or ["1024", "3072"])):
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
    for b in range(B):
        a = als[b % 8]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 8])
    p0 = np.stack([als[b % 8].p0 for b in range(B)]); q0 = np.stack([als[b % 8].q0 for b in range(B)]); v0 = np.stack([als[b % 8].v0 for b in range(B)])
    ts, dev = [], []
    for _ in range(8):
        h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B); ts.append(time.perf_counter() - t); dev.append(h.info(0)["device_time_us"])
    it = h.info(0)["num_iterations"]
    print(f"B={B:5d}: wall {np.median(ts[2:])*1e3:8.3f} ms  kernel {np.median(dev[2:]):9.1f} us  -> {B*it/np.median(ts[2:])/1e6:7.3f} M iterations/s", flush=True)
    h.close()
