"""LM6 throughput at more than 2 048 points per alignment (configs[2]: 1280x720, 8 000 points) over batch sizes: which kernel wins."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
H, W, N = 720, 1280, int(sys.argv[1]) if len(sys.argv) > 1 else 8000
als = [synth.make_alignment(2234 + i, H, W, N) for i in range(4)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
for B in (int(x) for x in (sys.argv[2:] or ["32", "64", "256", "512"])):
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, N, H, W)
    for b in range(B):
        a = als[b % 4]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 4])
    p0 = np.stack([als[b % 4].p0 for b in range(B)]); q0 = np.stack([als[b % 4].q0 for b in range(B)]); v0 = np.stack([als[b % 4].v0 for b in range(B)])
    ts = []
    for _ in range(6):
        h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B); ts.append(time.perf_counter() - t)
    it = h.info(0)["num_iterations"]
    print(f"N={N} B={B:5d} {h.last_launch()['kernel'][17:]:>22s}: {np.median(ts[2:])*1e3:8.3f} ms -> {B*it/np.median(ts[2:])/1e6:7.3f} M iterations/s  ({B*N*(it+1)/np.median(ts[2:])/1e9:5.1f} G point-evaluations/s)", flush=True)
    h.close()
