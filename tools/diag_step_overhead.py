"""Where the headline step's host time goes (ms_per_step - kernel_ms ~ 85 us at 4 096 alignments): set_states / launch / wait / results."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
als = [synth.make_alignment(5000 + i) for i in range(16)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
for b in range(B):
    a = als[b % 16]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 16])
h.prepare_frames(0, B)
p0 = np.stack([als[b % 16].p0 for b in range(B)]); q0 = np.stack([als[b % 16].q0 for b in range(B)]); v0 = np.stack([als[b % 16].v0 for b in range(B)])
T = []
for k in range(25):
    t0 = time.perf_counter(); h.set_states(0, p0, q0, v0)
    t1 = time.perf_counter(); h.optimize_batch(0, 0, B, sync=False)
    t2 = time.perf_counter(); h.sync()
    t3 = time.perf_counter(); tab = h.results(0, B)
    t4 = time.perf_counter()
    T.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, h.info(0)["device_time_us"] * 1e-6))
T = np.median(np.array(T[5:]), axis=0) * 1e6
print(f"B={B}: set_states {T[0]:.1f} us, launch (optimize_batch, no sync) {T[1]:.1f} us, sync {T[2]:.1f} us (kernel {T[4]:.1f} us), results {T[3]:.1f} us; step {T[:4].sum():.1f} us, host share {T[:4].sum() - T[4]:.1f} us")
h.close()
