"""A whole tracking step of 64 trackers (configs[4] end to end on one GPU), call by call: 64 event slices -> frames (one batched call), the 64
solves, the 64 MAD scales, getCoord / keyframe criterion of all 64.  `bench.py` reports the sum as latency.B64_step_ms."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = 64; nev = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
rng = np.random.default_rng(0)
als = [synth.make_alignment(5000 + i) for i in range(8)]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
for b in range(B):
    h.set_alignment(b, als[b % 8])
P0 = np.stack([als[b % 8].p0 for b in range(B)]); Q0 = np.stack([als[b % 8].q0 for b in range(B)]); V0 = np.stack([als[b % 8].v0 for b in range(B)])
sl = []
for b in range(B):
    fr = als[b % 8].frame
    strong = np.argwhere(np.abs(fr) > 0.25 * np.abs(fr).max())
    pk = strong[rng.integers(0, len(strong), nev)]
    sl.append((pk[:, 1].astype(np.uint16), pk[:, 0].astype(np.uint16), (fr[pk[:, 0], pk[:, 1]] > 0).astype(np.uint8)))
offs = (np.arange(B + 1) * nev).astype(np.int32)
cx = np.concatenate([s[0] for s in sl]); cy = np.concatenate([s[1] for s in sl]); cp = np.concatenate([s[2] for s in sl])
L = capi.lib()
def frames():
    assert L.eds_trk_build_event_frame_batch(h._h, 0, B, offs.ctypes.data_as(C.POINTER(C.c_int32)), cx.ctypes.data_as(C.POINTER(C.c_uint16)),
                                             cy.ctypes.data_as(C.POINTER(C.c_uint16)), cp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 0.5, 1, None) == 0
steps = [("frames", frames), ("states", lambda: h.set_states(0, P0, Q0, V0)), ("solve", lambda: h.optimize_batch(0, 0, B, sync=True)),
         ("MAD", lambda: h.loss_param_batch(capi.LP_MAD, 0, B)), ("getCoord", lambda: h.update_points_batch(0, B, False, want_points=False))]
acc = {k: [] for k, _ in steps}; tot = []
for rep in range(12):
    t0 = time.perf_counter()
    for k, f in steps:
        t = time.perf_counter(); f(); acc[k].append(time.perf_counter() - t)
    tot.append(time.perf_counter() - t0)
print(f"{B} trackers x {nev} events: " + "  ".join(f"{k} {np.median(v[2:])*1e6:.0f} us" for k, v in acc.items()) + f"  | step {np.median(tot[2:])*1e3:.3f} ms")
h.close()
