"""LM6 throughput when many alignments sample the SAME event frames (eds_trk_share_event_frame): 4 096 alignments, 8 distinct
frames; slot b shares the frame of slot b % 8, so every workgroup of XCD x samples ONE frame (1.26 MB: L2-resident).  Against the
bench's layout, where every slot owns a copy of its frame."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
als = [synth.make_alignment(5000 + i) for i in range(8)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
MODES = ("own copies", "shared, slot b -> frame b % 8", "shared, slot b -> frame (b // 512) % 8")
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1           # run one mode only (profiling)
ref = None
for mode in (MODES if only < 0 else (MODES[only],)):
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
    for b in range(B):
        k = b % 8 if mode != "shared, slot b -> frame (b // 512) % 8" else (b // 512) % 8
        a = als[k]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        if mode == "own copies" or b < 8: h.set_event_frame(b, fr[k if mode == "own copies" else b])
        else: h.share_event_frame(b, k)
    p0 = np.stack([als[(b % 8) if "//" not in mode else (b // 512) % 8].p0 for b in range(B)])
    q0 = np.stack([als[(b % 8) if "//" not in mode else (b // 512) % 8].q0 for b in range(B)])
    v0 = np.stack([als[(b % 8) if "//" not in mode else (b // 512) % 8].v0 for b in range(B)])
    ts, dev = [], []
    for _ in range(8):
        h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B); ts.append(time.perf_counter() - t); dev.append(h.info(0)["device_time_us"])
    tab = h.results(0, B)
    it = h.info(0)["num_iterations"]
    print(f"{mode:40s}: kernel {np.median(dev[2:]):9.1f} us -> {B*it/np.median(ts[2:])/1e6:7.3f} M iterations/s  (success {tab[:,15].mean():.3f})", flush=True)
    if mode == "own copies": ref = tab.copy()
    elif "//" not in mode and ref is not None: print("   identical to own copies:", np.array_equal(tab, ref))
    h.close()
