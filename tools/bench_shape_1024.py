"""Headline batch (4 096 alignments x 2 000 points, strips): the rule's shape — 512 threads x 4 points per lane, two wavefronts per SIMD — against
1 024 threads x 2 points per lane (four wavefronts per SIMD, <= 128 VGPRs), forced with the handle's knobs."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
als = [synth.make_alignment(5000 + i) for i in range(32)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
for b in range(B):
    a = als[b % 32]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 32])
h.prepare_frames(0, B)
p0 = np.stack([als[b % 32].p0 for b in range(B)]); q0 = np.stack([als[b % 32].q0 for b in range(B)]); v0 = np.stack([als[b % 32].v0 for b in range(B)])
ref = None
for rep in range(2):
    for thr, ppt in ((None, None), ("1024", "2")):
        h.set_knob("EDS_FUSED_THREADS", thr); h.set_knob("EDS_FUSED_PPT", ppt)
        ks = []
        for k in range(8):
            h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B, sync=True); ks.append(h.info(0)["device_time_us"])
        tab = h.results(0, B)[:, :7].copy()
        if ref is None: ref = tab
        print(f"threads {thr or 'rule'} ppt {ppt or 'rule'}: kernel {np.median(ks[2:]):8.1f} us  {h.last_launch()['kernel']}  max|dpose| vs rule {np.abs(tab - ref).max():.1e}", flush=True)
h.close()
