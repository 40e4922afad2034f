"""REF12 (the reference problem on the device) over the shapes its launcher distinguishes: kernel time and LM iterations/s at 4 096 / 256 / 64 /
1 alignments of 2 000 points on 640x480, resident frames (strips) and frames solved for the first time (tiles).  A/B tool."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
als = [synth.make_alignment(5000 + i) for i in range(16)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
for B in (int(x) for x in (sys.argv[1:] or ["4096", "256", "64", "1"])):
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=1), B, 2000, 480, 640)
    for b in range(B):
        a = als[b % 16]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 16])
    p0 = np.stack([als[b % 16].p0 for b in range(B)]); q0 = np.stack([als[b % 16].q0 for b in range(B)]); v0 = np.stack([als[b % 16].v0 for b in range(B)])
    for layout in ("strips", "tiles"):
        h.set_knob("EDS_FUSED_LAYOUT", "tiles" if layout == "tiles" else None)
        if layout == "strips": h.prepare_frames(0, B)
        ts, ds = [], []
        for _ in range(8 if B > 64 else 40):
            h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); ts.append(time.perf_counter() - t); ds.append(h.info(0)["device_time_us"])
        tab = h.results(0, B)
        dig = float(np.abs(tab[:, :13]).sum())
        print(f"B={B:5d} {layout:6s}: kernel {np.median(ds[2:]):8.1f} us  wall {np.median(ts[2:])*1e6:8.1f} us  {B*float(np.mean(tab[:,14]))/np.median(ts[2:])/1e6:6.2f} M LM it/s  {h.last_launch()['kernel']}  digest {dig:.9f}", flush=True)
    h.close()
