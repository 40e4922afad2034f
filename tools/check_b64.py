"""configs[4] on one GPU and its neighbours (33 .. 128 alignments per launch): team size x candidate groups, new frames (tiles) and re-solved ones."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
Bs = [int(x) for x in (sys.argv[1:] or ["24", "32", "48", "64", "96", "128"])]
als = [synth.make_alignment(5000 + i) for i in range(16)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
for layout in (("tiles",) if os.environ.get("CHECK_TILES_ONLY") else ("tiles", None)):
    for B in Bs:
        ref = None
        for team, groups in (("", ""), ("4", "1"), ("4", "2"), ("2", "1"), ("2", "2"), ("2", "4"), ("1", "1")):
            h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
            if team: h.set_knob("EDS_LM6_TEAM", team)
            if groups: h.set_knob("EDS_LM6_GROUPS", groups)
            if layout: h.set_knob("EDS_FUSED_LAYOUT", layout)
            for b in range(B):
                a = als[b % 16]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 16])
            p0 = np.stack([als[b % 16].p0 for b in range(B)]); q0 = np.stack([als[b % 16].q0 for b in range(B)]); v0 = np.stack([als[b % 16].v0 for b in range(B)])
            ts, ds = [], []
            for _ in range(40):
                h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); ts.append(time.perf_counter() - t); ds.append(h.info(0)["device_time_us"])
            tab = h.results(0, B)[:, :7].copy()
            if ref is None: ref = tab
            print(f"{layout or 'reuse':5s} B={B:4d} TEAM={team or 'rule':4s} GROUPS={groups or 'rule':4s}: wall {np.median(ts[5:])*1e6:7.1f} us  kernel {np.median(ds[5:]):7.1f} us  {h.last_launch()['kernel']:40s} max|dpose| vs rule {np.abs(tab-ref).max():.1e} flags {h.info(0)['flags']}", flush=True)
            h.close()
