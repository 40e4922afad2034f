"""configs[4] on one GPU — 64 alignments x 2 000 points in ONE launch (the latency regime): kernel and wall time per
step for the team sizes the launcher can form (EDS_LM6_TEAM), LM6.  A/B tool for VERDICT r3 Next #6."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
Bs = [int(x) for x in (sys.argv[1:] or ["64", "96", "128"])]
als = [synth.make_alignment(5000 + i) for i in range(16)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
for B in Bs:
    ref = None
    for team, gather in (("", ""), ("", "lane"), ("4", ""), ("2", ""), ("2", "lane"), ("1", ""), ("1", "lane")):
        if team: os.environ["EDS_LM6_TEAM"] = team
        else: os.environ.pop("EDS_LM6_TEAM", None)
        if gather: os.environ["EDS_FUSED_GATHER"] = gather
        else: os.environ.pop("EDS_FUSED_GATHER", None)
        h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
        for b in range(B):
            a = als[b % 16]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 16])
        p0 = np.stack([als[b % 16].p0 for b in range(B)]); q0 = np.stack([als[b % 16].q0 for b in range(B)]); v0 = np.stack([als[b % 16].v0 for b in range(B)])
        ts, ds = [], []
        for _ in range(40):
            h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); ts.append(time.perf_counter() - t); ds.append(h.info(0)["device_time_us"])
        tab = h.results(0, B)[:, :7].copy()
        if ref is None: ref = tab
        print(f"B={B:4d} EDS_LM6_TEAM={team or 'auto':4s} gather={gather or 'rule':4s}: wall {np.median(ts[5:])*1e6:7.1f} us  kernel {np.median(ds[5:]):7.1f} us  {h.last_launch()['kernel']}  max|dpose| vs auto {np.abs(tab-ref).max():.1e}", flush=True)
        h.close()
