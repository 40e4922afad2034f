"""A/B of frame layouts and the LDS patch cache on the two hot kernels (one child process per variant)."""
import importlib, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
    B = int(sys.argv[2]); samp = int(sys.argv[3])
    als = [synth.make_alignment(5000 + i) for i in range(8)]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10, sampling=samp)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in range(B):
        a = als[b % len(als)]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % len(als)])
    p0 = np.stack([als[b % len(als)].p0 for b in range(B)]); q0 = np.stack([als[b % len(als)].q0 for b in range(B)]); v0 = np.stack([als[b % len(als)].v0 for b in range(B)])
    ts = []
    for _ in range(6):
        h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B); ts.append(h.info(0)["device_time_us"])
    rj = h.bench_eval(0, B, 6, False, 20); rjr = h.bench_eval(0, B, 6, True, 20)
    print(json.dumps({"fused_us": float(np.median(ts[2:])), "resjac_ms": rj, "resjac_reduce_ms": rjr}))
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for samp in (0, 1):
    for layout in ("tiled", "rowmajor"):
        for cache in (1, 0):
            env = dict(os.environ, EDS_FRAME_LAYOUT=layout, EDS_FUSED_CACHE=str(cache))
            out = subprocess.run([sys.executable, __file__, "child", str(B), str(samp)], env=env, capture_output=True, text=True)
            try:
                r = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception:
                print("FAILED", layout, cache, out.stderr[-500:]); continue
            print(f"samp={samp} layout={layout:8s} cache={cache} B={B}: fused {r['fused_us']:8.1f} us = {B*10/r['fused_us']:6.2f} M it/s | resjac {r['resjac_ms']*1e3:7.1f} us = {B*2000/r['resjac_ms']/1e6:6.2f} G pt/s | +reduce {r['resjac_reduce_ms']*1e3:7.1f} us", flush=True)
