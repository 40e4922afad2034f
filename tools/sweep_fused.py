"""Sweep of the persistent solver: batch size x workgroup size -> device time per launch."""
import importlib, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
    B = int(sys.argv[2]); iters = int(sys.argv[3]); samp = int(sys.argv[4])
    als = [synth.make_alignment(5000 + i) for i in range(min(B, 8))]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=iters, sampling=samp)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in range(B):
        a = als[b % len(als)]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % len(als)])
    p0 = np.stack([als[b % len(als)].p0 for b in range(B)]); q0 = np.stack([als[b % len(als)].q0 for b in range(B)]); v0 = np.stack([als[b % len(als)].v0 for b in range(B)])
    ts = []
    for _ in range(8):
        h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B); ts.append(h.info(0)["device_time_us"])
    print(json.dumps({"B": B, "iters": iters, "us": float(np.median(ts[2:]))}))
    sys.exit(0)
for samp in (0, 1):
  for cache in (1, 0):
    for B in (1, 256, 1024, 4096):
        for iters in (10, 0):
            env = dict(os.environ, EDS_FUSED_CACHE=str(cache))
            out = subprocess.run([sys.executable, __file__, "child", str(B), str(iters), str(samp)], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
            r = json.loads(out); passes = iters + 2 if iters > 0 else 1
            print(f"samp={samp} cache={cache} B={B:5d} iters={iters:2d}: {r['us']:9.1f} us  per pass {r['us']/passes:8.2f} us  -> {B*max(iters,1)/r['us']:.3f} M it/s", flush=True)
