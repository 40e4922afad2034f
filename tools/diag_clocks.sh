#!/bin/bash
# Shader clock and power while the headline batch runs back to back (and idle before / after): does the chip hold its clock under this kernel?
cd /root/repo
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power|mclk" | head -4
python - <<'PY' &
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = 4096
als = [synth.make_alignment(5000 + i) for i in range(16)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
for b in range(B):
    a = als[b % 16]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 16])
h.prepare_frames(0, B)
p0 = np.stack([als[b % 16].p0 for b in range(B)]); q0 = np.stack([als[b % 16].q0 for b in range(B)]); v0 = np.stack([als[b % 16].v0 for b in range(B)])
print("running", flush=True)
t0 = time.time(); ks = []
while time.time() - t0 < 12:
    h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B, sync=True); ks.append(h.info(0)["device_time_us"])
print(f"kernel first 20 steps {np.median(ks[:20]):.1f} us, last 20 steps {np.median(ks[-20:]):.1f} us, {len(ks)} steps", flush=True)
PY
sleep 14
for k in 1 2 3 4 5 6; do echo "== under load, sample $k"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power|mclk" | head -4; sleep 1.2; done
wait
echo "== idle again"; sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power|mclk" | head -4
