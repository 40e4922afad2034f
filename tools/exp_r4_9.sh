cd /root/repo
cat > /tmp/r12.py <<'PY'
import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = int(sys.argv[1])
als = [synth.make_alignment(5000 + i) for i in range(16)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=1), B, 2000, 480, 640)
for b in range(B):
    a = als[b % 16]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 16])
h.prepare_frames(0, B)
p0 = np.stack([als[b % 16].p0 for b in range(B)]); q0 = np.stack([als[b % 16].q0 for b in range(B)]); v0 = np.stack([als[b % 16].v0 for b in range(B)])
for _ in range(3):
    h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B, sync=True)
print(B, h.last_launch()["kernel"], h.info(0)["device_time_us"], flush=True)
PY
for B in 4096 256; do echo "== B=$B"; EDS_HIP_LIB=$PWD/slam-eds_amd/csrc/libeds_hip_stamps.so python /tmp/r12.py $B 2>&1 | grep -E "stamps12|eds_fused12" | tail -6; done
