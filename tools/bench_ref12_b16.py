import importlib, os, sys, time
sys.path.insert(0, ".")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
als = [synth.make_alignment(5000 + i) for i in range(8)]
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
for B in (9, 12, 16):
  for nb in (1, 4):
    for team, g in (("", ""), ("8", "2"), ("4", "4"), ("4", "2"), ("8", "1")):
        h = capi.Handle(capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb), B, 2000, 480, 640)
        if team: h.set_knob("EDS_REF12_TEAM", team); h.set_knob("EDS_REF12_GROUPS", g)
        for b in range(B):
            a = als[b % 8]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % 8])
        p0 = np.stack([als[b % 8].p0 for b in range(B)]); q0 = np.stack([als[b % 8].q0 for b in range(B)]); v0 = np.stack([als[b % 8].v0 for b in range(B)])
        ds = []
        for _ in range(40):
            h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B, sync=True); ds.append(h.info(0)["device_time_us"])
        print(f"B={B} nb={nb} TEAM={team or 'rule'} GROUPS={g or 'rule'}: kernel {np.median(ds[5:]):.1f} us {h.last_launch()['kernel']} flags {h.info(0)['flags']}", flush=True)
        h.close()
