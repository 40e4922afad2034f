import importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(5000)
for nb, loss in ((1, capi.LOSS_NONE), (4, capi.LOSS_HUBER), (8, capi.LOSS_HUBER)):
    cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb, loss_type=loss, loss_param=0.3)
    h = capi.Handle(cfg, 1, 2000, 480, 640); h.set_alignment(0, al); h.set_knob("EDS_FUSED_LAYOUT", "tiles")
    ks = []
    for _ in range(200):
        p, q, v, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0); ks.append(info["device_time_us"])
    print(f"nb={nb}: kernel median {np.median(ks[20:]):.1f} us  iterations {info['num_iterations']}  state digest {float(np.abs(np.concatenate([p, q, v])).sum()):.12f}  {h.last_launch()['kernel']}", flush=True)
    h.close()
