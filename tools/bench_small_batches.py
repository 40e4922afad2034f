import importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
als = [synth.make_alignment(5000 + b) for b in range(8)]
for solver, name in ((capi.SOLVER_LM6, "LM6"), (capi.SOLVER_REF12, "REF12")):
    for B in (1, 64):
        h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=1), B, 2000, 480, 640)
        for b in range(B): h.set_alignment(b, als[b % 8])
        P0 = np.stack([als[b % 8].p0 for b in range(B)]); Q0 = np.stack([als[b % 8].q0 for b in range(B)]); V0 = np.stack([als[b % 8].v0 for b in range(B)])
        h.set_knob("EDS_FUSED_LAYOUT", "tiles")
        t = h.bench_batch(P0, Q0, V0, reps=200)
        print(f"{name} B={B}: step {t['step_us']:.1f} us  kernel {t['kernel_us']:.1f}  slowest {t['slowest_step_us']:.1f}", flush=True)
        h.close()
