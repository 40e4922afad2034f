import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(5000)
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), 1, 2000, 480, 640)
h.set_alignment(0, al)
f64 = np.ascontiguousarray(al.frame, dtype=np.float64); f32 = np.ascontiguousarray(al.frame, dtype=np.float32)
for name, f in (("f64", f64), ("f32", f32)):
    ts = []
    for rep in range(40):
        h.sync() if hasattr(h, "sync") else None
        t0 = time.perf_counter(); h.set_event_frame(0, f); t1 = time.perf_counter(); ts.append((t1 - t0) * 1e6)
        time.sleep(0.0005)
    print(name, "set_event_frame median %.1f us  min %.1f" % (np.median(ts[5:]), np.min(ts[5:])))
# pure host narrowing in numpy for scale
ts = []
out = np.empty_like(f32)
for rep in range(40):
    t0 = time.perf_counter(); np.copyto(out, f64, casting="same_kind"); ts.append((time.perf_counter() - t0) * 1e6)
print("numpy fp64->fp32 copy median %.1f us" % np.median(ts[5:]))
