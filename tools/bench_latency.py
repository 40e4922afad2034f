"""Single-alignment latency of the persistent kernels for different workgroup shapes (EDS_FUSED_THREADS / EDS_FUSED_PPT)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(1234)
for B in (1, 8, 64):
    for thr, ppt in ((512, 4), (1024, 2), (256, 8)):
        os.environ["EDS_FUSED_THREADS"] = str(thr); os.environ["EDS_FUSED_PPT"] = str(ppt)
        h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, al.N, al.H, al.W)
        for b in range(B): h.set_alignment(b, al)
        p0 = np.stack([al.p0] * B); q0 = np.stack([al.q0] * B); v0 = np.stack([al.v0] * B)
        ts, ds = [], []
        for _ in range(12):
            h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); ts.append(time.perf_counter() - t); ds.append(h.info(0)["device_time_us"])
        print(f"B={B:3d} threads {thr:4d} x {ppt}: wall {np.median(ts[2:])*1e6:7.1f} us  kernel {np.median(ds[2:]):7.1f} us", flush=True)
        h.close()
