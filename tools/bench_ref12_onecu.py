"""REF12 on 4 096 x 2 000 points, frames new for the solve (tiles) and frames solved before (strips), over the three batch shapes of
eds_fused12_kernel: paired = two 256-thread alignments per CU, no patch cache in the quad gather (the rule's choice);
wide = one 512-thread alignment per CU, 1 408 of 2 000 patches cached; full = one per CU, EVERY patch cached (round 6:
<S, 512, 2000, false, 1, Q>).  Prints kernel time, LM iterations/s, the kernel name, the worst state difference to the paired
shape's table and whether the step accounting is equal; run it under `rocprofv3 --pmc TCC_EA0_RDREQ_sum` for the request counts
(tools/profile_ref12_shapes.sh).      python3 tools/bench_ref12_onecu.py [B=4096] [reps=6]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
D = 16
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as pool:
    als = list(pool.map(lambda i: synth.make_alignment(5000 + i), range(D)))
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
h = capi.Handle(capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=1), B, 2000, 480, 640)
for b in range(B):
    a = als[b % D]; h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, fr[b % D])
p0 = np.stack([als[b % D].p0 for b in range(B)]); q0 = np.stack([als[b % D].q0 for b in range(B)]); v0 = np.stack([als[b % D].v0 for b in range(B)])
ref = {}
for layout in ("tiles", "strips"):
    h.set_knob("EDS_FUSED_LAYOUT", "tiles" if layout == "tiles" else None)
    if layout == "strips":
        h.prepare_frames(0, B)
    for shape in ("paired", "half", "wide", "full"):
        h.set_knob("EDS_REF12_KERNEL", shape)
        ts, ds = [], []
        for _ in range(REPS):
            h.set_states(0, p0, q0, v0); t = time.perf_counter(); h.optimize_batch(0, 0, B, sync=True); ts.append(time.perf_counter() - t); ds.append(h.info(0)["device_time_us"])
        tab = np.array(h.results(0, B))
        if shape == "paired":
            ref[layout] = tab
        d = float(np.abs(tab[:, :13] - ref[layout][:, :13]).max()); same = bool(np.array_equal(tab[:, 13:], ref[layout][:, 13:]))
        k = np.median(ds[1:])
        print(f"B={B:5d} {layout:6s} {shape:6s}: kernel {k:8.1f} us  {B * float(np.mean(tab[:, 14])) / np.median(ts[1:]) / 1e6:6.2f} M LM it/s  {h.last_launch()['kernel']:48s} "
              f"max |state - paired| {d:.2e}  accounting equal {same}  success {float(np.mean(tab[:, 15])):.3f}", flush=True)
h.set_knob("EDS_REF12_KERNEL", None)
if len(sys.argv) <= 3:          # the oracle's word on the new shape (16 distinct rows; the checker, outside every timing)
    sys.path.insert(0, os.path.join(ROOT, "oracle")); import pyoracle as po
    h.set_knob("EDS_FUSED_LAYOUT", "tiles"); h.set_knob("EDS_REF12_KERNEL", "full")
    h.set_states(0, p0, q0, v0); h.optimize_batch(0, 0, B, sync=True); tab = np.array(h.results(0, B))
    worst, mism = 0.0, 0
    for d in range(D):
        x = als[d]; x32 = synth.Alignment(**{**x.__dict__, "frame": fr[d].astype(np.float64)})
        r = po.Oracle(x32, num_blocks=1, max_num_iterations=10).solve_lm(x.p0, x.q0, x.v0)
        worst = max(worst, po.se3_distance(tab[d, 0:3], tab[d, 3:7], r["p"], r["q"])); mism += int(tab[d, 14] != r["num_iterations"])
    print(f"full shape against the oracle: max SE(3) distance {worst:.3e}, iteration-count mismatches {mism} of {D}")
h.close()
