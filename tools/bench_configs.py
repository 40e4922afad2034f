"""The rows of BASELINE.md §4 (one line per BASELINE.json config) measured on the GPU box: CPU oracle in its
reference-faithful mode (Jet autodiff, Ceres-LM; 1 thread and `eval_threads` = residual blocks) and in its optimised
mode (analytic 1x6 rows), the HIP path alone on the chip and batched, and the SE(3) difference GPU vs oracle.

    python tools/bench_configs.py            # prints a markdown table (≈ 2 minutes)
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po

ITERS = 10


def med(f, reps=5):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))


def gpu_single(al, solver, iters=ITERS, **kw):
    h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=iters, **kw), 1, al.N, al.H, al.W)
    h.set_alignment(0, al)
    out = {}

    def run():
        out["r"] = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    wall = med(run)
    dev = h.info(0)["device_time_us"] * 1e-6
    h.close()
    return wall, dev, out["r"]


def gpu_batch(als, B, solver, iters=ITERS, **kw):
    h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=iters, **kw), B, max(a.N for a in als), als[0].H, als[0].W)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in range(B):
        a = als[b % len(als)]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        h.set_event_frame(b, fr[b % len(als)])
    p0 = np.stack([als[b % len(als)].p0 for b in range(B)]); q0 = np.stack([als[b % len(als)].q0 for b in range(B)]); v0 = np.stack([als[b % len(als)].v0 for b in range(B)])

    def run():
        h.set_states(0, p0, q0, v0)
        h.optimize_batch(0, 0, B, sync=True)
    wall = med(run)
    it = float(np.mean(h.results(0, B)[:, 14]))
    h.close()
    return wall, it


rows = []
# configs[0] / [1]: 640x480, N = 2000, 10 iterations
al = synth.make_alignment(1234, start="truth_velocity")
o = po.Oracle(al)
t_cpu_opt = med(lambda: o.pose6_lm(al.p0, al.q0, al.v0, iters=ITERS, lambda0=0.01), 3)
ref6 = o.pose6_lm(al.p0, al.q0, al.v0, iters=ITERS, lambda0=0.01)
o12 = po.Oracle(al, num_blocks=1, max_num_iterations=ITERS)
t_cpu_f1 = med(lambda: o12.solve_lm(al.p0, al.q0, al.v0), 3)
ref12 = o12.solve_lm(al.p0, al.q0, al.v0)
o12t = po.Oracle(al, num_blocks=8, max_num_iterations=ITERS, eval_threads=8)
t_cpu_f8 = med(lambda: o12t.solve_lm(al.p0, al.q0, al.v0), 3)
wall6, dev6, (p6, q6, _, i6) = gpu_single(al, capi.SOLVER_LM6)
wall12, dev12, (p12, q12, v12, i12) = gpu_single(al, capi.SOLVER_REF12)
als = [synth.make_alignment(5000 + b) for b in range(8)]
bw6, it6 = gpu_batch(als, 1024, capi.SOLVER_LM6)
bw12, it12 = gpu_batch(als, 1024, capi.SOLVER_REF12)
rows.append(("640x480, N=2000, 10 LM6 iterations (pose only)", f"{ITERS / t_cpu_opt:,.0f} it/s (1 thread, analytic rows)",
             f"{i6['num_iterations'] / wall6:,.0f} it/s ({wall6 * 1e3:.3f} ms per alignment, kernel {dev6 * 1e3:.3f} ms)",
             f"{1024 * it6 / bw6 / 1e6:.2f} M it/s (B=1024)", f"{po.se3_distance(p6, q6, ref6['p'], ref6['q']):.1e}"))
rows.append(("640x480, N=2000, reference problem (12 parameters, Ceres-LM rules, Jet autodiff on the CPU)",
             f"{ref12['num_iterations'] / t_cpu_f1:,.0f} it/s (1 thread) / {ref12['num_iterations'] / t_cpu_f8:,.0f} it/s (8 blocks on 8 threads)",
             f"{i12['num_iterations'] / wall12:,.0f} it/s ({wall12 * 1e3:.3f} ms per alignment, kernel {dev12 * 1e3:.3f} ms)",
             f"{1024 * it12 / bw12 / 1e6:.2f} M it/s (B=1024)", f"{po.se3_distance(p12, q12, ref12['p'], ref12['q']):.1e}"))
# configs[2]: 1280x720, N = 8000, per-point Huber
al3 = synth.make_alignment(2234, H=720, W=1280, N=8000)
o3 = po.Oracle(al3)
tau, _ = po.loss_param(o3.pose6_eval(al3.p0, al3.q0, al3.v0)["r"], po.LP_MAD)
t3 = med(lambda: o3.pose6_lm(al3.p0, al3.q0, al3.v0, iters=ITERS, lambda0=0.01, huber_tau=tau), 3)
ref3 = o3.pose6_lm(al3.p0, al3.q0, al3.v0, iters=ITERS, lambda0=0.01, huber_tau=tau)
w3, d3, (p3, q3, _, i3) = gpu_single(al3, capi.SOLVER_LM6, huber_tau=tau)
bw3, it3 = gpu_batch([al3], 256, capi.SOLVER_LM6, huber_tau=tau)
rows.append(("1280x720, N=8000, per-point Huber, 10 LM6 iterations", f"{ITERS / t3:,.0f} it/s (1 thread)",
             f"{i3['num_iterations'] / w3:,.0f} it/s ({w3 * 1e3:.3f} ms per alignment, kernel {d3 * 1e3:.3f} ms)",
             f"{256 * it3 / bw3 / 1e6:.2f} M it/s (B=256)", f"{po.se3_distance(p3, q3, ref3['p'], ref3['q']):.1e}"))
# configs[3]: 4-level pyramid 2000 -> 16000 points, 6 iterations per level
levels = [(60, 80, 2000), (120, 160, 4000), (240, 320, 8000), (480, 640, 16000)]
lv = [synth.make_alignment(3234 + k, H=H, W=W, N=N, margin=4) for k, (H, W, N) in enumerate(levels)]
hs = []
for a in lv:
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=6), 1, a.N, a.H, a.W)
    h.set_alignment(0, a)
    hs.append(h)
state = {}


def pyr_gpu():
    p, q = lv[0].p0.copy(), lv[0].q0.copy()
    for a, h in zip(lv, hs):
        p, q, _, _ = h.optimize(0, p=p, q=q, v=a.v0)
    state["gpu"] = (p, q)


def pyr_cpu():
    p, q = lv[0].p0.copy(), lv[0].q0.copy()
    for a in lv:
        r = po.Oracle(a).pose6_lm(p, q, a.v0, iters=6, lambda0=0.01)
        p, q = r["p"], r["q"]
    state["cpu"] = (p, q)


tg, tc = med(pyr_gpu), med(pyr_cpu, 2)
for h in hs:
    h.close()
rows.append(("4-level pyramid 80x60 ... 640x480, 2000 -> 16000 points, 6 LM6 iterations per level", f"{24 / tc:,.0f} it/s (1 thread)",
             f"{24 / tg:,.0f} it/s ({tg * 1e3:.3f} ms per 4-level alignment)", "-", f"{po.se3_distance(*state['gpu'], *state['cpu']):.1e}"))
# configs[4]: batch of 64 independent alignments (8 per GPU on 8 GPUs; here all 64 on one)
als64 = [synth.make_alignment(5000 + b) for b in range(64)]
b64, itb = gpu_batch(als64, 64, capi.SOLVER_LM6)
t64 = med(lambda: [po.Oracle(a).pose6_lm(a.p0, a.q0, a.v0, iters=ITERS, lambda0=0.01) for a in als64[:8]], 2) * 8
rows.append(("batch of 64 alignments of 640x480 / 2000 points (one GPU; 8 per GPU when sharded)", f"{64 * ITERS / t64:,.0f} it/s (1 thread, sequential)",
             f"{64 * itb / b64:,.0f} it/s ({b64 * 1e3:.3f} ms per batch)", "-", "as row 1"))
print("| config | CPU oracle | 1 x MI355X, one alignment at a time | 1 x MI355X, batched | SE(3) distance GPU vs oracle |")
print("|---|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")
print(f"\nhost: {os.cpu_count()} hardware threads")
