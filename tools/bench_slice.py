"""Latency of one tracking step of a live event camera through the C ABI: events -> event frame on the device ->
REF12 (or LM6) alignment warm-started from the previous pose -> loss scale -> point maintenance (getCoord + culling).
One keyframe, one alignment at a time: the latency-bound regime of the real system."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
al = synth.make_alignment(1234)
rng = np.random.default_rng(0)
for solver, name in ((capi.SOLVER_REF12, "REF12 (4 blocks, Huber)"), (capi.SOLVER_LM6, "LM6")):
    for n_events in (20_000, 100_000):
        cfg = capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=4 if solver == capi.SOLVER_REF12 else 1,
                                  loss_type=capi.LOSS_HUBER if solver == capi.SOLVER_REF12 else capi.LOSS_NONE, loss_param=0.3)
        h = capi.Handle(cfg, 1, al.N, al.H, al.W)
        h.set_keyframe(0, al.norm_coord, al.grad, al.idp, al.weights, al.fx, al.fy, al.cx, al.cy)
        # events that reproduce the sign structure of the synthetic frame
        strong = np.argwhere(np.abs(al.frame) > 0.25 * np.abs(al.frame).max())
        pick = strong[rng.integers(0, len(strong), n_events)]
        ex, ey = pick[:, 1].astype(np.uint16), pick[:, 0].astype(np.uint16)
        pol = (al.frame[pick[:, 0], pick[:, 1]] > 0).astype(np.uint8)
        t_parts = np.zeros(4); reps = 30
        p, q, v = al.p0.copy(), al.q0.copy(), al.v0.copy()
        for r in range(reps + 3):
            t0 = time.perf_counter(); h.build_event_frame(0, ex, ey, pol)
            t1 = time.perf_counter(); p1, q1, v1, info = h.optimize(0, p=p, q=q, v=v)
            t2 = time.perf_counter(); tau = h.loss_param(0, capi.LP_MAD)
            t3 = time.perf_counter(); out = h.update_points(0, False)
            t4 = time.perf_counter()
            if r >= 3: t_parts += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3])
        t_parts *= 1e6 / reps
        print(f"{name:24s} {n_events:7d} events: frame {t_parts[0]:6.1f} us + solve {t_parts[1]:6.1f} us ({info['num_iterations']} it) + loss scale {t_parts[2]:5.1f} us + points {t_parts[3]:6.1f} us"
              f" = {t_parts.sum():7.1f} us per slice ({1e6 / t_parts.sum():6.0f} slices/s)", flush=True)
        h.close()
