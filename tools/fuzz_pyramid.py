"""Random pyramids (size, levels, point counts above and below 2 048, LM6 / REF12): eds_pyr_optimize against the same levels solved one
after the other on plain handles that were given the pyramid's own level frames and level intrinsics."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bad = 0
for t in range(trials):
    L = int(rng.integers(2, 5)); H = int(rng.integers(30, 70)) << (L - 1); W = int(rng.integers(40, 90)) << (L - 1)
    N0 = int(rng.integers(300, min(9000, (H - 40) * (W - 40) // 2))); ref12 = bool(rng.integers(0, 2))
    counts = [max(64, N0 >> l) for l in range(L)]
    al = synth.make_alignment(9500 + t, H=H, W=W, N=N0, rot_deg=0.4, trans_norm=0.008, blur_ksize=11, blur_sigma=3.0, start="ctor" if ref12 else "truth_velocity")
    cfg = capi.default_config(solver=capi.SOLVER_REF12 if ref12 else capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=5,
                              num_blocks=int(rng.integers(1, 4)) if ref12 else 1)
    pyr = capi.Pyramid(cfg, counts, H, W)
    for l, n in enumerate(counts):
        pyr.set_keyframe(l, al.norm_coord[:n], al.grad[:n], al.idp[:n], al.weights[:n], al.fx, al.fy, al.cx, al.cy)
    pyr.set_event_frame(al.frame)
    p, q, v, infos = pyr.optimize(al.p0, al.q0, al.v0)
    cp, cq, cv = al.p0.copy(), al.q0.copy(), al.v0.copy()
    chain = [None] * L
    for l in range(L - 1, -1, -1):
        hl, wl = pyr.level_size(l)
        h = capi.Handle(cfg, 1, counts[l], hl, wl)
        K = capi.Pyramid.level_intrinsics(l, al.fx, al.fy, al.cx, al.cy)
        n = counts[l]
        h.set_keyframe(0, al.norm_coord[:n], al.grad[:n], al.idp[:n], al.weights[:n], *K)
        h.set_event_frame(0, pyr.level_frame(l))
        try:
            cp, cq, cv, chain[l] = h.optimize(0, level=l, p=cp, q=cq, v=cv)
            chain[l]["kernel"] = h.last_launch()["kernel"]
        except capi.EdsError as e:
            chain[l] = {"error": str(e)}            # not usable: the next level starts from the last good pose
        h.close()
    d = po.se3_distance(p, q, cp, cq)
    # (REF12 adds its per-wavefront tiles to the block sums with fp64 LDS atomics, whose order varies from run to run: the SAME solve
    # repeated differs by up to ~1e-7 on a weakly determined level — seed 10, trial 7 does, in round 3's binary too; LM6 is bit-stable)
    tol = 1e-6 if ref12 else 1e-8
    ok = d <= tol and np.abs(v - cv).max() <= tol
    if not ok and ref12:
        # REF12 adds its wavefronts' tiles with fp64 LDS atomics: the last bits of its sums vary from run to run, and on an ILL-POSED
        # pyramid (a coarse level whose few points sit on noise: the cost is flat in the pose) that is enough to send two runs of the SAME
        # call apart — round 6's soak met one whose repeated solves differ by 1e-6 .. 1e+3 (tools/replay_pyramid_case.py).  The yardstick
        # for a difference between the two PATHS is therefore what the pyramid path's own repetitions differ by.
        again = [pyr.optimize(al.p0, al.q0, al.v0)[:2] for _ in range(3)]
        own = max(po.se3_distance(p, q, a_[0], a_[1]) for a_ in again)
        if own > 0.3 * d:
            print(f"trial {t}: {H}x{W} L={L} counts={counts} ref12={ref12}  distance {d:.2e} — ill-posed: the pyramid path's own repetitions differ by {own:.2e} (not counted)", flush=True)
            pyr.close()
            continue
    print(f"trial {t}: {H}x{W} L={L} counts={counts} ref12={ref12}  distance {d:.2e}  {'ok' if ok else 'DISAGREE'}", flush=True)
    if not ok:
        keys = ("num_iterations", "num_successful_steps", "termination", "initial_cost", "final_cost", "usable", "flags")
        for l in range(L - 1, -1, -1):
            print(f"    level {l}: pyramid " + " ".join(f"{k}={infos[l].get(k)}" for k in keys), flush=True)
            print(f"    level {l}: chain   " + " ".join(f"{k}={chain[l].get(k)}" for k in keys) + f" {chain[l].get('kernel', chain[l].get('error'))}", flush=True)
    bad += 0 if ok else 1
    pyr.close()
print(f"{trials} trials, {bad} disagreements")
sys.exit(1 if bad else 0)
