"""Random batches through the batched rows against their single-slot twins on a second handle: event frames (ragged slice sizes, empty
slices, levels, blur on / off, undistortion map on / off), loss scales (MAD / STD), getCoord with and without culling."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for t in range(trials):
    H, W = int(rng.integers(40, 130)), int(rng.integers(50, 170))
    B = int(rng.integers(1, 80)); level = int(rng.integers(0, 3)); sigma = float(rng.choice([0.0, 0.5, 1.0])); use_map = bool(rng.integers(0, 2))
    N = int(rng.integers(64, 900))
    als = [synth.make_alignment(9900 + 7 * t + k, H=H, W=W, N=N, margin=2) for k in range(3)]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=3)
    hb, hs = capi.Handle(cfg, B, N, H, W), capi.Handle(cfg, 1, N, H, W)
    if use_map:
        cc, rr = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        mx = (cc + 1.3 * np.sin(rr / 17.0)).astype(np.float32); my = (rr + 0.9 * np.cos(cc / 23.0)).astype(np.float32)
        hb.set_undistort_map(mx, my); hs.set_undistort_map(mx, my)
    slices = []
    for b in range(B):
        n = int(rng.choice([0, 1, int(rng.integers(2, 6000))], p=[0.05, 0.05, 0.9]))
        slices.append((rng.integers(0, W, n).astype(np.uint16), rng.integers(0, H, n).astype(np.uint16), rng.integers(0, 2, n).astype(np.uint8)))
    norms = hb.build_event_frame_batch(0, slices, level=level, blur_sigma=sigma)
    ok = True
    why = []
    for b in range(B):
        if len(slices[b][0]) == 0:
            if norms[b] != 0.0: ok = False; why.append(f"empty slice {b}: norm {norms[b]}")
            continue
        n1 = hs.build_event_frame(0, *slices[b], level=level, blur_sigma=sigma)
        f1, fb = hs.get_event_frame(0), hb.get_event_frame(b)
        good = abs(n1 - norms[b]) <= 1e-11 * max(n1, 1e-300) and (np.isnan(f1).all() and np.isnan(fb).all() if n1 == 0.0 else np.abs(f1 - fb).max() <= 1e-6 * max(np.abs(f1).max(), 1e-30))    # (no vote: 0 / 0 everywhere, as the reference)
        if not good: ok = False; why.append(f"frame {b} ({len(slices[b][0])} events): norm {n1!r} vs {norms[b]!r}, max |df| {np.abs(f1 - fb).max():.3e} of {np.abs(f1).max():.3e}")
    # keyframes + real frames for the solves, then the batched scales and getCoord against single-slot calls
    for b in range(B):
        hb.set_alignment(b, als[b % 3])
    hb.optimize_batch(0, 0, B)
    for method in (capi.LP_MAD, capi.LP_STD):
        tb = hb.loss_param_batch(method, 0, B)
        for b in rng.choice(B, size=min(B, 6), replace=False):
            one = hb.loss_param(int(b), method)
            if not abs(tb[b] - one) <= 1e-12 * max(abs(tb[b]), 1e-300): ok = False; why.append(f"loss scale method {method} slot {b}: {tb[b]!r} vs {one!r}")
    poses = []
    for b in range(B):
        a = als[b % 3]
        p = np.array([0.06, -0.03, 0.01]) * rng.uniform(-1, 1, 3); q = synth.quat_from_axis_angle(rng.standard_normal(3), 0.05 * rng.uniform())
        hb.set_state(b, p, q, a.v0); poses.append((p, q))
    delete = bool(rng.integers(0, 2))
    outs = hb.update_points_batch(0, B, delete, want_points=True)
    for b in rng.choice(B, size=min(B, 8), replace=False):
        a = als[int(b) % 3]
        hs.set_alignment(0, a); hs.set_state(0, poses[int(b)][0], poses[int(b)][1], a.v0)
        o1 = hs.update_points(0, delete)
        good = np.array_equal(o1["kept"], outs[int(b)]["kept"]) and np.array_equal(o1["coord"], outs[int(b)]["coord"]) and o1["mean_sq_flow"] == outs[int(b)]["mean_sq_flow"]
        if not good: ok = False; why.append(f"getCoord slot {b}: kept {len(o1['kept'])} vs {len(outs[int(b)]['kept'])}, flow {o1['mean_sq_flow']!r} vs {outs[int(b)]['mean_sq_flow']!r}")
    hb.close(); hs.close()
    print(f"trial {t}: {H}x{W} B={B} level={level} sigma={sigma} map={use_map} N={N} delete={delete}  {'ok' if ok else 'DISAGREE: ' + '; '.join(why[:4])}", flush=True)
    bad += 0 if ok else 1
print(f"{trials} trials, {bad} disagreements")
sys.exit(1 if bad else 0)
