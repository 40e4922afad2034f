"""Randomised cross-check of the rows around the solve (event frame, keyframe set-up, point maintenance, loss scale)
against their numpy oracles on the GPU box — a one-off hunt for rare divergences, not a test.

    python tools/fuzz_rows.py [cases] [seed]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")
import np_frame_oracle as fo
import np_keyframe_oracle as ko
import np_points_oracle as pto
import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for c in range(cases):
    H, W = int(rng.integers(24, 300)), int(rng.integers(24, 400))
    tag = f"case {c:3d} {H}x{W}"
    h = capi.Handle(capi.default_config(), 2, H * W, H, W)
    # ---- event frame
    n = int(rng.choice([0, 1, rng.integers(2, 200), rng.integers(200, 60000)]))
    x = rng.integers(0, W, n).astype(np.uint16); y = rng.integers(0, H, n).astype(np.uint16); pol = rng.integers(0, 2, n).astype(np.uint8)
    lut = rng.random() < 0.5
    if lut:
        cc, rr = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        mapx = (cc + rng.uniform(0, 3) * np.sin(rr / 23.0) + 0.004 * (cc - W / 2)).astype(np.float32)
        mapy = (rr + rng.uniform(0, 3) * np.cos(cc / 31.0) - 0.003 * (rr - H / 2)).astype(np.float32)
    else:
        mapx = mapy = None
    level = int(rng.integers(0, 4)); sigma = float(rng.choice([0.0, 0.5, 1.1])); expw = bool(rng.integers(0, 2))
    h.set_undistort_map(mapx, mapy)
    if n > 0:
        ref, rn = fo.event_frame(x, y, pol, H, W, mapx, mapy, level=level, sigma=sigma, use_exp_weights=expw)
        if rn > 0:
            gn = h.build_event_frame(1, x, y, pol, level=level, blur_sigma=sigma, use_exp_weights=expw)
            got = h.get_event_frame(1)
            if abs(gn - rn) > 1e-10 * rn or np.abs(got - ref).max() > 2e-7 * np.abs(ref).max():
                print(tag, f"event frame n={n} level={level} sigma={sigma}: norm {gn} vs {rn}, max diff {np.abs(got - ref).max():.2e}"); bad += 1
    # ---- keyframe set-up
    img = rng.standard_normal((H, W))
    for _ in range(2):
        img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 0) + np.roll(img, -1, 1)) / 5.0
    img = (img - img.min()) / (img.max() - img.min())
    if rng.random() < 0.5:
        img = np.round(img * 255).astype(np.uint8)
    elif rng.random() < 0.5:
        img = img.astype(np.float32)
    cell = int(rng.choice([4, 7, 16, 20, 32])); cell = min(cell, H, W)
    method = int(rng.integers(0, 2)); ncell = (H // cell) * (W // cell)
    npts = int(ncell * rng.integers(1, min(cell * cell, 12) + 1))
    K = (0.8 * W, 0.8 * W, (W - 1) / 2, (H - 1) / 2)
    md = int(rng.choice([0, 1, rng.integers(2, 3000)]))
    dxy = np.stack([rng.uniform(0, W - 1, md), rng.uniform(0, H - 1, md)], axis=1) if md else None
    didp = rng.uniform(0.2, 1.0, md) if md else None
    thr = float(rng.choice([0.0, 0.5, 0.7, 0.9]))
    refk = ko.keyframe(img, K, method, npts, cell=cell, depth_xy=dxy, depth_idp=didp, weight_threshold=thr)
    try:
        out = h.build_keyframe(0, img, K, method=method, num_points=npts, cell=cell, depth_xy=dxy, depth_idp=didp, weight_threshold=thr)
        ok = (out["coord"].shape == refk["coord"].shape and np.array_equal(out["coord"], refk["coord"]) and np.array_equal(out["idp"], refk["idp"])
              and np.abs(out["grad"] - refk["grad"]).max() <= 1e-11 * max(1.0, np.abs(refk["grad"]).max()) and np.abs(out["weights"] - refk["weights"]).max() <= 1e-13)
        if not ok:
            print(tag, f"keyframe cell={cell} method={method} npts={npts} depth={md} thr={thr}: GPU {out['coord'].shape} vs oracle {refk['coord'].shape}"); bad += 1
    except capi.EdsError as e:
        if len(refk["coord"]) != 0:
            print(tag, f"keyframe cell={cell} method={method} npts={npts} depth={md} thr={thr}: GPU error '{e}' but the oracle has {len(refk['coord'])} points"); bad += 1
    h.close()
    # ---- point maintenance + loss scale on a synthetic alignment of this size
    N = int(min(rng.integers(1, 9000), (H - 4) * (W - 4) // 3))
    al = synth.make_alignment(int(rng.integers(1 << 30)), H=H, W=W, N=N, margin=2)
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=2), 1, N, H, W)
    h.set_alignment(0, al)
    h.optimize_batch(0, 0, 1)
    for method_lp, lp in ((capi.LP_MAD, po.LP_MAD), (capi.LP_STD, po.LP_STD)):
        tau = h.loss_param_batch(method_lp)[0]
        r = h.residuals(0)
        tr = po.loss_param(r, lp)[0]
        if not (abs(tau - tr) <= 1e-12 * max(abs(tr), 1e-300) or (np.isnan(tau) and np.isnan(tr))):
            print(tag, f"loss scale N={N} method={method_lp}: {tau} vs {tr}"); bad += 1
        h.optimize_batch(0, 0, 1)
    p = 0.1 * rng.standard_normal(3) * rng.random()
    q = synth.quat_from_axis_angle(rng.standard_normal(3), 0.08 * rng.random())
    delete = bool(rng.integers(0, 2))
    refp = pto.get_coord(al.norm_coord, al.idp, al.coord, (al.fx, al.fy, al.cx, al.cy), H, W, p, q, delete)
    h.set_state(0, p, q, al.v0)
    out = h.update_points(0, delete)
    if not np.array_equal(out["kept"], refp["kept"]):
        # a point within fp32 round-off of the frame border may fall on the other side
        sym = np.setxor1d(out["kept"], refp["kept"])
        _, _, u, v = __import__("np_oracle").project(al, p, q)
        print(tag, f"update_points N={N}: kept differs at {len(sym)} points"); bad += 1
    elif len(refp["kept"]) and (np.abs(out["coord"] - refp["coord"]).max() > 2e-4 or abs(out["mean_sq_flow"] - refp["mean_sq_flow"]) > 1e-4 * max(refp["mean_sq_flow"], 1e-12)):
        print(tag, f"update_points N={N}: coord diff {np.abs(out['coord'] - refp['coord']).max():.2e}"); bad += 1
    h.close()
print(f"{cases} cases, {bad} disagreements")
sys.exit(1 if bad else 0)
