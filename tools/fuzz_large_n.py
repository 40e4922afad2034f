"""Random (N, batch, sampler, Huber) cases — by default N > 2 048; `fuzz_large_n.py seed trials nmin nmax` for another range, e.g. 513 2048
for the small teams on handles that are exactly as large as their alignments: the team path optimize picks against the one-CU streaming kernel (EDS_LM6_TEAM=1)
on the same handle — poses within 1e-6, accept patterns identical — and REF12 teams of 8 / 16 against EDS_REF12_TEAM=1."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nmin = int(sys.argv[3]) if len(sys.argv) > 3 else 2049
nmax = int(sys.argv[4]) if len(sys.argv) > 4 else 12000
bad = 0
for t in range(trials):
    N = int(rng.integers(nmin, nmax)); B = int(rng.integers(1, 24)) if rng.random() < 0.6 else int(rng.integers(24, 200)); sampling = int(rng.integers(0, 2));   # (large batches: members of 2 048 points)
    tau = float(rng.choice([0.0, 0.01]))
    ref12 = bool(rng.integers(0, 3) == 0)
    H, W = 240, 320
    als = [synth.make_alignment(8800 + 10 * t + k, H=H, W=W, N=N, start="ctor" if ref12 else "truth_velocity") for k in range(2)]
    cfg = capi.default_config(solver=capi.SOLVER_REF12 if ref12 else capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, sampling=sampling, max_num_iterations=6,
                              huber_tau=0.0 if ref12 else tau, num_blocks=int(rng.integers(1, 5)) if ref12 else 1)
    h = capi.Handle(cfg, B, N, H, W)
    for b in range(B):
        a = als[b % 2]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        if b < 2: h.set_event_frame(b, a.frame)
        else: h.share_event_frame(b, b % 2)
    P0 = np.stack([als[b % 2].p0 for b in range(B)]); Q0 = np.stack([als[b % 2].q0 for b in range(B)]); V0 = np.stack([als[b % 2].v0 for b in range(B)])
    out = {}
    for mode in ("auto", "one"):
        for k in ("EDS_LM6_TEAM", "EDS_REF12_TEAM"):
            if mode == "one": os.environ[k] = "1"
            else: os.environ.pop(k, None)
        h.set_states(0, P0, Q0, V0)
        h.optimize_batch(0, 0, B)
        out[mode] = (h.results(0, B).copy(), [h.info(b)["num_iterations"] for b in range(B)])
    for k in ("EDS_LM6_TEAM", "EDS_REF12_TEAM"): os.environ.pop(k, None)
    h.close()
    (ta, ia), (to, io) = out["auto"], out["one"]
    worst = max(po.se3_distance(ta[b, 0:3], ta[b, 3:7], to[b, 0:3], to[b, 3:7]) for b in range(B))
    ok = worst <= (1e-5 if ref12 else 1e-6) and ia == io and np.array_equal(ta[:, 15], to[:, 15])
    print(f"trial {t}: N={N} B={B} sampling={sampling} tau={tau} ref12={ref12}  worst {worst:.2e}  {'ok' if ok else 'DISAGREE'}", flush=True)
    bad += 0 if ok else 1
print(f"{trials} trials, {bad} disagreements")
sys.exit(1 if bad else 0)
