"""Conditioning check for a disagreement fuzz_parity.py reported (CPU only, oracle only): regenerates case `index` of `seed` and solves it
with the fp64 oracle twice more on inputs perturbed at the level of fp32 rounding (frame and start state times 1 + 6e-8 * noise).  If the
oracle's own answers move by as much as the GPU path differed, the case is ill-conditioned (a near-tie in an accept decision, or a Cauchy /
Huber block whose s sits at the corrector's square-root singularity s = a^2), not a discrepancy.

    python tools/replay_parity_case.py seed index [index ...]
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po

seed = int(sys.argv[1]); wanted = sorted(int(x) for x in sys.argv[2:])
rng = np.random.default_rng(seed)
for c in range(wanted[-1] + 1):
    # (the draws of fuzz_parity.py, in its order)
    H, W = int(rng.integers(40, 500)), int(rng.integers(48, 660))
    N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), rng.integers(600, 2100), rng.integers(2100, 5200)], p=[0.1, 0.3, 0.45, 0.15]))
    N = min(N, (H - 4) * (W - 4) // 3)
    al_seed, rot, tn = int(rng.integers(1 << 30)), float(rng.uniform(0.05, 1.0)), float(rng.uniform(0.001, 0.02))
    sampling = int(rng.integers(0, 2))
    solver = str(rng.choice(["lm6", "gn6", "ref12"], p=[0.45, 0.1, 0.45]))
    iters = int(rng.integers(1, 14)) if solver != "gn6" else int(rng.integers(1, 4))
    far = rng.random() < 0.25
    p0 = (0.2 if far else 0.004) * rng.standard_normal(3)
    qa, qb = rng.standard_normal(3), rng.random()
    use_true = rng.random() < 0.7
    if solver == "ref12":
        nb, loss, lp = int(rng.integers(1, 9)), int(rng.integers(0, 3)), float(rng.uniform(0.05, 1.0))
        nc = bool(rng.random() < 0.3)
        scale = float(rng.uniform(5.0, 80.0)) if nc else 1.0
    else:
        tau = float(rng.choice([0.0, 0.0, 0.005, 0.05]))
    if c not in wanted:
        continue
    al = synth.make_alignment(al_seed, H=H, W=W, N=N, margin=2, rot_deg=rot, trans_norm=tn)
    q0 = synth.quat_from_axis_angle(qa, (0.05 if far else 0.003) * qb)
    v0 = al.v_true if use_true else al.v0
    if solver != "ref12":
        print(f"case {c}: {solver} (pose-only) — not handled here"); continue
    if nc:
        al = type(al)(**{**al.__dict__, "frame": al.frame * scale})
    kw = dict(sampling=sampling, num_blocks=nb, nc=nc, loss_type=loss, loss_param=lp, max_num_iterations=iters)
    ref = po.Oracle(al, **kw).solve_lm(p0, q0, v0)
    print(f"case {c}: {H}x{W} N={N} ref12 it={iters} nb={nb} loss={loss} a={lp:.3f} nc={nc}{' far' if far else ''}: oracle it {ref['num_iterations']} "
          f"ok {ref['num_successful_steps']} cost {ref['initial_cost']:.9e} -> {ref['final_cost']:.9e}")
    prng = np.random.default_rng(12345)
    for k in range(4):
        eps = 6e-8
        alp = type(al)(**{**al.__dict__, "frame": al.frame * (1.0 + eps * prng.standard_normal(al.frame.shape))})
        r = po.Oracle(alp, **kw).solve_lm(p0 * (1 + eps * prng.standard_normal(3)), q0, v0)
        d = po.se3_distance(r["p"], r["q"], ref["p"], ref["q"])
        print(f"   oracle on inputs perturbed by 6e-8: it {r['num_iterations']} ok {r['num_successful_steps']} final cost {r['final_cost']:.9e}  pose moved by {d:.2e}")
