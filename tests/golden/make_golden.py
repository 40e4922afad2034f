"""Generates the golden fixtures of tests/golden/*.npz.

The reference (uzh-rpg/slam-eds) has no tests or golden vectors and cannot be built here
(SURVEY.md §4, §8c), so these vectors come from the repo's own C++ oracle
(oracle/eds_oracle.hpp) and are cross-checked, inside this script, against the independent
numpy oracle (oracle/np_oracle.py).  Parity with the real reference stays UNPINNED.

    python tests/golden/make_golden.py
"""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
synth = importlib.import_module("slam-eds_amd.synth")
import np_oracle as npo   # noqa: E402
import pyoracle as po     # noqa: E402


def digest(al):
    h = hashlib.sha256()
    for a in (al.norm_coord, al.grad, al.idp, al.weights, al.frame):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def eval_pose(seed):
    rng = np.random.default_rng(seed + 99)
    q = synth.quat_from_axis_angle(rng.standard_normal(3), 0.003)
    p = 0.002 * rng.standard_normal(3)
    return p, q


def case(seed, H, W, N, nb, store_inputs):
    al = synth.make_alignment(seed, H=H, W=W, N=N)
    p, q = eval_pose(seed)
    v = al.v_true + 0.05 * np.random.default_rng(seed + 7).standard_normal(6)
    v /= np.linalg.norm(v)
    # Solves start from a GENERIC near-identity pose, not from identity itself: the synthetic keyframe
    # points sit on integer pixels, so at identity every projection lands exactly on a cell border,
    # where the bilinear sampler's gradient is discontinuous and the chosen cell hinges on the last
    # bit of the fp64 projection (the bicubic sampler is C1 and does not care).
    srng = np.random.default_rng(seed + 1234)
    sp = 2e-4 * srng.standard_normal(3)
    sq = synth.quat_from_axis_angle(srng.standard_normal(3), 4e-4)
    out = dict(seed=seed, H=H, W=W, N=N, num_blocks=nb, eval_p=p, eval_q=q, eval_v=v, sha256=digest(al),
               start_p=sp, start_q=sq)
    if store_inputs:
        out.update(norm_coord=al.norm_coord, grad=al.grad, idp=al.idp, weights=al.weights, frame=al.frame,
                   K=np.array([al.fx, al.fy, al.cx, al.cy]), p0=al.p0, q0=al.q0, v0=al.v0)
    for sampling, tag in ((po.BICUBIC, "bc"), (po.BILINEAR, "bl")):
        o = po.Oracle(al, sampling=sampling, num_blocks=nb, max_num_iterations=10)
        e12 = o.eval12(p, q, v)
        e6 = o.pose6_eval(p, q, v)
        # independent cross-check (closed forms, numpy)
        r_np, J_np, J6_np = npo.jacobians(al, p, q, v, nb, "bicubic" if sampling == po.BICUBIC else "bilinear")
        assert np.abs(e12["r_raw"] - r_np).max() < 1e-12
        assert np.abs(e12["J_local_raw"] - J_np).max() < 1e-9
        assert np.abs(e6["J"] - J6_np).max() < 1e-9
        out[f"{tag}_r"] = e12["r_raw"]
        # full-size cases keep every 8th Jacobian row (plus the full normal matrices) to stay small
        sub = slice(None) if store_inputs else slice(None, None, 8)
        out[f"{tag}_J12"] = e12["J_local_raw"][sub]
        out[f"{tag}_J6"] = e6["J"][sub]
        out[f"{tag}_J12tJ12"] = e12["J_local_raw"].T @ e12["J_local_raw"]
        out[f"{tag}_J12tr"] = e12["J_local_raw"].T @ e12["r_raw"]
        out[f"{tag}_H6"] = e6["H"]
        out[f"{tag}_b6"] = e6["b"]
        out[f"{tag}_cost"] = e12["cost"]
        lm = o.pose6_lm(sp, sq, al.v0, iters=10, lambda0=0.01)
        out[f"{tag}_lm6_inc"] = lm["increments"]; out[f"{tag}_lm6_cost"] = lm["costs"]
        out[f"{tag}_lm6_acc"] = lm["accepted"]; out[f"{tag}_lm6_p"] = lm["p"]; out[f"{tag}_lm6_q"] = lm["q"]
        gn = o.pose6_gn(sp, sq, al.v0, iters=2)
        out[f"{tag}_gn6_inc"] = gn["increments"]; out[f"{tag}_gn6_p"] = gn["p"]; out[f"{tag}_gn6_q"] = gn["q"]
        for loss, lname in ((po.LOSS_NONE, "none"), (po.LOSS_HUBER, "huber"), (po.LOSS_CAUCHY, "cauchy")):
            oo = po.Oracle(al, sampling=sampling, num_blocks=nb, loss_type=loss, loss_param=0.3, max_num_iterations=10)
            s = oo.solve_lm(sp, sq, al.v0)
            out[f"{tag}_ref12_{lname}"] = np.concatenate([s["p"], s["q"], s["v"], [s["final_cost"], s["num_iterations"],
                                                         s["num_successful_steps"], s["termination"]]])
            if loss == po.LOSS_NONE:
                r_fin = oo.eval12(s["p"], s["q"], s["v"], jac=False)["r_raw"]
                out[f"{tag}_mad_tau"] = po.loss_param(r_fin, po.LP_MAD)[0]
                out[f"{tag}_std_tau"] = po.loss_param(r_fin, po.LP_STD)[0]
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "small_n64.npz"), **case(11, 48, 64, 64, 1, True))
    np.savez_compressed(os.path.join(HERE, "small_n200_nb3.npz"), **case(12, 60, 80, 200, 3, True))
    for seed in (1234, 1235):
        np.savez_compressed(os.path.join(HERE, f"full_{seed}.npz"), **case(seed, 480, 640, 2000, 1, False))
    np.savez_compressed(os.path.join(HERE, "full_1236_nb4.npz"), **case(1236, 480, 640, 2000, 4, False))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
