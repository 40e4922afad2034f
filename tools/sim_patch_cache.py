"""What would an LDS patch cache hit?  CPU simulation on the oracle's LM6 trajectories of the bench's alignments.

Measurement tool (Next #1: "measure first"): for every point and every pass of a 10-iteration LM6 solve the oracle's
accepted / rejected candidate poses give the integer cell of the bicubic patch; the script replays that sequence against several
cache policies and prints gathers per point-pass, plus what each gather costs in 128-byte lines on the 4x4 tiles (two tiles per line).

    python tools/sim_patch_cache.py [--n 16] [--iters 10]

Uses oracle/ (test infrastructure) — never imported by the product.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import importlib

synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po


def cells_for_pose(al, p, q):
    R = po.quat_to_R(q)
    x, y = al.norm_coord[:, 0], al.norm_coord[:, 1]
    z = 1.0 / (al.idp + 1e-5)
    P = (R @ np.stack([x * z, y * z, z])).T + p
    u = al.fx * P[:, 0] / P[:, 2] + al.cx
    v = al.fy * P[:, 1] / P[:, 2] + al.cy
    return np.floor(v).astype(np.int64), np.floor(u).astype(np.int64), u, v


def lines_of_patch(r0, c0):
    """128-byte lines (8 columns x 4 rows: two 4x4 tiles side by side) a 4x4 patch with first tap (r0-1, c0-1) touches."""
    ra, ca = r0 - 1, c0 - 1
    nr = (ra + 3) // 4 - ra // 4 + 1
    nc = (ca + 3) // 8 - ca // 8 + 1
    return nr * nc


def lines_of_window(r0, c0, k):
    """lines of the (4 + 2k)^2 window around the patch"""
    ra, ca = r0 - 1 - k, c0 - 1 - k
    n = 4 + 2 * k
    nr = (ra + n - 1) // 4 - ra // 4 + 1
    nc = (ca + n - 1) // 8 - ca // 8 + 1
    return nr * nc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lambda0", type=float, default=0.01)
    a = ap.parse_args()
    tot = {}
    per_pass = None

    def add(k, v):
        tot[k] = tot.get(k, 0.0) + v

    for s in range(a.n):
        al = synth.make_alignment(5000 + s)
        o = po.Oracle(al)
        ref = o.pose6_lm(al.p0, al.q0, al.v0, iters=a.iters, lambda0=a.lambda0)
        # pose sequence: pass 0 = start, pass k = candidate k
        poses = [(al.p0.copy(), al.q0.copy())]
        acc_flags = [1]
        pa, qa = al.p0.copy(), al.q0.copy()
        for k in range(ref["iterations"]):
            pc, qc = po.se3_left_update(ref["increments"][k], pa, qa)
            poses.append((pc, qc))
            acc_flags.append(int(ref["accepted"][k]))
            if ref["accepted"][k]:
                pa, qa = pc, qc
        N = al.N
        npass = len(poses)
        rc = [cells_for_pose(al, p, q) for p, q in poses]
        if per_pass is None:
            per_pass = np.zeros((5, 64))
        add("point_passes", N * npass)
        add("passes", npass)
        add("rejected", npass - sum(acc_flags))
        # P1: single slot (what the kernels do today)
        tagr, tagc = np.full(N, -99), np.full(N, -99)
        g1 = l1 = 0
        # P2: accepted backup: slot L = last evaluated, slot B = accepted pose's patch (saved when L is overwritten while it holds it)
        Lr, Lc = np.full(N, -99), np.full(N, -99)
        Br, Bc = np.full(N, -99), np.full(N, -99)
        L_is_acc = np.zeros(N, bool)
        g2 = l2 = 0
        # P3: +-1 window (6x6), P3b: +-2 window (8x8)
        w1r, w1c = np.full(N, -99), np.full(N, -99)
        w2r, w2c = np.full(N, -99), np.full(N, -99)
        g3 = l3 = g4 = l4 = 0
        # P5: two-entry LRU
        e0 = np.full((N, 2), -99); e1 = np.full((N, 2), -99)
        g5 = l5 = 0
        # P6: single slot, INCREMENTAL refetch: a patch that overlaps the cached one fetches only its new taps (rows / columns kept in place
        # by toroidal addressing) — same gathers as P1, fewer lines per gather
        ir, ic = np.full(N, -99), np.full(N, -99)
        l6 = 0
        disp_acc, disp_rej = [], []
        prev_u = prev_v = None
        for k in range(npass):
            r0, c0, u, v = rc[k]
            lp = np.array([lines_of_patch(int(r), int(c)) for r, c in zip(r0, c0)])
            # P1
            m = (tagr != r0) | (tagc != c0)
            g1 += m.sum(); l1 += lp[m].sum()
            per_pass[0, k] += m.sum()
            tagr, tagc = r0.copy(), c0.copy()
            # P2
            hitL = (Lr == r0) & (Lc == c0)
            hitB = ~hitL & (Br == r0) & (Bc == c0)
            miss = ~hitL & ~hitB
            g2 += miss.sum(); l2 += lp[miss].sum()
            per_pass[1, k] += miss.sum()
            # miss: save L to B if L holds the accepted patch, then L = new
            sv = miss & L_is_acc
            Br[sv], Bc[sv] = Lr[sv], Lc[sv]
            # restore: L <- B (B keeps it too)
            Lr[hitB], Lc[hitB] = Br[hitB], Bc[hitB]
            Lr[miss], Lc[miss] = r0[miss], c0[miss]
            L_is_acc[miss | hitB] = False
            L_is_acc[hitB] = False
            if acc_flags[k]:
                L_is_acc[:] = True       # whatever L holds now is the accepted pose's patch
            else:
                # rejected: L holds a rejected candidate's patch unless it hit in place on the accepted one
                pass
            # P3
            m3 = (np.abs(w1r - r0) > 1) | (np.abs(w1c - c0) > 1)
            g3 += m3.sum(); l3 += np.array([lines_of_window(int(r), int(c), 1) for r, c in zip(r0[m3], c0[m3])]).sum()
            per_pass[2, k] += m3.sum()
            w1r[m3], w1c[m3] = r0[m3], c0[m3]
            m4 = (np.abs(w2r - r0) > 2) | (np.abs(w2c - c0) > 2)
            g4 += m4.sum(); l4 += np.array([lines_of_window(int(r), int(c), 2) for r, c in zip(r0[m4], c0[m4])]).sum()
            w2r[m4], w2c[m4] = r0[m4], c0[m4]
            # round 5 (VERDICT r4 Next #2): what the LATENCY regime asks of a window — not gathers per point but whether a WAVEFRONT's pass
            # (64 consecutive points of a team member) issues no gather at all; one missing point keeps the dependent gather on the chain
            if k > 0:
                for nm_, mm_ in (("slot", m), ("w1", m3), ("w2", m4)):
                    w_ = mm_[: (N // 64) * 64].reshape(-1, 64).any(1)
                    add(f"waves_with_a_gather {nm_}", w_.sum()); add(f"waves {nm_}", w_.size)
                    if k >= npass - 3:
                        add(f"late waves_with_a_gather {nm_}", w_.sum()); add(f"late waves {nm_}", w_.size)
                    t_ = mm_[: (N // 512) * 512].reshape(-1, 512).any(1)          # ... and a team member of 512 points (its pass ends at its slowest wavefront)
                    add(f"members_with_a_gather {nm_}", t_.sum()); add(f"members {nm_}", t_.size)
                    if k >= npass - 3:
                        add(f"late members_with_a_gather {nm_}", t_.sum()); add(f"late members {nm_}", t_.size)
            # P6
            for n in np.nonzero((ir != r0) | (ic != c0))[0]:
                new_t = {(int(r0[n]) - 1 + i, int(c0[n]) - 1 + j) for i in range(4) for j in range(4)}
                if ir[n] > -90:
                    new_t -= {(int(ir[n]) - 1 + i, int(ic[n]) - 1 + j) for i in range(4) for j in range(4)}
                l6 += len({(r // 4, c // 8) for r, c in new_t})
            ir, ic = r0.copy(), c0.copy()
            # P5 LRU-2
            key = np.stack([r0, c0], 1)
            h0 = (e0 == key).all(1); h1 = (e1 == key).all(1) & ~h0
            m5 = ~h0 & ~h1
            g5 += m5.sum(); l5 += lp[m5].sum()
            # on hit in e1: swap so e0 = MRU; on miss: e1 = e0, e0 = key
            sw = h1
            tmp = e0[sw].copy(); e0[sw] = e1[sw]; e1[sw] = tmp
            e1[m5] = e0[m5]; e0[m5] = key[m5]
            if k > 0:
                d = np.hypot(u - prev_u, v - prev_v)
                (disp_acc if acc_flags[k] else disp_rej).append(d)
            prev_u, prev_v = u, v
        for nm, g, l in (("P1 single slot", g1, l1), ("P2 accepted backup", g2, l2), ("P3 6x6 window", g3, l3), ("P4 8x8 window", g4, l4), ("P5 LRU-2", g5, l5),
                         ("P6 incremental", g1, l6)):
            add(nm + " gathers", g); add(nm + " lines", l)
        add("disp_acc_mean", np.mean(np.concatenate(disp_acc)) if disp_acc else 0)
        add("disp_rej_mean", np.mean(np.concatenate(disp_rej)) if disp_rej else 0)
        print(f"seed {5000 + s}: passes {npass} accepted {sum(acc_flags) - 1} pattern {''.join(str(f) for f in acc_flags[1:])}", flush=True)
    pp = tot["point_passes"]
    print(f"\n{a.n} alignments, {tot['passes'] / a.n:.1f} passes each, rejected {tot['rejected'] / tot['passes']:.2f}")
    print(f"mean displacement between consecutive passes: accepted candidates {tot['disp_acc_mean'] / a.n:.2f} px, rejected {tot['disp_rej_mean'] / a.n:.2f} px")
    for nm in ("P1 single slot", "P2 accepted backup", "P3 6x6 window", "P4 8x8 window", "P5 LRU-2", "P6 incremental"):
        g, l = tot[nm + " gathers"], tot[nm + " lines"]
        print(f"{nm:22s}: gathers / point-pass {g / pp:.3f}   lines / gather {l / max(g, 1):.2f}   lines / point-pass {l / pp:.3f}")
    print("misses per pass (fraction of points), P1 | P2 | P3:")
    for k in range(int(tot["passes"] / a.n + 0.5)):
        print(f"  pass {k:2d}: {per_pass[0, k] / (a.n * 2000):.2f} | {per_pass[1, k] / (a.n * 2000):.2f} | {per_pass[2, k] / (a.n * 2000):.2f}")

    print("\nwavefront passes (64 consecutive points, passes 2..) that still issue a gather — the latency regime's question:")
    for nm_, label in (("slot", "one slot"), ("w1", "+-1 px window (6x6)"), ("w2", "+-2 px window (8x8)")):
        print(f"  {label:22s}: {tot['waves_with_a_gather ' + nm_] / tot['waves ' + nm_]:.3f} of all, {tot['late waves_with_a_gather ' + nm_] / tot['late waves ' + nm_]:.3f} of the last three passes; "
              f"team members of 512 points: {tot['members_with_a_gather ' + nm_] / tot['members ' + nm_]:.3f} / {tot['late members_with_a_gather ' + nm_] / tot['late members ' + nm_]:.3f}")

if __name__ == "__main__":
    main()
