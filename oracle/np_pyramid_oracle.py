"""CPU oracle of the coarse-to-fine path (BASELINE.json configs[3]) — TEST INFRASTRUCTURE ONLY, like everything under oracle/.

The reference's event tracker has no resolution pyramid; configs[3] is an extension patterned on the DSO-derived coarse tracker
the reference carries, and this file restates exactly the pieces that extension takes from it:
  * pyramid level l from level l-1 by 2x2 box averaging, in fp32, 0.25f * (a + b + c + d) in that order
    (reference src/tracking/HessianBlocks.cpp:173-176); sizes w >> l, h >> l (src/tracking/CoarseTracker.cpp:105-106)
  * level intrinsics fx_l = fx_{l-1} * 0.5, cx_l = (cx_0 + 0.5) / 2^l - 0.5 (CoarseTracker.cpp:107-110)
  * coarsest level first, pose (and velocity) carried to the next finer level (CoarseTracker.cpp:545-664)
The per-level alignment is the oracle's own (pyoracle.Oracle: pose6_lm / solve_lm).
"""
from __future__ import annotations

import numpy as np


def box_down(frame32: np.ndarray) -> np.ndarray:
    """One pyramid step on an fp32 image: (H >> 1) x (W >> 1), each pixel 0.25f * (((a + b) + c) + d) of its 2x2 block."""
    f = np.asarray(frame32, dtype=np.float32)
    H, W = f.shape[0] >> 1, f.shape[1] >> 1
    a, b = f[0:2 * H:2, 0:2 * W:2], f[0:2 * H:2, 1:2 * W:2]
    c, d = f[1:2 * H:2, 0:2 * W:2], f[1:2 * H:2, 1:2 * W:2]
    return (np.float32(0.25) * (((a + b) + c) + d)).astype(np.float32)


def build_pyramid(frame: np.ndarray, levels: int):
    """Level 0 is the frame as the library stores it (fp32); returns the list of fp32 levels."""
    out = [np.ascontiguousarray(frame, dtype=np.float32)]
    for _ in range(1, levels):
        out.append(box_down(out[-1]))
    return out


def level_intrinsics(level: int, fx: float, fy: float, cx: float, cy: float):
    for _ in range(level):
        fx *= 0.5
        fy *= 0.5
    if level:
        cx = (cx + 0.5) / float(1 << level) - 0.5
        cy = (cy + 0.5) / float(1 << level) - 0.5
    return fx, fy, cx, cy


def track(po, synth, al, point_counts, iters, solver="lm6", **okw):
    """Coarse-to-fine solve of alignment `al` (level-0 inputs): level l uses the first point_counts[l] points and iters[l]
    iterations.  Returns the final (p, q, v) and the per-level results (finest first), as eds_pyr_optimize reports them."""
    L = len(point_counts)
    pyr = build_pyramid(al.frame, L)
    p, q, v = al.p0.copy(), al.q0.copy(), al.v0.copy()
    per_level = [None] * L
    for l in range(L - 1, -1, -1):
        n = point_counts[l]
        fx, fy, cx, cy = level_intrinsics(l, al.fx, al.fy, al.cx, al.cy)
        a = synth.Alignment(H=pyr[l].shape[0], W=pyr[l].shape[1], fx=fx, fy=fy, cx=cx, cy=cy, norm_coord=al.norm_coord[:n],
                            grad=al.grad[:n], idp=al.idp[:n], weights=al.weights[:n], frame=pyr[l].astype(np.float64),
                            coord=al.coord[:n])
        if solver == "lm6":
            r = po.Oracle(a, **okw).pose6_lm(p, q, v, iters=iters[l], lambda0=0.01)
            p, q = r["p"], r["q"]
        else:
            r = po.Oracle(a, max_num_iterations=iters[l], **okw).solve_lm(p, q, v)
            if r["usable"]:
                p, q, v = r["p"], r["q"], r["v"]
        per_level[l] = r
    return p, q, v, per_level
