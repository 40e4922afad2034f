"""CPU oracle of the post-solve point maintenance — TEST INFRASTRUCTURE ONLY.

fp64 numpy restatement of reference ``Tracker::getCoord(delete_out_point)`` (src/tracking/Tracker.cpp:319-376), of the
index-aligned erase it triggers (``KeyFrame::erasePoint``, src/tracking/KeyFrame.cpp:1060-1106) and of
``Tracker::needNewKeyframe`` (Tracker.cpp:650-654).  PARITY UNPINNED (no reference tests exist).
"""
from __future__ import annotations

import numpy as np


def quat_to_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def get_coord(norm_coord, idp, coord, K, rows, cols, p, q, delete_out_point=True):
    """Returns dict(coord, tracks, kept, mean_sq_flow).

    p = R (x, y, 1)/mu + px with the RAW inverse depth mu (no 1e-5 here, :343-347); xp = fx p0/p2 + cx (:350-351);
    outlier = xp < 0 || xp > cols || yp < 0 || yp > rows (:354); kept points keep their order; track = new - old
    keyframe pixel (:364-365); squared_norm_flow = mean |track|^2 over the kept points (:366,372)."""
    fx, fy, cx, cy = K
    z = 1.0 / np.asarray(idp, dtype=np.float64)
    X = np.stack([norm_coord[:, 0] * z, norm_coord[:, 1] * z, z], axis=1)
    P = X @ quat_to_R(np.asarray(q, dtype=np.float64)).T + np.asarray(p, dtype=np.float64)
    xp = fx * P[:, 0] / P[:, 2] + cx
    yp = fy * P[:, 1] / P[:, 2] + cy
    outlier = (xp < 0.0) | (xp > cols) | (yp < 0.0) | (yp > rows)
    keep = ~(outlier & bool(delete_out_point))
    new = np.stack([xp, yp], axis=1)[keep]
    tracks = new - np.asarray(coord, dtype=np.float64)[keep]
    n = int(keep.sum())
    return dict(coord=new, tracks=tracks, kept=np.nonzero(keep)[0], mean_sq_flow=float(np.sum(tracks ** 2) / n) if n else 0.0)


def need_new_keyframe(mean_sq_flow, rows, cols, weight_factor=0.03):
    """Tracker::needNewKeyframe (Tracker.cpp:650-654); note the float sqrtf of the reference."""
    image_weight = (cols + rows) * weight_factor
    return bool(image_weight * float(np.sqrt(np.float32(mean_sq_flow))) / (cols + rows) > 1)
