"""Deterministic synthetic inputs for the event-to-model tracker.

A synthetic alignment is a keyframe point set (what ``KeyFrame`` hands to
``Tracker::optimize``: reference src/tracking/KeyFrame.hpp:80-96) plus one
normalised brightness-increment frame (``EventFrame::event_frame[level]``,
reference src/tracking/EventFrame.cpp:359-383) generated from a ground-truth
pose/velocity so that tracking has a known answer.

Workload generation only — no tracker arithmetic lives here.  Pure numpy so the
same bytes are produced in the build container and on the GPU box.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

__all__ = ["Alignment", "make_alignment", "intrinsics", "gaussian_blur", "quat_from_axis_angle"]


@dataclass
class Alignment:
    """One (keyframe, event-frame) pair.  Arrays are C-contiguous float64."""
    H: int
    W: int
    fx: float
    fy: float
    cx: float
    cy: float
    norm_coord: np.ndarray          # N x 2   (coord - c) / f      KeyFrame.cpp:417-423
    grad: np.ndarray                # N x 2   log-image gradient   KeyFrame.cpp:426-430
    idp: np.ndarray                 # N       inverse depth (mu)   DepthPoints.cpp:230-237
    weights: np.ndarray             # N       in (0.7, 1]          KeyFrame.cpp:451,1168-1181
    frame: np.ndarray               # H x W   frame / ||frame||_F  EventFrame.cpp:359-383
    coord: np.ndarray               # N x 2   integer pixel coordinates (x, y)
    # ground truth used to synthesise the frame
    p_true: np.ndarray = field(default_factory=lambda: np.zeros(3))
    q_true: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0, 1.0]))   # xyzw
    v_true: np.ndarray = field(default_factory=lambda: np.zeros(6))
    # suggested start (Tracker ctor: p = 0, q = identity; Tracker.cpp:43-46)
    p0: np.ndarray = field(default_factory=lambda: np.zeros(3))
    q0: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0, 1.0]))
    v0: np.ndarray = field(default_factory=lambda: np.zeros(6))

    @property
    def N(self) -> int:
        return int(self.idp.shape[0])


def intrinsics(H: int, W: int):
    """fx = fy = 0.78125 W (500 @ 640, 1000 @ 1280); principal point at the centre."""
    f = 0.78125 * W
    return f, f, (W - 1) / 2.0, (H - 1) / 2.0


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = np.sin(0.5 * angle)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, np.cos(0.5 * angle)])


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def gaussian_blur(img, ksize: int, sigma: float):
    """Separable Gaussian with reflect-101 borders (what cv::GaussianBlur does by
    default; reference call site src/utils/Utils.cpp:113-119)."""
    if ksize <= 1 or sigma <= 0:
        return img
    r = ksize // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    k /= k.sum()
    pad = np.pad(img, ((r, r), (r, r)), mode="reflect")
    tmp = np.zeros((img.shape[0] + 2 * r, img.shape[1]))
    for i, kv in enumerate(k):
        tmp += kv * pad[:, i:i + img.shape[1]]
    out = np.zeros_like(img)
    for i, kv in enumerate(k):
        out += kv * tmp[i:i + img.shape[0], :]
    return out


def _flow_rows(x, y, rho):
    """d flow / d v, two N x 6 matrices (reference PhotometricError.hpp:114-122)."""
    one = np.ones_like(x)
    zero = np.zeros_like(x)
    f0 = np.stack([-rho, zero, x * rho, x * y, -(one + x * x), y], axis=1)
    f1 = np.stack([zero, -rho, y * rho, one + y * y, -x * y, -x], axis=1)
    return f0, f1


def render_frame(H, W, K, norm_coord, grad, idp, p, q, v, *, blur_ksize: int = 7, blur_sigma: float = 1.5, noise: float = 0.05, rng=None):
    """The brightness-increment frame the event camera would integrate for keyframe points seen under pose (p, q) with
    velocity v: the normalised model m_hat = A v / ||A v|| splatted at the projections with 4-tap bilinear voting
    (Utils.cpp:83-107), blurred, optionally noised, divided by its Frobenius norm (EventFrame.cpp:359-383)."""
    fx, fy, cx, cy = K
    x, y = norm_coord[:, 0], norm_coord[:, 1]
    f0, f1 = _flow_rows(x, y, idp)
    A = -(grad[:, :1] * f0 + grad[:, 1:] * f1)
    m = A @ v
    m_hat = m / np.sqrt(1e-3 + np.sum(m * m))       # global norm = one block
    z = 1.0 / (idp + 1e-5)
    P = (_quat_to_R(q) @ np.stack([x * z, y * z, z], axis=0)).T + p
    u = fx * P[:, 0] / P[:, 2] + cx
    vv = fy * P[:, 1] / P[:, 2] + cy
    img = np.zeros((H, W))
    x0 = np.floor(u).astype(np.int64)
    y0 = np.floor(vv).astype(np.int64)
    ax = u - x0
    ay = vv - y0
    for dx, dy, wgt in ((0, 0, (1 - ax) * (1 - ay)), (0, 1, (1 - ax) * ay), (1, 0, ax * (1 - ay)), (1, 1, ax * ay)):
        xi, yi = x0 + dx, y0 + dy
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        np.add.at(img, (yi[ok], xi[ok]), (wgt * m_hat)[ok])
    img = gaussian_blur(img, blur_ksize, blur_sigma)
    if noise > 0:
        img = img + (rng or np.random.default_rng(0)).normal(0.0, noise * np.abs(img).max(), size=img.shape)
    return img / np.linalg.norm(img)


def make_alignment(seed: int = 1234, H: int = 480, W: int = 640, N: int = 2000, *,
                   rot_deg: float = 0.2, trans_norm: float = 0.004,
                   blur_ksize: int = 7, blur_sigma: float = 1.5, noise: float = 0.05,
                   unit_weights: bool = False, start: str = "truth_velocity", margin: int = 16,
                   layout: str = "uniform") -> Alignment:
    """Build one deterministic alignment (numpy PCG64, ``default_rng(seed)``).

    ``rot_deg`` / ``trans_norm`` set the ground-truth offset of the event frame
    from the keyframe; ``blur_*`` the splat blur (SURVEY §8d names the
    reference's 3x3, sigma 0.5 — pass ``blur_ksize=3, blur_sigma=0.5`` for that;
    the defaults widen the basin so Gauss-Newton converges from identity).
    ``start``: "truth_velocity" (pose-only mode, v = v*) or "ctor"
    (v = normalize(0.001 * ones), Tracker.cpp:45-46).
    ``layout``: "uniform" (distinct pixels uniform over the frame) or
    "edges" — the points strung along ~40 random contours, 0-2 pixels off them, the
    way a gradient-selected keyframe of a real scene looks (KeyFrame.cpp:740-823 keeps
    the strongest gradients of every cell, i.e. edge pixels).  Same raster order of the
    20x20 cells, same everything else: only where the points sit differs.
    """
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = intrinsics(H, W)
    # distinct integer pixels in [margin, W-1-margin] x [margin, H-1-margin]
    w_in, h_in = W - 2 * margin, H - 2 * margin
    if layout == "uniform":
        flat = rng.choice(w_in * h_in, size=N, replace=False)
    elif layout == "edges":
        seen, chosen = set(), []
        while len(chosen) < N:
            # one contour: a quadratic Bezier arc between two random points, sampled every ~0.7 pixel, each sample jittered by 0-2 pixels
            a, b, c = (rng.uniform([0, 0], [w_in - 1, h_in - 1]) for _ in range(3))
            n_s = int(1.5 * (np.linalg.norm(b - a) + np.linalg.norm(c - b))) + 2
            t = np.linspace(0.0, 1.0, n_s)[:, None]
            xy = (1 - t) ** 2 * a + 2 * (1 - t) * t * b + t ** 2 * c + rng.integers(-2, 3, size=(n_s, 2))
            xy = np.clip(np.rint(xy).astype(np.int64), [0, 0], [w_in - 1, h_in - 1])
            take = max(1, N // 40)
            for x_, y_ in xy[rng.permutation(n_s)]:
                f_ = int(y_) * w_in + int(x_)
                if f_ not in seen:
                    seen.add(f_); chosen.append(f_); take -= 1
                    if take == 0 or len(chosen) == N:
                        break
        flat = np.asarray(chosen[:N], dtype=np.int64)
    else:
        raise ValueError(f"layout must be 'uniform' or 'edges', not {layout!r}")
    px = (flat % w_in + margin).astype(np.float64)
    py = (flat // w_in + margin).astype(np.float64)
    # the reference selects points patch by patch over a grid scanned row-major (KeyFrame.cpp:752-787,
    # 20x20 cells), so a real keyframe's point vectors are in raster order of their grid cell
    order = np.argsort((py // 20) * 4096 + (px // 20), kind="stable")
    px, py = px[order], py[order]
    coord = np.stack([px, py], axis=1)
    norm_coord = np.stack([(px - cx) / fx, (py - cy) / fy], axis=1)
    idp = rng.uniform(0.2, 1.0, size=N)
    grad = rng.standard_normal((N, 2))
    grad /= np.median(np.linalg.norm(grad, axis=1))
    weights = np.ones(N) if unit_weights else 1.0 - rng.uniform(0.0, 0.3, size=N)   # (0.7, 1]
    v_true = rng.standard_normal(6)
    v_true /= np.linalg.norm(v_true)
    axis = rng.standard_normal(3)
    q_true = quat_from_axis_angle(axis, np.deg2rad(rot_deg))
    t_dir = rng.standard_normal(3)
    p_true = trans_norm * t_dir / np.linalg.norm(t_dir)

    img = render_frame(H, W, (fx, fy, cx, cy), norm_coord, grad, idp, p_true, q_true, v_true,
                       blur_ksize=blur_ksize, blur_sigma=blur_sigma, noise=noise, rng=rng)

    if start == "ctor":
        v0 = np.full(6, 0.001)
        v0 /= np.linalg.norm(v0)
    else:
        v0 = v_true.copy()
    return Alignment(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy,
                     norm_coord=np.ascontiguousarray(norm_coord), grad=np.ascontiguousarray(grad),
                     idp=np.ascontiguousarray(idp), weights=np.ascontiguousarray(weights),
                     frame=np.ascontiguousarray(img), coord=coord,
                     p_true=p_true, q_true=q_true, v_true=v_true,
                     p0=np.zeros(3), q0=np.array([0.0, 0.0, 0.0, 1.0]), v0=v0)
