"""Second, independent CPU oracle: vectorised numpy fp64 with CLOSED-FORM Jacobians.

TEST INFRASTRUCTURE ONLY.  It exists to cross-check oracle/eds_oracle.hpp (which
gets its Jacobians from forward-mode autodiff of a restatement of reference
src/tracking/PhotometricError.hpp:124-182): the two share no code and no
derivation technique, and both are compared with central finite differences in
tests/test_oracle.py.  Parity with the real reference stays UNPINNED (no
reference tests/golden vectors exist — SURVEY.md §4, §8c).

The closed forms are those of SURVEY.md §8(a):
  a_i = -(gx_i df0/dv + gy_i df1/dv),  m = A v,  S = v^T G v + 1e-3 (per block),
  r_i = w_i (m_i / sqrt(S) - E(u_i, v_i)),
  J_t = -w_i gradE_P,  J_theta = -2 w_i (R X_i) x gradE_P   (Ceres local quaternion),
  J_v = w_i (a_i / n - m_i (G v)^T / n^3) (I - v v^T / |v|^2) / |v|,
  J_xi = -w_i [gradE_P, P x gradE_P]                        (SE(3) left perturbation).
"""
from __future__ import annotations

import numpy as np

EPS = 1e-5          # PhotometricError.hpp:200
S0 = 1e-3           # PhotometricError.hpp:132


def quat_to_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def flow_matrix(al):
    """A (N x 6): m = A v  (PhotometricError.hpp:114-122,136-143)."""
    x, y = al.norm_coord[:, 0], al.norm_coord[:, 1]
    rho = al.idp
    one, zero = np.ones_like(x), np.zeros_like(x)
    f0 = np.stack([-rho, zero, x * rho, x * y, -(one + x * x), y], axis=1)
    f1 = np.stack([zero, -rho, y * rho, one + y * y, -x * y, -x], axis=1)
    return -(al.grad[:, :1] * f0 + al.grad[:, 1:] * f1)


def _hermite(p0, p1, p2, p3, x):
    a = 0.5 * (-p0 + 3 * p1 - 3 * p2 + p3)
    b = 0.5 * (2 * p0 - 5 * p1 + 4 * p2 - p3)
    c = 0.5 * (-p0 + p2)
    return p1 + x * (c + x * (b + x * a)), c + x * (2 * b + 3 * a * x)


def bicubic(frame, row, col):
    """Vectorised ceres::BiCubicInterpolator over a clamped Grid2D: f, df/drow, df/dcol."""
    H, W = frame.shape
    r0 = np.floor(row).astype(np.int64)
    c0 = np.floor(col).astype(np.int64)
    fr, fc = row - r0, col - c0
    vals, dcs = [], []
    for k in range(-1, 3):
        rr = np.clip(r0 + k, 0, H - 1)
        taps = [frame[rr, np.clip(c0 + j, 0, W - 1)] for j in range(-1, 3)]
        f, d = _hermite(*taps, fc)
        vals.append(f)
        dcs.append(d)
    f, dfdr = _hermite(*vals, fr)
    dfdc, _ = _hermite(*dcs, fr)
    return f, dfdr, dfdc


def bilinear(frame, row, col):
    H, W = frame.shape
    r0 = np.floor(row).astype(np.int64)
    c0 = np.floor(col).astype(np.int64)
    dy, dx = row - r0, col - c0
    g = lambda r, c: frame[np.clip(r, 0, H - 1), np.clip(c, 0, W - 1)]
    tl, tr, bl, br = g(r0, c0), g(r0, c0 + 1), g(r0 + 1, c0), g(r0 + 1, c0 + 1)
    f = (1 - dy) * ((1 - dx) * tl + dx * tr) + dy * ((1 - dx) * bl + dx * br)
    return f, (1 - dx) * (bl - tl) + dx * (br - tr), (1 - dy) * (tr - tl) + dy * (br - bl)


def block_ranges(N, num_blocks):
    """Tracker.cpp:178-195."""
    ne = N // num_blocks
    out = []
    for b in range(num_blocks):
        n = ne + (N - (b + 1) * ne if b + 1 == num_blocks else 0)
        out.append((b * ne, n))
    return out


def project(al, p, q):
    z = 1.0 / (al.idp + EPS)
    X = np.stack([al.norm_coord[:, 0] * z, al.norm_coord[:, 1] * z, z], axis=1)
    RX = X @ quat_to_R(q).T
    P = RX + np.asarray(p)
    u = al.fx * P[:, 0] / P[:, 2] + al.cx
    v = al.fy * P[:, 1] / P[:, 2] + al.cy
    return RX, P, u, v


def residual(al, p, q, v, num_blocks=1, sampling="bicubic", nc=False):
    """nc=True: PhotometricErrorNC.hpp:151-186 - the sampled brightness is L2-normalised per block as well."""
    A = flow_matrix(al)
    m = A @ np.asarray(v)
    nrm = np.empty(al.N)
    for s, n in block_ranges(al.N, num_blocks):
        nrm[s:s + n] = np.sqrt(S0 + np.sum(m[s:s + n] ** 2))
    _, _, u, vv = project(al, p, q)
    E = (bicubic if sampling == "bicubic" else bilinear)(al.frame, vv, u)[0]
    if nc:
        E = E.copy()
        for s, n in block_ranges(al.N, num_blocks):
            E[s:s + n] /= np.sqrt(S0 + np.sum(E[s:s + n] ** 2))
    return al.weights * (m / nrm - E)


def jacobians(al, p, q, v, num_blocks=1, sampling="bicubic", nc=False):
    """Closed-form r, J_local (N x 12, Ceres local coordinates) and J_se3 (N x 6).
    nc=True (PhotometricErrorNC): E_i -> E_i/||E||_block, so the pose columns become
    w_i (J'_i/||E|| - E_i sum_j E_j J'_j / ||E||^3) with J'_j = -dE_j/d(pose) (un-weighted)."""
    v = np.asarray(v, dtype=np.float64)
    A = flow_matrix(al)
    m = A @ v
    w = al.weights
    RX, P, u, vv = project(al, p, q)
    E, E_row, E_col = (bicubic if sampling == "bicubic" else bilinear)(al.frame, vv, u)
    iz = 1.0 / P[:, 2]
    gP = np.stack([E_col * al.fx * iz,
                   E_row * al.fy * iz,
                   -(E_col * al.fx * P[:, 0] + E_row * al.fy * P[:, 1]) * iz * iz], axis=1)
    J = np.zeros((al.N, 12))
    J[:, 0:3] = -w[:, None] * gP
    J[:, 3:6] = -2.0 * w[:, None] * np.cross(RX, gP)
    vn = np.linalg.norm(v)
    proj = (np.eye(6) - np.outer(v, v) / (vn * vn)) / vn
    r = np.empty(al.N)
    for s, n in block_ranges(al.N, num_blocks):
        Ab, mb = A[s:s + n], m[s:s + n]
        G = Ab.T @ Ab
        S = v @ G @ v + S0
        nn = np.sqrt(S)
        Jv = Ab / nn - np.outer(mb, G @ v) / nn ** 3
        J[s:s + n, 6:12] = w[s:s + n, None] * (Jv @ proj)
        if nc:
            Eb = E[s:s + n]
            Jp = np.concatenate([-gP[s:s + n], -2.0 * np.cross(RX[s:s + n], gP[s:s + n])], axis=1)   # un-weighted
            SE = S0 + np.sum(Eb ** 2)
            nE = np.sqrt(SE)
            J[s:s + n, 0:6] = w[s:s + n, None] * (Jp / nE - np.outer(Eb, Eb @ Jp) / nE ** 3)
            r[s:s + n] = w[s:s + n] * (mb / nn - Eb / nE)
        else:
            r[s:s + n] = w[s:s + n] * (mb / nn - E[s:s + n])
    J_se3 = np.concatenate([-w[:, None] * gP, -w[:, None] * np.cross(P, gP)], axis=1)   # plain residual only
    return r, J, J_se3


def quat_mul(a, b):
    x1, y1, z1, w1 = a
    x2, y2, z2, w2 = b
    return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                     w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])


def state_plus(p, q, v, d):
    """Ceres Plus over (p additive, EigenQuaternionParameterization, UnitNormVectorAddition)."""
    d = np.asarray(d, dtype=np.float64)
    p2 = np.asarray(p) + d[0:3]
    nd = np.linalg.norm(d[3:6])
    if nd > 0:
        qd = np.concatenate([np.sin(nd) / nd * d[3:6], [np.cos(nd)]])
        q2 = quat_mul(qd, np.asarray(q))
    else:
        q2 = np.asarray(q, dtype=np.float64).copy()
    v2 = np.asarray(v) + d[6:12]
    v2 = v2 / np.linalg.norm(v2)
    return p2, q2, v2


def se3_exp_matrix(xi):
    """exp of the 4x4 twist by scaling-and-squaring-free series (independent of Sophus' closed form)."""
    ups, om = xi[:3], xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    M[:3, 3] = ups
    out, term = np.eye(4), np.eye(4)
    for k in range(1, 30):
        term = term @ M / k
        out = out + term
    return out


def fd_jacobian_local(al, p, q, v, num_blocks=1, h=1e-6, sampling="bicubic", nc=False):
    """Central finite differences of the residual in Ceres local coordinates (N x 12)."""
    J = np.zeros((al.N, 12))
    for k in range(12):
        d = np.zeros(12)
        d[k] = h
        rp = residual(al, *state_plus(p, q, v, d), num_blocks, sampling, nc)
        rm = residual(al, *state_plus(p, q, v, -d), num_blocks, sampling, nc)
        J[:, k] = (rp - rm) / (2 * h)
    return J


def fd_jacobian_se3(al, p, q, v, num_blocks=1, h=1e-6, sampling="bicubic"):
    """Central finite differences w.r.t. the SE(3) left perturbation T <- exp(xi) T (N x 6)."""
    T = np.eye(4)
    T[:3, :3] = quat_to_R(q)
    T[:3, 3] = p
    z = 1.0 / (al.idp + EPS)
    X = np.stack([al.norm_coord[:, 0] * z, al.norm_coord[:, 1] * z, z, np.ones(al.N)], axis=0)
    A = flow_matrix(al)
    m = A @ np.asarray(v)
    nrm = np.empty(al.N)
    for s, n in block_ranges(al.N, num_blocks):
        nrm[s:s + n] = np.sqrt(S0 + np.sum(m[s:s + n] ** 2))
    samp = bicubic if sampling == "bicubic" else bilinear

    def res(Tm):
        P = (Tm @ X)[:3].T
        u = al.fx * P[:, 0] / P[:, 2] + al.cx
        vv = al.fy * P[:, 1] / P[:, 2] + al.cy
        return al.weights * (m / nrm - samp(al.frame, vv, u)[0])

    J = np.zeros((al.N, 6))
    for k in range(6):
        xi = np.zeros(6)
        xi[k] = h
        J[:, k] = (res(se3_exp_matrix(xi) @ T) - res(se3_exp_matrix(-xi) @ T)) / (2 * h)
    return J
