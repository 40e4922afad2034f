"""Property tests (hypothesis) of the oracle's third-party restatements: they cannot pin the oracle to the reference (nothing
can, here), but they pin it to the mathematics those libraries document — Ceres' cubic interpolation, loss functions and
local parameterisations, Sophus' exp / log."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

finite = dict(allow_nan=False, allow_infinity=False)
small = st.floats(min_value=-1.0, max_value=1.0, **finite)
S = settings(max_examples=60, deadline=None, derandomize=True, database=None)     # deterministic: the same examples on every run


@S
@given(st.lists(small, min_size=6, max_size=6), st.floats(min_value=1e-3, max_value=3.0, **finite))
def test_se3_exp_log_round_trip(po, xi, scale):
    xi = np.array(xi) * np.array([1, 1, 1, scale, scale, scale])
    if np.linalg.norm(xi[3:]) > 3.0:
        xi[3:] *= 3.0 / np.linalg.norm(xi[3:])
    t, q = po.se3_exp(xi)
    assert abs(np.linalg.norm(q) - 1.0) < 1e-12
    back = po.se3_log(t, q)
    # (Sophus switches to the small-angle form only below 1e-10: around |omega| ~ 1e-6 its (theta - sin theta)/theta^3 loses
    # seven digits, which the restatement reproduces on purpose)
    assert np.abs(back - xi).max() < 1e-7
    t2, q2 = po.se3_exp(-xi)                               # exp(-xi) is the inverse: exp(xi) exp(-xi) = identity
    t3, q3 = po.se3_left_update(xi, t2, q2)
    assert np.abs(t3).max() < 1e-8 and min(np.abs(q3 - [0, 0, 0, 1]).max(), np.abs(q3 + [0, 0, 0, 1]).max()) < 1e-12


@S
@given(st.lists(small, min_size=4, max_size=4), st.lists(small, min_size=3, max_size=3))
def test_quaternion_plus_keeps_unit_norm_and_composes_rotations(po, q, d):
    q = np.array(q)
    if np.linalg.norm(q) < 1e-3:
        q = np.array([0.1, 0.2, -0.3, 0.9])
    q /= np.linalg.norm(q)
    d = 0.5 * np.array(d)
    x13 = np.concatenate([np.zeros(3), q, np.array([1, 0, 0, 0, 0, 0.0])])
    out = po.state_plus(x13, np.concatenate([np.zeros(3), d, np.zeros(6)]))
    q2 = out[3:7]
    assert abs(np.linalg.norm(q2) - 1.0) < 1e-12
    # EigenQuaternionParameterization: q <- q_delta * q with q_delta a rotation by 2 |d| about d  (Tracker.cpp:111-112,197)
    R, R2 = po.quat_to_R(q), po.quat_to_R(q2)
    dR = R2 @ R.T
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    assert ang == pytest.approx(min(2 * np.linalg.norm(d), 2 * np.pi - 2 * np.linalg.norm(d)), abs=1e-7)


@S
@given(st.lists(small, min_size=6, max_size=6), st.lists(small, min_size=6, max_size=6))
def test_unit_velocity_plus_stays_on_the_sphere_and_its_jacobian_annihilates_v(po, v, d):
    v = np.array(v)
    if np.linalg.norm(v) < 1e-2:
        v = np.ones(6)
    v /= np.linalg.norm(v)
    out = po.state_plus(np.concatenate([np.zeros(3), [0, 0, 0, 1.0], v]), np.concatenate([np.zeros(6), 0.3 * np.array(d)]))
    assert abs(np.linalg.norm(out[7:]) - 1.0) < 1e-12
    P = po.unit_plus_jacobian(v)                            # (I - v v^T/|v|^2)/|v|: rank 5, v in its null space (PhotometricError.hpp:32-54)
    assert np.abs(P @ v).max() < 1e-12 and np.linalg.matrix_rank(P, tol=1e-9) == 5


@S
@given(st.integers(min_value=0, max_value=2), st.floats(min_value=1e-3, max_value=10.0, **finite), st.floats(min_value=0.0, max_value=1e4, **finite))
def test_loss_functions_are_concave_robustifiers(po, kind, a, s):
    rho = po.loss_eval(kind, a, s)
    assert rho[0] <= s + 1e-12 * max(s, 1.0) and rho[0] >= 0.0           # rho(s) <= s
    assert 0.0 < rho[1] <= 1.0 and rho[2] <= 0.0                            # 0 < rho' <= 1, rho'' <= 0
    if s <= a * a:
        assert rho[0] == pytest.approx(s, rel=1e-12) or kind == 2           # Huber and trivial loss are the identity inside the threshold
    h = 1e-5 * min(max(s, a * a), a * a * 10)               # well inside the scale a^2 on which the loss bends
    num = (po.loss_eval(kind, a, s + h)[0] - po.loss_eval(kind, a, max(s - h, 0.0))[0]) / (s + h - max(s - h, 0.0))
    if not (kind == 1 and abs(s - a * a) < 2 * h):          # Huber's kink
        assert num == pytest.approx(rho[1], rel=1e-3, abs=1e-6)             # rho' is the derivative of rho


@S
@given(st.floats(min_value=1.5, max_value=20.5, **finite), st.floats(min_value=1.5, max_value=26.5, **finite),
       st.lists(st.floats(min_value=-2, max_value=2, **finite), min_size=3, max_size=3))
def test_bicubic_reproduces_affine_images_exactly(po, r, c, abc):
    a, b, c0 = abc
    rr, cc = np.meshgrid(np.arange(24.0), np.arange(30.0), indexing="ij")
    img = a * rr + b * cc + c0
    f, fr, fc = po.bicubic(img, r, c)
    assert f == pytest.approx(a * r + b * c + c0, abs=1e-9) and fr == pytest.approx(a, abs=1e-9) and fc == pytest.approx(b, abs=1e-9)
