"""CPU tests of the PRODUCT's host-side logic (slam-eds_amd/csrc/eds_math.hpp, eds_solver.hpp).

These headers are plain C++ outside hipcc, so the solver state machines that drive the GPU (and
run inside the persistent kernel) are compiled here with g++ and fed with reduced sums from the
oracle's evaluator: they must then retrace the oracle's solvers step for step.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_logic", "harness.cpp")
LIB = os.path.join(HERE, "host_logic", "libhost_logic.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class _Pb(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("_pad", C.c_int32), ("grad", _dp),
                ("norm_coord", _dp), ("idp", _dp), ("weights", _dp), ("frame", _dp), ("fx", C.c_double),
                ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


@pytest.fixture(scope="module")
def hl():
    deps = [SRC] + [os.path.join(HERE, "..", p) for p in ("oracle/eds_oracle.hpp", "slam-eds_amd/csrc/eds_math.hpp",
                                                         "slam-eds_amd/csrc/eds_solver.hpp", "slam-eds_amd/csrc/eds_layout.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-misleading-indentation", "-o", LIB, SRC])
    return C.CDLL(LIB)


def _d(a):
    return a.ctypes.data_as(_dp)


def _problem(al):
    keep = [np.ascontiguousarray(x, dtype=np.float64) for x in (al.grad, al.norm_coord, al.idp, al.weights, al.frame)]
    pb = _Pb(al.N, al.H, al.W, 0, *[_d(k) for k in keep], al.fx, al.fy, al.cx, al.cy)
    return pb, keep


@pytest.fixture(scope="module")
def al(synth):
    return synth.make_alignment(33, H=96, W=128, N=400)


@pytest.mark.parametrize("nb,loss", [(1, 0), (3, 1), (4, 2)])
def test_solver12_retraces_oracle_lm(hl, po, al, nb, loss):
    pb, keep = _problem(al)
    ref = po.Oracle(al, num_blocks=nb, loss_type=loss, loss_param=0.25, max_num_iterations=12).solve_lm(al.p0, al.q0, al.v0)
    p, q, v = al.p0.copy(), al.q0.copy(), al.v0.copy()
    out5, costs = np.zeros(5, dtype=np.int32), np.zeros(2)
    rc = hl.hl_solver12_run(C.byref(pb), 0, nb, loss, C.c_double(0.25), 12, C.c_double(1e-6), C.c_double(1e-8),
                            C.c_double(1e-6), _d(p), _d(q), _d(v), out5.ctypes.data_as(_ip), _d(costs))
    assert rc == 0 and ref["usable"]
    assert out5[0] == ref["termination"]
    assert out5[1] == ref["num_successful_steps"] and out5[2] == ref["num_unsuccessful_steps"]
    assert costs[0] == pytest.approx(ref["initial_cost"], rel=1e-12)
    assert costs[1] == pytest.approx(ref["final_cost"], rel=1e-9)
    assert po.se3_distance(p, q, ref["p"], ref["q"]) < 1e-9
    assert np.abs(v - ref["v"]).max() < 1e-9
    # one pass per LM iteration plus the final residual pass
    assert out5[3] <= ref["num_iterations"] + 1


def test_solver12_zero_iterations_and_failure(hl, po, al):
    pb, keep = _problem(al)
    p, q, v = al.p0.copy(), al.q0.copy(), al.v0.copy()
    out5, costs = np.zeros(5, dtype=np.int32), np.zeros(2)
    hl.hl_solver12_run(C.byref(pb), 0, 1, 0, C.c_double(1.0), 0, C.c_double(1e-6), C.c_double(1e-8), C.c_double(1e-6),
                       _d(p), _d(q), _d(v), out5.ctypes.data_as(_ip), _d(costs))
    assert out5[0] == 1 and out5[1] == 1 and out5[2] == 0           # NO_CONVERGENCE, iteration 0 counted successful
    assert np.array_equal(p, al.p0) and np.array_equal(v, al.v0)
    bad = type(al)(**{**al.__dict__, "frame": np.full_like(al.frame, np.nan)})
    pb2, keep2 = _problem(bad)
    rc = hl.hl_solver12_run(C.byref(pb2), 0, 1, 0, C.c_double(1.0), 5, C.c_double(1e-6), C.c_double(1e-8),
                            C.c_double(1e-6), _d(p), _d(q), _d(v), out5.ctypes.data_as(_ip), _d(costs))
    assert rc == -1 and out5[0] == 2                                 # FAILURE, state untouched
    assert np.array_equal(p, al.p0) and np.array_equal(q, al.q0)


@pytest.mark.parametrize("damped", [0, 1])
@pytest.mark.parametrize("tau", [0.0, 0.02])
def test_solver6_retraces_oracle(hl, po, al, damped, tau):
    pb, keep = _problem(al)
    o = po.Oracle(al)
    ref = o.pose6_lm(al.p0, al.q0, al.v0, iters=8, lambda0=0.01, huber_tau=tau) if damped else \
        o.pose6_gn(al.p0, al.q0, al.v0, iters=8, huber_tau=tau)
    p, q = al.p0.copy(), al.q0.copy()
    inc, costs, acc, out3 = np.zeros((128, 6)), np.zeros(128), np.zeros(128, dtype=np.int32), np.zeros(3, dtype=np.int32)
    hl.hl_solver6_run(C.byref(pb), 0, 1, damped, 8, C.c_double(0.01), C.c_double(tau), _d(p), _d(q), _d(al.v0), _d(inc),
                      _d(costs), acc.ctypes.data_as(_ip), out3.ctypes.data_as(_ip))
    n = out3[0]
    assert n == ref["iterations"] == 8 and out3[2] == 0
    assert out3[1] == (8 + 2 if damped else 8 + 1)                   # passes: (initial +) iterations + final
    assert np.allclose(inc[:n], ref["increments"], rtol=1e-9, atol=1e-15)
    assert np.allclose(costs[:n], ref["costs"], rtol=1e-12)
    if damped:
        assert np.array_equal(acc[:n], ref["accepted"])
    assert po.se3_distance(p, q, ref["p"], ref["q"]) < 1e-12


def test_math_helpers_match_oracle(hl, po):
    rng = np.random.default_rng(0)
    for scale in (1e-13, 1e-5, 0.3):
        xi = scale * rng.standard_normal(6)
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        t = rng.standard_normal(3)
        t2, q2 = t.copy(), q.copy()
        hl.hl_se3_left_update(_d(xi), _d(t2), _d(q2))
        tr, qr = po.se3_left_update(xi, t, q)
        assert np.allclose(t2, tr, atol=1e-14) and np.allclose(q2, qr, atol=1e-14)
    x = np.concatenate([rng.standard_normal(3), q, rng.standard_normal(6)])
    d = 0.01 * rng.standard_normal(12)
    po_, qo, vo = np.zeros(3), np.zeros(4), np.zeros(6)
    hl.hl_state_plus12(_d(x[:3].copy()), _d(x[3:7].copy()), _d(x[7:].copy()), _d(d), _d(po_), _d(qo), _d(vo))
    assert np.allclose(np.concatenate([po_, qo, vo]), po.state_plus(x, d), atol=1e-15)
    A = rng.standard_normal((12, 12)); A = A @ A.T + 12 * np.eye(12); b = rng.standard_normal(12); xs = np.zeros(12)
    assert hl.hl_cholesky(12, _d(np.ascontiguousarray(A)), _d(b), _d(xs)) == 1
    assert np.allclose(xs, np.linalg.solve(A, b), rtol=1e-11)
    Abad = -np.eye(6)
    assert hl.hl_cholesky(6, _d(np.ascontiguousarray(Abad)), _d(b[:6].copy()), _d(xs)) == 0
    for t_, a, s in ((1, 0.5, 0.1), (1, 0.5, 4.0), (2, 0.5, 1.0), (0, 1.0, 2.0)):
        out = np.zeros(2)
        hl.hl_loss_eval(t_, C.c_double(a), C.c_double(s), _d(out))
        assert np.allclose(out, po.loss_eval(t_, a, s)[:2], rtol=1e-15)


def test_pose_block_closed_form_norm(hl, npo, synth):
    # 1/n and G v / n^3 from the Gram matrix must equal the two-pass sums of the reference functor
    al = synth.make_alignment(2, H=48, W=64, N=90)
    A = npo.flow_matrix(al)
    nb = 3
    G = np.zeros((16, 36))
    for k, (s, n) in enumerate(npo.block_ranges(al.N, nb)):
        G[k] = (A[s:s + n].T @ A[s:s + n]).ravel()
    v = al.v_true
    pbk = np.zeros(hl.hl_pose_stride())
    hl.hl_fill_pose_block(_d(al.p0), _d(al.q0), _d(v), _d(np.ascontiguousarray(G)), nb, _d(pbk))
    for k, (s, n) in enumerate(npo.block_ranges(al.N, nb)):
        m = A[s:s + n] @ v
        S = 1e-3 + np.sum(m * m)
        blk = pbk[68 + 8 * k: 68 + 8 * k + 8]
        assert blk[0] == pytest.approx(1 / np.sqrt(S), rel=1e-12)
        assert np.allclose(blk[1:7], (A[s:s + n].T @ m) / S ** 1.5, rtol=1e-10)
    assert np.allclose(pbk[0:9].reshape(3, 3), np.eye(3))
    assert np.allclose(pbk[32:68].reshape(6, 6), np.eye(6) - np.outer(v, v), atol=1e-14)


@pytest.mark.parametrize("H,W", [(480, 640), (61, 83), (4, 4), (5, 1023)])
@pytest.mark.parametrize("tiled", [1, 0])
def test_frame_allocation_index_is_a_bijection(hl, H, W, tiled):
    """eds_layout.hpp: padded extent + one-tile margin; every logical pixel of the allocation maps to its own element,
    and a 4x4 tile is 16 consecutive floats (one 64-byte sector)."""
    Hp, Wp, idx = C.c_int(0), C.c_int(0), C.c_longlong(0)
    hl.hl_frame_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    assert hl.hl_frame_layout(H, W, tiled, C.byref(Hp), C.byref(Wp), 0, 0, C.byref(idx)) == 0
    assert Hp.value == ((H + 3) // 4) * 4 + 8 and Wp.value == ((W + 3) // 4) * 4 + 8
    origin = idx.value
    assert origin == ((Wp.value // 4) + 1) * 16 if tiled else origin == 4 * Wp.value + 4       # one tile row down, one tile right
    if tiled:
        got = []
        for r in range(4):
            for c in range(4):
                hl.hl_frame_layout(H, W, 1, C.byref(Hp), C.byref(Wp), r, c, C.byref(idx))
                got.append(idx.value - origin)
        assert got == list(range(16))
        hl.hl_frame_layout(H, W, 1, C.byref(Hp), C.byref(Wp), -1, -1, C.byref(idx))           # margin pixel: last element of the tile up-left
        assert idx.value == origin - (Wp.value // 4 + 1) * 16 + 15


@pytest.mark.parametrize("H,W", [(480, 640), (61, 83), (720, 1280)])
@pytest.mark.parametrize("phases", [1, 2, 4])
def test_strip_copies_hold_every_patch_in_place(hl, H, W, phases):
    """eds_layout.hpp, round 3: for every patch position a kernel can ask for, the 16 taps sit at eds_strips_row_offset + 32 k + 4 j of
    the strip copies (built here with the conversion kernel's rule); with 4 row phases every patch starts on a 128-byte boundary, i.e.
    is exactly ONE L2 line — the property the layout exists for; with 1 phase 1 in 4 patches is."""
    Hp, Wp = ((H + 3) & ~3) + 8, ((W + 3) & ~3) + 8
    stats = (C.c_longlong * 3)()
    bad = hl.hl_strips_layout(Hp, Wp, phases, stats)
    patches, aligned, one_line = stats[0], stats[1], stats[2]
    assert bad == 0 and patches == (Hp - 4) * (Wp - 4)
    assert aligned == patches                                       # every patch starts at a multiple of 32 * phases bytes
    if phases == 4:
        assert one_line == patches
    else:
        assert abs(one_line / patches - phases / 4.0) < 0.02        # 1 phase: rows = 0 mod 4 only; 2 phases: rows = 0, 1 mod 4


def test_strip_copy_rule(hl):
    """eds_layout.hpp eds_strips_decide: the copies are made for frames that are solved again (0 tiles, 1 copies current, 2 convert)."""
    d = hl.hl_strips_decide
    assert d(0, 0, 0, 4096) == 1                                    # everything current
    assert d(0, 4096, 4096, 4096) == 0                              # first solve on new frames: the tiles
    assert d(0, 4096, 0, 4096) == 2                                 # the same frames again: convert
    assert d(0, 372, 372, 4096) == 2 and d(0, 373, 373, 4096) == 0  # a few new frames among many: converted at once (< 1 in 11)
    assert d(0, 40, 3, 40) == 2 and d(0, 40, 4, 40) == 0
    assert d(1, 4096, 4096, 4096) == 2 and d(1, 0, 0, 8) == 1       # eager
    assert d(2, 0, 0, 8) == 0 and d(2, 8, 0, 8) == 0                # never

