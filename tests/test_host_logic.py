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
                                                         "slam-eds_amd/csrc/eds_solver.hpp", "slam-eds_amd/csrc/eds_layout.hpp",
                                                         "slam-eds_amd/csrc/eds_launch_rule.hpp")]
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



# ---- the launch rule (slam-eds_amd/csrc/eds_launch_rule.hpp): which kernel a solve launches, as a pure function -------------------
TEAM_OK, COOLDOWN, STRIPS, RETRY = 2, 4, 8, 1


def _lm6(hl, knobs="", maxN=2000, count=4096, bicubic=1, iters=10, lm6=1, huber=0, H=480, flags=TEAM_OK | STRIPS):
    out = np.zeros(15, dtype=np.int32)
    rc = hl.hl_lm6_rule(knobs.encode(), np.array([maxN, count, bicubic, iters, lm6, huber, H], dtype=np.int32).ctypes.data_as(_ip), int(flags),
                        out.ctypes.data_as(_ip))
    assert rc == 0
    keys = ("kind", "S", "P", "T", "Q", "K", "bilinear_tu", "wide_members", "threads", "ppt", "strips_eligible", "wants_team", "exists", "note_T", "G")
    return dict(zip(keys, (int(x) for x in out)))


def _ref12(hl, knobs="", maxN=2000, count=4096, bicubic=1, nc=0, H=480, flags=TEAM_OK | STRIPS):
    out = np.zeros(10, dtype=np.int32)
    rc = hl.hl_ref12_rule(knobs.encode(), np.array([maxN, count, bicubic, nc, H], dtype=np.int32).ctypes.data_as(_ip), int(flags), out.ctypes.data_as(_ip))
    assert rc == 0
    return dict(zip(("S", "T", "CAP", "NC", "K", "Q", "strips_eligible", "wants_team", "exists", "G"), (int(x) for x in out)))


def _k6(d):
    return (d["kind"], d["S"], d["P"], d["T"], d["Q"], d["K"])


def test_launch_rule_pose_only_table(hl):
    """DESIGN.md 3.7, row by row: the kernel eds_fused_solve launches is a pure function of the shape of the range, the handle's knobs,
    the teams' time-out policy and whether the strip copies of the frames are current."""
    FUSED, TEAM, STREAM = 0, 1, 2
    # the headline: 4 096 x 2 000 points, bicubic — strips when the copies are there, the tile kernel on a frame's first solve
    assert _k6(_lm6(hl)) == (FUSED, 0, 4, 512, 3, 1)
    assert _k6(_lm6(hl, flags=TEAM_OK)) == (FUSED, 0, 4, 512, 1, 1)
    assert _k6(_lm6(hl, huber=1)) == (FUSED, 0, 4, 512, 4, 1) and _k6(_lm6(hl, huber=1, flags=TEAM_OK)) == (FUSED, 0, 4, 512, 2, 1)
    assert _k6(_lm6(hl, "EDS_FUSED_LAYOUT=tiles")) == (FUSED, 0, 4, 512, 1, 1)
    # the bilinear sampler: its strip kernel on copies, its lane kernel (second translation unit) otherwise
    d = _lm6(hl, bicubic=0)
    assert _k6(d) == (FUSED, 1, 4, 512, 3, 1) and not d["bilinear_tu"]
    d = _lm6(hl, bicubic=0, flags=TEAM_OK)
    assert _k6(d) == (FUSED, 1, 4, 512, 0, 1) and d["bilinear_tu"] and d["note_T"] == 512
    # below 32 alignments without teams (GN6 forms none): the lane-per-patch gather
    assert _k6(_lm6(hl, count=8, lm6=0)) == (FUSED, 0, 4, 512, 0, 1)
    assert _k6(_lm6(hl, "EDS_FUSED_GATHER=lane")) == (FUSED, 0, 4, 512, 0, 1)
    # small keyframes: fewer threads, one point per lane
    d = _lm6(hl, maxN=256, count=4096)
    assert d["threads"] == 256 and _k6(d) == (FUSED, 0, 1, 512, 1, 1)
    d = _lm6(hl, maxN=700, count=40)
    assert d["threads"] == 512 and d["ppt"] == 2 and _k6(d) == (FUSED, 0, 2, 512, 3, 1)
    # the latency regime: 4 CUs up to 64 alignments of more than 1 024 points, 2 up to 128; never for 513 .. 1 024 points, GN6, a retry,
    # a running cool-down
    assert _k6(_lm6(hl, count=1)) == (TEAM, 0, 1, 512, 0, 4) and _k6(_lm6(hl, count=64)) == (TEAM, 0, 1, 512, 1, 4)
    assert _k6(_lm6(hl, count=65)) == (TEAM, 0, 2, 512, 3, 2) and _k6(_lm6(hl, count=128, flags=TEAM_OK)) == (TEAM, 0, 2, 512, 1, 2)
    assert _k6(_lm6(hl, count=129)) == (FUSED, 0, 4, 512, 3, 1)
    assert _lm6(hl, maxN=1024, count=1)["K"] == 1 and _lm6(hl, count=1, lm6=0)["K"] == 1 and _lm6(hl, count=1, iters=0)["K"] == 1
    assert _lm6(hl, count=1, flags=STRIPS)["K"] == 1 and _lm6(hl, count=1, flags=RETRY | TEAM_OK | STRIPS)["K"] == 1
    d = _lm6(hl, count=1, bicubic=0)
    assert _k6(d) == (TEAM, 1, 1, 512, 0, 4) and d["bilinear_tu"]
    # beyond 2 048 points: teams of 1 024 points per member at any batch size while at most one workgroup per CU, members of 2 048 beyond
    assert _k6(_lm6(hl, maxN=8000, count=1, H=720)) == (TEAM, 0, 2, 512, 0, 8)
    assert _k6(_lm6(hl, maxN=8000, count=32, H=720)) == (TEAM, 0, 2, 512, 3, 8)
    d = _lm6(hl, maxN=8000, count=256, H=720, huber=1)
    assert _k6(d) == (TEAM, 0, 4, 512, 4, 4) and d["wide_members"]
    assert _k6(_lm6(hl, maxN=8000, count=256, H=720, flags=TEAM_OK)) == (TEAM, 0, 4, 512, 1, 4)
    assert _k6(_lm6(hl, "EDS_TEAM_WIDE=0", maxN=8000, count=256, H=720)) == (TEAM, 0, 2, 512, 3, 8)
    assert _k6(_lm6(hl, maxN=16000, count=64)) == (TEAM, 0, 4, 512, 3, 8) and _k6(_lm6(hl, maxN=16000, count=16)) == (TEAM, 0, 2, 512, 3, 16)
    assert _k6(_lm6(hl, maxN=4000, count=64)) == (TEAM, 0, 2, 512, 3, 4) and _k6(_lm6(hl, maxN=4000, count=65)) == (TEAM, 0, 4, 512, 3, 2)
    d = _lm6(hl, maxN=8000, count=256, H=720, bicubic=0)          # the bilinear sampler: wide members on the strips only
    assert _k6(d) == (TEAM, 1, 4, 512, 3, 4)
    d = _lm6(hl, maxN=8000, count=256, H=720, bicubic=0, flags=TEAM_OK)
    assert _k6(d) == (TEAM, 1, 2, 512, 0, 8) and d["bilinear_tu"]
    # ... the streaming kernel where no team can form: GN6, more than 16 384 points, teams paused; a handful of very large ones resident
    assert _k6(_lm6(hl, maxN=8000, count=256, lm6=0)) == (STREAM, 0, 2048, 512, 0, 1)
    assert _k6(_lm6(hl, maxN=20000, count=64)) == (STREAM, 0, 2048, 512, 0, 1)
    assert _k6(_lm6(hl, maxN=8000, count=8, lm6=0)) == (FUSED, 0, 0, 1024, 0, 1)
    assert _k6(_lm6(hl, maxN=3000, count=8, lm6=0)) == (STREAM, 0, 2048, 512, 0, 1)
    assert _k6(_lm6(hl, "EDS_LM6_KERNEL=paired", lm6=0)) == (STREAM, 0, 1024, 256, 0, 1)
    # knobs that override the team size stay inside what is feasible and never reach past the time-out policy
    assert _lm6(hl, "EDS_LM6_TEAM=2", count=16)["K"] == 2 and _lm6(hl, "EDS_LM6_TEAM=1", count=16)["K"] == 1
    assert _lm6(hl, "EDS_LM6_TEAM=8", count=16)["K"] == 8 and _lm6(hl, "EDS_LM6_TEAM=2", maxN=4000, count=16)["K"] == 4
    assert _lm6(hl, "EDS_LM6_TEAM=4", count=16, flags=COOLDOWN | STRIPS)["K"] == 1
    assert _lm6(hl, "EDS_LM6_SPEC=0", count=16)["K"] == 1
    # candidate groups (round 5): G teams of four evaluate G prepared candidates at once — as many groups as fit HALF the CUs
    # (G x 4 x count <= 128), two while all workgroups of the launch still get a CU of their own; members of 512 points only; the
    # knob overrides, 1 switches them off
    assert [_lm6(hl, count=c)["G"] for c in (1, 4, 5, 8, 9, 16, 17, 32, 33, 64)] == [8, 8, 4, 4, 2, 2, 2, 2, 1, 1]
    assert _lm6(hl, "EDS_LM6_GROUPS=1", count=1)["G"] == 1 and _lm6(hl, "EDS_LM6_GROUPS=2", count=1)["G"] == 2 and _lm6(hl, "EDS_LM6_GROUPS=8", count=16)["G"] == 8
    assert _lm6(hl, count=1, bicubic=0)["G"] == 8 and _lm6(hl, count=32)["Q"] == 1 and _lm6(hl, count=32)["exists"]
    assert _lm6(hl, maxN=8000, count=1, H=720)["G"] == 1 and _lm6(hl, "EDS_LM6_TEAM=2", count=4)["G"] == 1 and _lm6(hl, count=129)["G"] == 1
    assert _lm6(hl, count=1, flags=RETRY | TEAM_OK | STRIPS)["G"] == 1 and _lm6(hl, count=4, flags=COOLDOWN | STRIPS)["G"] == 1


def test_launch_rule_force_knobs(hl):
    """EDS_FORCE_FUSED6 / EDS_FORCE_FUSED12 (test hooks of tests/test_instances_gpu.py): honoured where the instantiation exists and can solve
    the range, ignored (the rule stands) where not."""
    FUSED, TEAM = 0, 1
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,2,1024,1,1,1", maxN=2000, count=8)) == (FUSED, 0, 2, 1024, 1, 1)
    assert _lm6(hl, "EDS_FORCE_FUSED6=0,2,1024,1,1,1", maxN=2000, count=8)["threads"] == 1024
    d = _lm6(hl, "EDS_FORCE_FUSED6=0,2,512,3,16,1", maxN=16000, count=2)
    assert _k6(d) == (TEAM, 0, 2, 512, 3, 16) and d["G"] == 1 and d["strips_eligible"]
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,2,512,3,16,1", maxN=16000, count=2, flags=TEAM_OK)) != (TEAM, 0, 2, 512, 3, 16)       # no current copies: the rule stands
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,1,512,0,4,8", maxN=2000, count=2))[5] == 4 and _lm6(hl, "EDS_FORCE_FUSED6=0,1,512,0,4,8", maxN=2000, count=2)["G"] == 8
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,2,512,1,1,1", maxN=2000, count=8)) != (FUSED, 0, 2, 512, 1, 1)                       # 1 024 lane slots for 2 000 points
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=1,4,512,3,1,1", maxN=2000, count=8)) != (FUSED, 1, 4, 512, 3, 1)                       # the other sampler
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,4,512,4,1,1", maxN=2000, count=8)) != (FUSED, 0, 4, 512, 4, 1)                       # the Huber variant without a threshold
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,4,512,4,1,1", maxN=2000, count=8, huber=1)) == (FUSED, 0, 4, 512, 4, 1)
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,3,512,1,1,1", maxN=1000, count=8)) == _k6(_lm6(hl, maxN=1000, count=8))                # not compiled: ignored
    assert _k6(_lm6(hl, "EDS_FORCE_FUSED6=0,1,512,0,4,1", maxN=2000, count=2, lm6=0))[0] == FUSED                                # teams need the damped solver
    K = lambda d: (d["S"], d["T"], d["CAP"], d["NC"], d["K"], d["Q"])
    assert K(_ref12(hl, "EDS_FORCE_FUSED12=0,256,320,0,1,0", count=4)) == (0, 256, 320, 0, 1, 0)
    assert K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,1408,0,16,0", count=4)) == (0, 512, 1408, 0, 16, 0)
    assert K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,1408,0,4,2", count=4)) == (0, 512, 1408, 0, 4, 2) and K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,1408,0,4,2", count=4, flags=TEAM_OK)) != (0, 512, 1408, 0, 4, 2)
    assert K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,1408,1,1,1", count=4, nc=1)) == (0, 512, 1408, 1, 1, 1) and K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,1408,1,1,1", count=4)) != (0, 512, 1408, 1, 1, 1)
    assert K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,1408,0,8,0", count=4, flags=STRIPS)) != (0, 512, 1408, 0, 8, 0)                  # the time-out policy says no
    assert all(_ref12(hl, f"EDS_FORCE_FUSED12={f}", count=4)["exists"] for f in ("0,256,320,0,1,2", "1,512,1408,0,2,0", "0,512,1408,0,1,7"))


def test_launch_rule_ref12_table(hl):
    K = lambda d: (d["S"], d["T"], d["CAP"], d["NC"], d["K"], d["Q"])
    NB2 = 2 << 8                                                                                                    # (flags bits 8..15: residual blocks)
    assert K(_ref12(hl)) == (0, 256, 320, 0, 1, 2) and K(_ref12(hl, flags=TEAM_OK | NB2)) == (0, 256, 320, 0, 1, 1)       # the batch: two alignments per CU
    # round 6: ONE residual block of at most 2 000 points on NEW frames (tiles) — the paired shape with 736 cache slots per alignment; on
    # the strip copies only by knob; the full-cache one-per-CU shape only by knob (it loses: profiles/r06_ref12_shapes.txt)
    assert K(_ref12(hl, flags=TEAM_OK)) == (0, 256, 736, 0, 1, 1) and K(_ref12(hl, flags=TEAM_OK, maxN=2001)) == (0, 256, 320, 0, 1, 1)
    assert K(_ref12(hl, "EDS_REF12_KERNEL=paired", flags=TEAM_OK)) == (0, 256, 320, 0, 1, 1) and K(_ref12(hl, "EDS_REF12_KERNEL=half")) == (0, 256, 736, 0, 1, 2)
    assert K(_ref12(hl, "EDS_REF12_KERNEL=full", flags=TEAM_OK)) == (0, 512, 2000, 0, 1, 1) and K(_ref12(hl, "EDS_REF12_KERNEL=full")) == (0, 512, 2000, 0, 1, 2)
    assert K(_ref12(hl, "EDS_REF12_KERNEL=full", flags=TEAM_OK | NB2)) == (0, 512, 1408, 0, 1, 1) and K(_ref12(hl, "EDS_REF12_KERNEL=full", nc=1))[2] == 1408
    assert K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,2000,0,1,1", count=300, flags=TEAM_OK)) == (0, 512, 2000, 0, 1, 1)
    assert K(_ref12(hl, "EDS_FORCE_FUSED12=0,512,2000,0,1,1", count=300, flags=TEAM_OK | NB2)) != (0, 512, 2000, 0, 1, 1)       # two blocks: refused
    assert K(_ref12(hl, count=512, flags=TEAM_OK)) == (0, 256, 320, 0, 1, 0)                                       # tiles: the quad gather from 1 024 on
    assert K(_ref12(hl, count=256)) == (0, 512, 1408, 0, 1, 2) and K(_ref12(hl, count=65, flags=TEAM_OK)) == (0, 512, 1408, 0, 1, 0)
    # (round 5: where candidate groups are formed a member's patch cache is 512 points; EDS_REF12_GROUPS=1 is the one-team launch)
    assert K(_ref12(hl, count=1)) == (0, 512, 512, 0, 8, 0) and K(_ref12(hl, count=8)) == (0, 512, 512, 0, 8, 0) and K(_ref12(hl, count=16)) == (0, 512, 512, 0, 4, 0)
    assert K(_ref12(hl, "EDS_REF12_GROUPS=1", count=1)) == (0, 512, 1408, 0, 8, 0) and K(_ref12(hl, "EDS_REF12_GROUPS=1", count=16)) == (0, 512, 1408, 0, 8, 0)
    assert K(_ref12(hl, maxN=3000, count=16)) == (0, 512, 512, 0, 8, 0) and _ref12(hl, maxN=3000, count=16)["G"] == 2       # members of 375 points: teams of 8 stay
    assert K(_ref12(hl, count=17)) == (0, 512, 512, 0, 4, 0) and K(_ref12(hl, count=33)) == (0, 512, 1408, 0, 4, 0) and K(_ref12(hl, count=64)) == (0, 512, 1408, 0, 4, 2)
    assert K(_ref12(hl, maxN=1000, count=8)) == (0, 512, 1408, 0, 2, 0) and K(_ref12(hl, maxN=500, count=8))[4] == 1
    assert K(_ref12(hl, maxN=16000, count=4)) == (0, 512, 1408, 0, 16, 0) and K(_ref12(hl, maxN=8000, count=64)) == (0, 512, 1408, 0, 8, 0)
    # teams of 8 / 16 have no strip instantiation: their frames are not converted
    assert not _ref12(hl, "EDS_REF12_TEAM=8", count=64)["strips_eligible"] and _ref12(hl, count=64)["strips_eligible"]
    # the NC residual: no teams, no strips; the bilinear sampler: the lane gather
    assert K(_ref12(hl, nc=1)) == (0, 256, 320, 1, 1, 1) and K(_ref12(hl, nc=1, count=8)) == (0, 512, 1408, 1, 1, 0)
    assert K(_ref12(hl, bicubic=0)) == (1, 256, 320, 0, 1, 0) and K(_ref12(hl, bicubic=0, count=4)) == (1, 512, 512, 0, 8, 0)
    assert K(_ref12(hl, "EDS_REF12_KERNEL=wide")) == (0, 512, 1408, 0, 1, 2) and K(_ref12(hl, "EDS_REF12_KERNEL=paired", count=8)) == (0, 256, 320, 0, 1, 0)
    assert K(_ref12(hl, "EDS_REF12_TEAM=4", count=8, flags=COOLDOWN))[4] == 1 and K(_ref12(hl, count=8, flags=RETRY | TEAM_OK))[4] == 1
    # candidate groups (round 5): as many teams as give every workgroup a CU of its own; teams of 8 and 4 (lane gather) only
    assert [_ref12(hl, count=c)["G"] for c in (1, 8, 9, 16, 17, 32, 33, 64, 65)] == [4, 4, 4, 4, 2, 2, 1, 1, 1]
    assert [_ref12(hl, count=c)["K"] for c in (1, 8, 9, 16, 17, 32, 33)] == [8, 8, 4, 4, 4, 4, 4]
    assert _ref12(hl, "EDS_REF12_GROUPS=1", count=1)["G"] == 1 and _ref12(hl, "EDS_REF12_GROUPS=2", count=1)["G"] == 2 and _ref12(hl, "EDS_REF12_GROUPS=4", count=40)["G"] == 1
    assert _ref12(hl, maxN=4097, count=1)["G"] == 1 and _ref12(hl, maxN=4096, count=1)["G"] == 4 and _ref12(hl, count=1, bicubic=0)["G"] == 4
    assert _ref12(hl, count=1)["CAP"] == 512 and _ref12(hl, "EDS_REF12_GROUPS=1", count=1)["CAP"] == 1408 and _ref12(hl, maxN=2049, count=20)["G"] == 1
    assert _ref12(hl, count=1, nc=1)["G"] == 1 and _ref12(hl, count=4, flags=COOLDOWN | STRIPS)["G"] == 1 and _ref12(hl, maxN=1000, count=8)["G"] == 1


def test_launch_rule_counts_the_devices_own_cus(hl):
    """the candidate-group rule formed groups for a 256-CU part whatever the device; the handle now carries its device's CU
    count (hipDeviceProp_t.multiProcessorCount at eds_trk_create) and the rule never asks for more workgroups than that."""
    assert _ref12(hl, count=8)["G"] == 4 and _ref12(hl, "cus=256", count=8)["G"] == 4
    assert _ref12(hl, "cus=128", count=8)["G"] == 2 and _ref12(hl, "cus=64", count=8)["G"] == 1
    for cus in (64, 128, 256):
        for c in (1, 2, 8, 16, 32):
            d = _ref12(hl, f"cus={cus}", count=c)
            assert d["G"] == 1 or c * d["K"] * d["G"] <= cus, (cus, c, d)
            e = _lm6(hl, f"cus={cus}", count=c)
            assert e["G"] == 1 or c * e["K"] * e["G"] <= cus, (cus, c, e)
    assert _lm6(hl, count=1)["G"] == 8 and _lm6(hl, "cus=32", count=1)["G"] == 4


def test_launch_rule_never_leaves_the_instantiations_the_library_holds(hl):
    """Closure: over a sweep of shapes, samplers, policies and knob settings the rule always names a kernel that was compiled."""
    rng = np.random.default_rng(7)
    knobs = ["", "EDS_FUSED_LAYOUT=tiles", "EDS_FUSED_GATHER=lane", "EDS_FUSED_GATHER=quad", "EDS_LM6_TEAM=1", "EDS_LM6_TEAM=2", "EDS_LM6_TEAM=4",
             "EDS_LM6_TEAM=8", "EDS_LM6_TEAM=16", "EDS_TEAM_WIDE=1", "EDS_TEAM_WIDE=0", "EDS_LM6_KERNEL=resident", "EDS_LM6_KERNEL=wide",
             "EDS_LM6_KERNEL=paired", "EDS_FUSED_THREADS=256", "EDS_FUSED_THREADS=1024", "EDS_FUSED_PPT=2", "EDS_FUSED_PPT=3", "EDS_FUSED_PPT=0",
             "EDS_LM6_SPEC=0", "EDS_FUSED_THREADS=1024;EDS_FUSED_PPT=2;EDS_LM6_KERNEL=resident", "EDS_LM6_GROUPS=1", "EDS_LM6_GROUPS=2",
             "EDS_LM6_GROUPS=4", "EDS_LM6_GROUPS=8", "EDS_LM6_GROUPS=8;EDS_LM6_TEAM=4"]
    n = 0
    for kn in knobs:
        for _ in range(300):
            maxN = int(rng.choice([1, 63, 64, 200, 512, 513, 1024, 1025, 2000, 2048, 2049, 4000, 4096, 4097, 8000, 8192, 8722, 16000, 16384, 16385, 30000]))
            count = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 4096, 5000]))
            d = _lm6(hl, kn, maxN=maxN, count=count, bicubic=int(rng.integers(2)), iters=int(rng.choice([0, 1, 10])), lm6=int(rng.integers(2)),
                     huber=int(rng.integers(2)), H=int(rng.choice([120, 480, 720, 9000])), flags=int(rng.integers(16)))
            assert d["exists"], (kn, maxN, count, d)
            assert d["kind"] == 2 or d["K"] == 1 or d["P"] * 512 * d["K"] >= min(maxN, 16384) or d["K"] in (2, 4, 8, 16), d
            if d["kind"] != 2 and d["P"] > 0 and d["kind"] == 0:
                assert d["P"] * d["threads"] >= maxN, (kn, maxN, count, d)       # register-resident points: every point has a lane slot
            n += 1
    for kn in ["", "EDS_FUSED_LAYOUT=tiles", "EDS_FUSED_GATHER=lane", "EDS_FUSED_GATHER=quad", "EDS_REF12_TEAM=1", "EDS_REF12_TEAM=2", "EDS_REF12_TEAM=4",
               "EDS_REF12_TEAM=8", "EDS_REF12_TEAM=16", "EDS_REF12_KERNEL=wide", "EDS_REF12_KERNEL=paired", "EDS_REF12_GROUPS=1", "EDS_REF12_GROUPS=2",
               "EDS_REF12_GROUPS=4", "EDS_REF12_GROUPS=4;EDS_REF12_TEAM=8"]:
        for _ in range(300):
            d = _ref12(hl, kn, maxN=int(rng.choice([64, 512, 513, 1024, 1025, 2000, 4096, 4097, 8192, 8193, 16000])),
                       count=int(rng.choice([1, 16, 17, 32, 33, 64, 65, 256, 257, 1023, 1024, 4096])), bicubic=int(rng.integers(2)), nc=int(rng.integers(2)),
                       H=int(rng.choice([480, 9000])), flags=int(rng.integers(16)))
            assert d["exists"], (kn, d)
            assert not (d["NC"] and d["K"] > 1) and not (d["Q"] == 2 and d["NC"])
    assert n == len(knobs) * 300


def test_knob_names_and_strip_budget(hl):
    # every knob with a value it takes and one it does not: invalid values are REFUSED (-2) instead of coerced, unknown names -1
    table = {"EDS_REF12_EXEC": ("device", "gpu"), "EDS_FUSED_THREADS": ("256", "100"), "EDS_FUSED_PPT": ("2", "-1"), "EDS_LM6_SPEC": ("0", "yes"),
             "EDS_LM6_KERNEL": ("paired", "fast"), "EDS_FUSED_LAYOUT": ("tiles", "rows"), "EDS_TEAM_TEST_DROP_MEMBER": ("1", "2"), "EDS_LM6_TEAM": ("4", "3"),
             "EDS_TEAM_WIDE": ("0", "wide"), "EDS_FUSED_GATHER": ("lane", "1"), "EDS_FUSED_REPORT": ("1", "on"), "EDS_REF12_KERNEL": ("wide", "1"),
             "EDS_REF12_TEAM": ("8", "5"), "EDS_STRIPS_PHASES": ("2", "3"), "EDS_STRIPS_POLICY": ("never", "always"), "EDS_STRIPS_BUDGET_PCT": ("30", "0"),
             "EDS_NO_SPIN": ("1", "x"), "EDS_POLL_RESULTS": ("0", "2"), "EDS_UPLOAD": ("bands", "1"), "EDS_FRAME_LAYOUT": ("rowmajor", "1"), "EDS_REDUCE_PPL": ("8", "abc"),
             "EDS_LM6_GROUPS": ("4", "3"), "EDS_UPLOAD_THREADS": ("6", "0"), "EDS_UPLOAD_DMA": ("1", "2"), "EDS_UPLOAD_STREAMS": ("1", "3"),
             "EDS_FORCE_FUSED6": ("0,4,512,3,1,1", "0,4,512"), "EDS_FORCE_FUSED12": ("0,256,320,0,1,2", "a,b"), "EDS_REF12_GROUPS": ("2", "3")}
    for name, (good, bad) in table.items():
        assert hl.hl_knob_set(name.encode(), good.encode()) == 0, name
        assert hl.hl_knob_set(name.encode(), bad.encode()) == -2, name
        assert hl.hl_knob_set(name.encode(), None) == 0 and hl.hl_knob_set(name.encode(), b"") == 0
    names = open(os.path.join(HERE, "..", "slam-eds_amd", "csrc", "eds_launch_rule.hpp")).read().split("#define EDS_KNOB_NAMES(X)")[1].split("\n\n")[0]
    assert sorted(table) == sorted(set(__import__("re").findall(r'X\("(EDS_[A-Z0-9_]+)"\)', names)))      # the table above covers every knob the library has
    assert hl.hl_knob_set(b"EDS_NO_SUCH_KNOB", b"1") == -1
    # the environment at eds_trk_create: a value a knob refuses is REPORTED (the create fails with EDS_ERR_INVALID), not skipped
    hl.hl_knobs_from_env.restype = C.c_char_p
    saved = {k: os.environ.pop(k, None) for k in table}
    try:
        assert hl.hl_knobs_from_env() is None
        os.environ["EDS_LM6_TEAM"] = "4"; os.environ["EDS_FUSED_REPORT"] = "1"
        assert hl.hl_knobs_from_env() is None
        os.environ["EDS_NO_SPIN"] = "yes"
        assert hl.hl_knobs_from_env() == b"EDS_NO_SPIN"
        os.environ["EDS_NO_SPIN"] = "1"; os.environ["EDS_LM6_TEAM"] = "0"
        assert hl.hl_knobs_from_env() == b"EDS_LM6_TEAM"
    finally:
        for k in table:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    f = hl.hl_strips_phases_for_budget
    f.argtypes = [C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int]
    two = 2 * 1264128                                       # two column copies of one 640x480 frame (488 x 648 floats)
    GB = 1 << 30
    assert f(4, 4096, two, 240 * GB, 50) == 4               # the bench's batch: 41 GB of 120 GB allowed
    assert f(4, 4096, two, 60 * GB, 50) == 2 and f(4, 4096, two, 30 * GB, 50) == 1 and f(4, 4096, two, 10 * GB, 50) == 0
    assert f(4, 4096, two, 60 * GB, 95) == 4 and f(1, 1, two, 1 * GB, 50) == 1 and f(4, 100000, two, 250 * GB, 50) == 0
