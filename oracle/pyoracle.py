"""ctypes front-end of the C++ CPU oracle (oracle/libeds_oracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/eds_oracle.hpp header; parity unpinned).
Importable from tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke()
— never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libeds_oracle.so")

BICUBIC, BILINEAR = 0, 1
LOSS_NONE, LOSS_HUBER, LOSS_CAUCHY = 0, 1, 2
LP_CONSTANT, LP_MAD, LP_STD = 0, 1, 2
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2

_dp = C.POINTER(C.c_double)


class _Problem(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("_pad", C.c_int32),
                ("grad", _dp), ("norm_coord", _dp), ("idp", _dp), ("weights", _dp), ("frame", _dp),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


class _Config(C.Structure):
    _fields_ = [("sampling", C.c_int32), ("nc", C.c_int32), ("num_blocks", C.c_int32), ("loss_type", C.c_int32),
                ("loss_param", C.c_double), ("max_num_iterations", C.c_int32), ("eval_threads", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double)]


class _Summary(C.Structure):
    _fields_ = [("termination", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("num_residuals", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("seconds", C.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (a few seconds).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("eds_oracle_capi.cpp", "eds_oracle.hpp", "eds_cpu_fast.hpp")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libeds_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.eds_oracle_loss_param.restype = C.c_double
        _lib.eds_oracle_se3_distance.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Oracle:
    """One alignment's inputs bound to the C++ oracle."""

    def __init__(self, al, *, sampling=BICUBIC, nc=False, num_blocks=1, loss_type=LOSS_NONE, loss_param=1.0,
                 max_num_iterations=10, eval_threads=1, function_tolerance=1e-6, gradient_tolerance=1e-8,
                 parameter_tolerance=1e-6):
        self._keep = [_f64(al.grad), _f64(al.norm_coord), _f64(al.idp), _f64(al.weights), _f64(al.frame)]
        g, nc_, idp, w, fr = self._keep
        self.N = int(idp.shape[0])
        self.pb = _Problem(self.N, int(al.H), int(al.W), 0, _p(g), _p(nc_), _p(idp), _p(w), _p(fr),
                           float(al.fx), float(al.fy), float(al.cx), float(al.cy))
        self.cfg = _Config(int(sampling), int(bool(nc)), int(num_blocks), int(loss_type), float(loss_param),
                           int(max_num_iterations), int(eval_threads), float(function_tolerance),
                           float(gradient_tolerance), float(parameter_tolerance))

    def eval12(self, p, q, v, jac=True):
        N = self.N
        p, q, v = _f64(p), _f64(q), _f64(v)
        out = dict(r_raw=np.zeros(N), r=np.zeros(N), cost=C.c_double(0.0))
        if jac:
            out.update(J_global=np.zeros((N, 13)), J_local_raw=np.zeros((N, 12)), J_local=np.zeros((N, 12)),
                       gradient=np.zeros(12))
        rc = lib().eds_oracle_eval12(C.byref(self.pb), C.byref(self.cfg), _p(p), _p(q), _p(v), _p(out["r_raw"]),
                                     _p(out["r"]), _p(out.get("J_global")), _p(out.get("J_local_raw")),
                                     _p(out.get("J_local")), C.byref(out["cost"]), _p(out.get("gradient")))
        out["cost"] = out["cost"].value
        out["ok"] = rc == 0
        return out

    def solve_lm(self, p, q, v):
        p, q, v = _f64(p).copy(), _f64(q).copy(), _f64(v).copy()
        s = _Summary()
        rc = lib().eds_oracle_solve_lm(C.byref(self.pb), C.byref(self.cfg), _p(p), _p(q), _p(v), C.byref(s))
        return dict(p=p, q=q, v=v, usable=rc == 0, termination=s.termination,
                    num_successful_steps=s.num_successful_steps, num_unsuccessful_steps=s.num_unsuccessful_steps,
                    num_iterations=s.num_successful_steps + s.num_unsuccessful_steps, num_residuals=s.num_residuals,
                    initial_cost=s.initial_cost, final_cost=s.final_cost, seconds=s.seconds)

    def pose6_eval(self, p, q, v, huber_tau=0.0):
        N = self.N
        p, q, v = _f64(p), _f64(q), _f64(v)
        r, J, hw, H, b = np.zeros(N), np.zeros((N, 6)), np.zeros(N), np.zeros((6, 6)), np.zeros(6)
        cost = C.c_double(0.0)
        lib().eds_oracle_pose6_eval(C.byref(self.pb), C.byref(self.cfg), _p(p), _p(q), _p(v), C.c_double(huber_tau),
                                    _p(r), _p(J), _p(hw), _p(H), _p(b), C.byref(cost))
        return dict(r=r, J=J, hw=hw, H=H, b=b, cost=cost.value)

    def pose6_gn(self, p, q, v, iters=10, huber_tau=0.0):
        p, q, v = _f64(p).copy(), _f64(q).copy(), _f64(v)
        inc, costs = np.zeros((iters, 6)), np.zeros(iters)
        sec = C.c_double(0.0)
        n = lib().eds_oracle_pose6_gn(C.byref(self.pb), C.byref(self.cfg), _p(p), _p(q), _p(v), C.c_double(huber_tau),
                                      int(iters), _p(inc), _p(costs), C.byref(sec))
        return dict(p=p, q=q, iterations=n, increments=inc[:n], costs=costs[:n], seconds=sec.value)

    def pose6_lm(self, p, q, v, iters=10, lambda0=0.01, huber_tau=0.0):
        p, q, v = _f64(p).copy(), _f64(q).copy(), _f64(v)
        inc, costs, acc = np.zeros((iters, 6)), np.zeros(iters), np.zeros(iters, dtype=np.int32)
        sec, c0 = C.c_double(0.0), C.c_double(0.0)
        n = lib().eds_oracle_pose6_lm(C.byref(self.pb), C.byref(self.cfg), _p(p), _p(q), _p(v), C.c_double(huber_tau),
                                      int(iters), C.c_double(lambda0), _p(inc), _p(costs),
                                      acc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(c0), C.byref(sec))
        return dict(p=p, q=q, iterations=n, increments=inc[:n], costs=costs[:n], accepted=acc[:n],
                    initial_cost=c0.value, seconds=sec.value)


class FastLM6:
    """The optimised CPU variant (oracle/eds_cpu_fast.hpp) of Oracle.pose6_lm for a fixed velocity: a baseline, not a checker.
    Construction converts the inputs once (the counterpart of set_keyframe / set_event_frame); solve() is what gets timed."""

    def __init__(self, oracle: "Oracle", v):
        self._o = oracle
        lib().eds_oracle_fast_prepare.restype = C.c_void_p
        self._h = C.c_void_p(lib().eds_oracle_fast_prepare(C.byref(oracle.pb), _p(_f64(v))))

    def solve(self, p, q, iters=10, lambda0=0.01):
        p, q = _f64(p).copy(), _f64(q).copy()
        acc = np.zeros(iters, dtype=np.int32)
        sec = C.c_double(0.0)
        n = lib().eds_oracle_fast_lm6(self._h, _p(p), _p(q), int(iters), C.c_double(lambda0),
                                      acc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(sec))
        return dict(p=p, q=q, iterations=n, accepted=acc[:n], seconds=sec.value)

    def solve_scalar(self, p, q, iters=10, lambda0=0.01):
        """The scalar point loop (the checker of the vectorised one)."""
        p, q = _f64(p).copy(), _f64(q).copy()
        acc = np.zeros(iters, dtype=np.int32)
        n = lib().eds_oracle_fast_lm6_scalar(self._h, _p(p), _p(q), int(iters), C.c_double(lambda0), acc.ctypes.data_as(C.POINTER(C.c_int32)))
        return dict(p=p, q=q, iterations=n, accepted=acc[:n])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().eds_oracle_fast_free(self._h)
            self._h = None


def fast_is_vectorised() -> bool:
    return bool(lib().eds_oracle_fast_is_vectorised())


def bench_lm6(oracles, starts, iters=10, lambda0=0.01, threads=1, budget_s=1.0, fast=None):
    """All-core throughput of the pose-only solve, driven from C (eds_oracle_bench_lm6: persistent std::threads, no interpreter in the
    loop).  oracles: Oracle objects; starts: (p0, q0, v0) per oracle; fast: FastLM6 handles of the same alignments (the optimised
    variant) or None (the oracle's autodiff solve).  Returns dict(solves, iterations, seconds, iterations_per_s)."""
    n = len(oracles)
    probs = (_Problem * n)(*[o.pb for o in oracles])
    p0 = np.ascontiguousarray(np.stack([_f64(s[0]) for s in starts])); q0 = np.ascontiguousarray(np.stack([_f64(s[1]) for s in starts]))
    v0 = np.ascontiguousarray(np.stack([_f64(s[2]) for s in starts]))
    handles = (C.c_void_p * n)(*[f._h for f in fast]) if fast is not None else None
    its, el = C.c_longlong(0), C.c_double(0.0)
    L = lib()
    L.eds_oracle_bench_lm6.restype = C.c_longlong
    done = L.eds_oracle_bench_lm6(n, probs, handles, C.byref(oracles[0].cfg), _p(p0), _p(q0), _p(v0), int(iters), C.c_double(lambda0), int(threads),
                                  C.c_double(budget_s), 1 if fast is not None else 0, C.byref(its), C.byref(el))
    return dict(solves=int(done), iterations=int(its.value), seconds=el.value, iterations_per_s=its.value / max(el.value, 1e-9))


def loss_param(residuals, method, current=0.0):
    """Returns (tau, reordered residuals) — the reference reorders kf->residuals in place."""
    r = _f64(residuals).copy()
    tau = lib().eds_oracle_loss_param(_p(r), int(r.shape[0]), int(method), C.c_double(current))
    return tau, r


def bicubic(frame, row, col):
    fr = _f64(frame)
    out = np.zeros(3)
    lib().eds_oracle_bicubic(_p(fr), int(fr.shape[0]), int(fr.shape[1]), C.c_double(row), C.c_double(col), _p(out))
    return out   # f, df/drow, df/dcol


def bilinear(frame, row, col):
    fr = _f64(frame)
    out = np.zeros(3)
    lib().eds_oracle_bilinear(_p(fr), int(fr.shape[0]), int(fr.shape[1]), C.c_double(row), C.c_double(col), _p(out))
    return out


def se3_exp(xi):
    xi = _f64(xi); t, q = np.zeros(3), np.zeros(4)
    lib().eds_oracle_se3_exp(_p(xi), _p(t), _p(q))
    return t, q


def se3_log(t, q):
    t, q = _f64(t), _f64(q); xi = np.zeros(6)
    lib().eds_oracle_se3_log(_p(t), _p(q), _p(xi))
    return xi


def se3_left_update(xi, t, q):
    xi, t, q = _f64(xi), _f64(t).copy(), _f64(q).copy()
    lib().eds_oracle_se3_left_update(_p(xi), _p(t), _p(q))
    return t, q


def se3_distance(ta, qa, tb, qb):
    ta, qa, tb, qb = _f64(ta), _f64(qa), _f64(tb), _f64(qb)
    return lib().eds_oracle_se3_distance(_p(ta), _p(qa), _p(tb), _p(qb))


def state_plus(x13, d12):
    x13, d12 = _f64(x13), _f64(d12); out = np.zeros(13)
    lib().eds_oracle_state_plus(_p(x13), _p(d12), _p(out))
    return out


def quat_to_R(q):
    q = _f64(q); R = np.zeros((3, 3))
    lib().eds_oracle_quat_to_R(_p(q), _p(R))
    return R


def quat_plus_jacobian(q):
    q = _f64(q); J = np.zeros((4, 3))
    lib().eds_oracle_quat_plus_jacobian(_p(q), _p(J))
    return J


def unit_plus_jacobian(v):
    v = _f64(v); J = np.zeros((6, 6))
    lib().eds_oracle_unit_plus_jacobian(_p(v), _p(J))
    return J


def loss_eval(loss_type, a, s):
    rho = np.zeros(3)
    lib().eds_oracle_loss_eval(int(loss_type), C.c_double(a), C.c_double(s), _p(rho))
    return rho
