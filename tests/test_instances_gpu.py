"""EVERY compiled instantiation of the persistent kernels against the CPU oracle (Next #6).

The library exports its X-macro lists (eds_trk_kernel_instances: the launchers dispatch over exactly these), and the knobs
EDS_FORCE_FUSED6 / EDS_FORCE_FUSED12 launch one named instantiation wherever it can solve the range (csrc/eds_launch_rule.hpp).  The
test walks both lists: for every instantiation a small batch whose shape it can solve (point count for its lane slots, sampler, Huber
variant, strip copies, team size), the launch checked by name (eds_trk_last_launch), the result by the oracle — accept pattern /
iteration counts equal, pose within 1e-6 (SURVEY 8c asks 1e-4), kept residuals within 1e-5 relative (Tracker.cpp:223-230) — so nothing
that is compiled is reachable only by a knob nobody tests.  The soak of tools/fuzz_*.py runs here too, at reduced counts."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 240, 320
PS, = (np.array([1e-3, -2e-3, 5e-4]),)          # a generic start: at the identity the bilinear gradient is discontinuous (pixel centres)
_cache = {}


def _alignment(synth, seed, N):
    key = (seed, N)
    if key not in _cache:
        a = synth.make_alignment(seed, H=H, W=W, N=N)
        _cache[key] = (a, np.ascontiguousarray(a.frame, dtype=np.float32))
    return _cache[key]


def _rounded(synth, a, f32):
    return synth.Alignment(**{**a.__dict__, "frame": f32.astype(np.float64)})


def test_the_library_reports_its_instantiations(gpu, capi):
    f6, f12 = capi.kernel_instances(0), capi.kernel_instances(1)
    assert len(f6) >= 90 and len(f12) >= 20 and len(set(f6)) == len(f6) and len(set(f12)) == len(f12)
    assert (0, 4, 512, 1, 1, 1) in f6 and (0, 4, 512, 3, 1, 1) in f6 and (0, 1, 512, 0, 4, 8) in f6 and (0, 256, 320, 0, 1, 1) in f12
    assert capi.lib().eds_trk_kernel_instances(3, -1, None) == -1
    print(f"\n[instances] eds_fused6_kernel: {len(f6)}, eds_fused12_kernel: {len(f12)}")


def _fused6_cases(capi):
    return capi.kernel_instances(0)


def test_every_fused6_instantiation_vs_oracle(gpu, capi, synth, po):
    qs = synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    checked = 0
    for (S, P, T, Q, K, G) in capi.kernel_instances(0):
        huber = Q in (2, 4)
        cap = P * (512 if K > 1 else T) * K if P > 0 else 2500
        N = cap - 37 if K > 1 else min(cap - 13, 2000)
        B = 3
        tau = 0.004 if huber else 0.0
        iters = 6
        cfg = capi.default_config(sampling=S, solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=iters, huber_tau=tau)
        h = capi.Handle(cfg, B, N, H, W)
        als = [_alignment(synth, 7000 + b, N) for b in range(B)]
        for b, (a, f32) in enumerate(als):
            h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
            h.set_event_frame(b, f32)
        if Q >= 3:
            h.prepare_frames(0, B)                       # the strip instantiations read the copies
        h.set_knob("EDS_FORCE_FUSED6", f"{S},{P},{T},{Q},{K},{G}")
        h.set_states(0, np.stack([PS] * B), np.stack([qs] * B), np.stack([a.v0 for a, _ in als]))
        h.optimize_batch(0, 0, B)
        li = h.last_launch()
        want = f"eds_fused6_kernel<{S}, {P}, {T}, {Q}, {K}" + (f", {G}>" if G > 1 else ">")
        assert li["kernel"] == want, (li["kernel"], want)
        assert h.info(0)["flags"] == 0, "team time-out"
        tab = h.results(0, B)
        for b, (a, f32) in enumerate(als):
            o = po.Oracle(_rounded(synth, a, f32), sampling=po.BICUBIC if S == 0 else po.BILINEAR)
            ref = o.pose6_lm(PS, qs, a.v0, iters=iters, lambda0=cfg.lambda0, huber_tau=tau)
            assert tab[b, 15] == 1.0 and tab[b, 14] == ref["iterations"], (want, b)
            assert np.array_equal(h.trace(b)["accepted"], ref["accepted"]), (want, b, h.trace(b)["accepted"], ref["accepted"])
            # (the bilinear sampler's gradient is discontinuous across pixel borders: with thousands of points a few sit within fp32 round-off of
            # one, and the fp64 oracle takes the other side — SURVEY 8c's 1e-4 for it, 1e-6 for the reference's bicubic sampler)
            assert po.se3_distance(tab[b, 0:3], tab[b, 3:7], ref["p"], ref["q"]) <= (1e-6 if S == 0 else 1e-4), (want, b)
            if b == 0 and S == 0:
                er = o.pose6_eval(tab[b, 0:3], tab[b, 3:7], a.v0)["r"]
                assert np.abs(h.residuals(b) - er).max() <= 1e-5 * np.abs(er).max(), want
        h.close()
        checked += 1
    print(f"\n[instances] {checked} eds_fused6_kernel instantiations launched by name and checked against the oracle")
    assert checked == len(capi.kernel_instances(0))


def test_every_fused12_instantiation_vs_oracle(gpu, capi, synth, po):
    qs = synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    checked = 0
    # family 1: the one-team instantiations; family 2 (round 5): the candidate-group ones {S, T, NC, K, Q, G}, reached by forcing the one-team
    # instantiation they extend and asking for G groups (their patch cache is 512 points: a member's slice must fit)
    cases = [(S, T, CAP, NC, K, Q, 1) for (S, T, CAP, NC, K, Q) in capi.kernel_instances(1)] + [(S, T, 1408, NC, K, Q, G) for (S, T, NC, K, Q, G) in capi.kernel_instances(2)]
    assert len(capi.kernel_instances(2)) >= 6
    for (S, T, CAP, NC, K, Q, G) in cases:
        N = (2000 if K <= 4 else (4000 if K == 8 else 9000)) if G == 1 else 500 * K - 11
        B = 3
        nblk = 1 if CAP in (2000, 736) else 2            # (the slim shapes of round 6 hold the sums of ONE residual block)
        cfg = capi.default_config(sampling=S, solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=8, num_blocks=nblk, nc=NC,
                                  loss_type=capi.LOSS_HUBER, loss_param=0.3)
        h = capi.Handle(cfg, B, N, H, W)
        als = [_alignment(synth, 7100 + b, N) for b in range(B)]
        for b, (a, f32) in enumerate(als):
            h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
            h.set_event_frame(b, f32)
        if Q == 2:
            h.prepare_frames(0, B)
        h.set_knob("EDS_FORCE_FUSED12", f"{S},{T},{CAP},{NC},{K},{Q}")
        h.set_knob("EDS_REF12_GROUPS", str(G))
        h.set_states(0, np.stack([PS] * B), np.stack([qs] * B), np.stack([a.v0 for a, _ in als]))
        h.optimize_batch(0, 0, B)
        li = h.last_launch()
        want = f"eds_fused12_kernel<{S}, {T}, {CAP if G == 1 else 512}, {'true' if NC else 'false'}, {K}, {Q}" + (f", {G}>" if G > 1 else ">")
        assert li["kernel"] == want, (li["kernel"], want)
        assert h.info(0)["flags"] == 0, "team time-out"
        tab = h.results(0, B)
        for b, (a, f32) in enumerate(als):
            o = po.Oracle(_rounded(synth, a, f32), sampling=po.BICUBIC if S == 0 else po.BILINEAR, nc=bool(NC), num_blocks=nblk, loss_type=po.LOSS_HUBER,
                          loss_param=0.3, max_num_iterations=8)
            ref = o.solve_lm(PS, qs, a.v0)
            info = h.info(b)
            assert info["success"] and (info["num_successful_steps"], info["num_unsuccessful_steps"]) == (ref["num_successful_steps"], ref["num_unsuccessful_steps"]), (want, b)
            assert info["termination"] == ref["termination"], (want, b)
            if S == 1 and NC:
                # bilinear sampler AND the NC residual (sampled brightness normalised per block, PhotometricErrorNC.hpp:151-186 — a functor the
                # reference ships but does not call): every point that sits within fp32 round-off of a pixel border moves the block norm too, and
                # the weakly determined velocity drifts by 1e-3 over 8 steps.  Checked by FUNCTION value instead of trajectory: the residuals the
                # kernel kept are the oracle's at the kernel's own solution, and the step accounting above is equal.
                er = o.eval12(tab[b, 0:3], tab[b, 3:7], tab[b, 7:13], jac=False)["r_raw"]
                assert np.abs(h.residuals(b) - er).max() <= 1e-4 * np.abs(er).max(), (want, b)
            else:
                tol = 1e-6 if S == 0 else 1e-4
                assert po.se3_distance(tab[b, 0:3], tab[b, 3:7], ref["p"], ref["q"]) <= tol and np.abs(tab[b, 7:13] - ref["v"]).max() <= tol, (want, b)
        h.close()
        checked += 1
    print(f"\n[instances] {checked} eds_fused12_kernel instantiations launched by name and checked against the oracle")
    assert checked == len(capi.kernel_instances(1)) + len(capi.kernel_instances(2))


@pytest.mark.parametrize("script,args", [("fuzz_parity.py", ["40", "901"]), ("fuzz_batch.py", ["12", "902"]), ("fuzz_rows.py", ["40", "903"]),
                                         ("fuzz_api_order.py", ["905", "120"]), ("fuzz_strips_policy.py", ["906", "25"]), ("fuzz_batch_rows.py", ["907", "10"]),
                                         ("fuzz_large_n.py", ["908", "6"]), ("fuzz_pyramid.py", ["909", "5"])])
def test_fuzz_soak_at_reduced_counts(gpu, script, args):
    """tools/fuzz_*.py (randomised cross-checks of every solve path, the batched launches, the rows around the solve and random orders of
    the no-wait calls against the host loop / the oracles) — a reduced soak inside the GPU suite; a disagreement is a non-zero exit."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
    assert r.returncode == 0, (script, r.returncode, tail)
