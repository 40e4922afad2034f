"""Speculative candidate groups of the LM6 team kernel (csrc/eds_fused.hip, template argument GROUPS; rule: csrc/eds_launch_rule.hpp).

G teams of K CUs evaluate G prepared LM candidates per round and every workgroup replays edss::Solver6::on_eval over the G results in
order.  The claim under test: decisions, lambdas, trace records, iteration counts, poses and kept residuals are those of the sequential
solver BIT FOR BIT (every candidate's sums are added in the same member order as a team of K adds them), and they are the CPU oracle's
(reference template: CoarseTracker.cpp:545-664 accept / reject with lambda x 0.5 / x 4; path: Tracker.cpp:104-241)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(capi, als, frames, B, groups, iters=10, lambda0=0.01, sampling=None, huber_tau=0.0, starts=None):
    cfg = capi.default_config(sampling=capi.SAMPLE_BICUBIC if sampling is None else sampling, solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE,
                              max_num_iterations=iters, lambda0=lambda0, huber_tau=huber_tau)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    if groups is not None:
        h.set_knob("EDS_LM6_GROUPS", str(groups))
    for b in range(B):
        a = als[b % len(als)]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        h.set_event_frame(b, frames[b % len(als)])
    P, Q, V = starts if starts is not None else (np.stack([als[b % len(als)].p0 for b in range(B)]), np.stack([als[b % len(als)].q0 for b in range(B)]),
                                                 np.stack([als[b % len(als)].v0 for b in range(B)]))
    h.set_states(0, P, Q, V)
    h.optimize_batch(0, 0, B)
    tab = h.results(0, B).copy()
    res = [h.residuals(b) for b in range(min(B, 6))]
    tr = [h.trace(b) for b in range(min(B, 6))]
    li, fl = h.last_launch(), h.info(0)["flags"]
    h.close()
    return tab, res, tr, li, fl


def _same(x, y):
    return (np.array_equal(x[0], y[0]) and all(np.array_equal(a, b) for a, b in zip(x[1], y[1])) and
            all(np.array_equal(a["accepted"], b["accepted"]) and np.array_equal(a["costs"], b["costs"]) and np.array_equal(a["increments"], b["increments"])
                for a, b in zip(x[2], y[2])))


@pytest.fixture(scope="module")
def scene(synth):
    als = [synth.make_alignment(5000 + i) for i in range(4)] + [synth.make_alignment(1234), synth.make_alignment(77, N=1500)]
    return als, [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]


def test_groups_equal_the_sequential_team_solver_bit_for_bit_and_the_oracle(gpu, capi, synth, po, scene):
    als, fr = scene
    for B in (1, 6, 16, 32):
        base = _solve(capi, als, fr, B, 1)
        assert base[3]["cus_per_alignment"] == 4 and base[4] == 0
        for G in (None, 2, 4, 8):
            if G is not None and B * 4 * G > 256:
                continue                                    # (beyond one workgroup per CU the rule never forms groups; tools/check_groups.py times it)
            r = _solve(capi, als, fr, B, G)
            assert r[4] == 0, "team time-out"
            if G is not None:
                assert r[3]["cus_per_alignment"] == 4 * G and r[3]["kernel"].endswith(f", 4, {G}>"), r[3]
            else:
                assert r[3]["cus_per_alignment"] > 4, r[3]   # the rule forms groups in this regime
            assert _same(r, base), (B, G)
        for b in range(min(B, 6)):
            a = als[b % len(als)]
            x32 = synth.Alignment(**{**a.__dict__, "frame": fr[b % len(als)].astype(np.float64)})
            ref = po.Oracle(x32).pose6_lm(a.p0, a.q0, a.v0, iters=10, lambda0=0.01)
            assert np.array_equal(base[2][b]["accepted"], ref["accepted"]) and base[0][b, 14] == ref["iterations"]
            assert po.se3_distance(base[0][b, 0:3], base[0][b, 3:7], ref["p"], ref["q"]) <= 1e-6          # tolerance: SURVEY 8c asks 1e-4; measured 1e-9
            o = po.Oracle(x32).pose6_eval(base[0][b, 0:3], base[0][b, 3:7], a.v0)
            assert np.max(np.abs(base[1][b] - o["r"])) <= 1e-5 * np.max(np.abs(o["r"]))                    # residuals at the returned pose (Tracker.cpp:223-230)


@pytest.mark.parametrize("case", ["one_iteration", "three_iterations", "all_rejected", "candidates_used_up", "warm_start", "huber", "bilinear"])
def test_groups_edge_cases_equal_the_sequential_solver(gpu, capi, synth, po, scene, case):
    als, fr = scene
    kw = {}
    if case == "one_iteration": kw = dict(iters=1)
    elif case == "three_iterations": kw = dict(iters=3)
    elif case == "all_rejected": kw = dict(iters=4, lambda0=1e-9)              # (lambda is clamped to 1e-6 after the first rejection: GN-sized steps, rejected on this problem)
    elif case == "candidates_used_up": kw = dict(iters=24, lambda0=1e-12)      # more than EDS_NSPEC rejections in a row: proposals are made again
    elif case == "huber": kw = dict(huber_tau=0.004)
    elif case == "bilinear": kw = dict(sampling=capi.SAMPLE_BILINEAR)
    B = 3
    if case == "warm_start":                                                    # from the solved pose: mostly accepted small steps
        t0 = _solve(capi, als, fr, B, 1)[0]
        kw = dict(starts=(t0[:, 0:3].copy(), t0[:, 3:7].copy(), np.stack([als[b % len(als)].v0 for b in range(B)])))
    base = _solve(capi, als, fr, B, 1, **kw)
    seen = set()
    for G in (2, 4, 8):
        r = _solve(capi, als, fr, B, G, **kw)
        assert r[4] == 0 and r[3]["cus_per_alignment"] == 4 * G
        assert _same(r, base), (case, G)
        seen.add("".join(str(int(x)) for x in r[2][0]["accepted"]))
    assert len(seen) == 1
    if case in ("one_iteration", "three_iterations", "all_rejected", "candidates_used_up", "huber"):
        a = als[0]
        x32 = synth.Alignment(**{**a.__dict__, "frame": fr[0].astype(np.float64)})
        ref = po.Oracle(x32).pose6_lm(a.p0, a.q0, a.v0, iters=kw.get("iters", 10), lambda0=kw.get("lambda0", 0.01), huber_tau=kw.get("huber_tau", 0.0))
        assert np.array_equal(base[2][0]["accepted"], ref["accepted"]), (case, base[2][0]["accepted"], ref["accepted"])
        assert po.se3_distance(base[0][0, 0:3], base[0][0, 3:7], ref["p"], ref["q"]) <= 1e-6
    if case == "candidates_used_up":
        assert "0" * 9 in seen.pop()                                            # the run of rejections really outlasted the prepared candidates


# ---- the reference problem (REF12): eds_fused12_kernel's candidate groups ---------------------------------------------------------------
def _solve12(capi, als, frames, B, groups, nb=1, loss=None, iters=10):
    cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=iters, num_blocks=nb,
                              loss_type=capi.LOSS_NONE if loss is None else loss, loss_param=0.3)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    if groups is not None:
        h.set_knob("EDS_REF12_GROUPS", str(groups))
    for b in range(B):
        a = als[b % len(als)]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        h.set_event_frame(b, frames[b % len(als)])
    h.set_states(0, np.stack([als[b % len(als)].p0 for b in range(B)]), np.stack([als[b % len(als)].q0 for b in range(B)]), np.stack([als[b % len(als)].v0 for b in range(B)]))
    h.optimize_batch(0, 0, B)
    tab = h.results(0, B).copy()
    res = [h.residuals(b) for b in range(min(B, 4))]
    infos = [h.info(b) for b in range(min(B, 6))]
    li = h.last_launch()
    h.close()
    return tab, res, infos, li


@pytest.mark.parametrize("nb,loss", [(1, None), (4, "huber"), (2, "cauchy")])
def test_ref12_groups_take_the_reference_solvers_decisions(gpu, capi, synth, po, scene, nb, loss):
    """Ceres' trust-region LM as Tracker::optimize runs it (Tracker.cpp:104-241), G prepared steps per round: step accounting and termination
    equal the one-team launch's and the oracle's; state and kept residuals equal the one-team launch's to 1e-9 (REF12 is not bit-reproducible
    from run to run with or without groups: the wavefronts' tiles meet in LDS by fp64 atomics) and the oracle's to 1e-6 / 1e-5."""
    als, fr = scene
    lt = {None: capi.LOSS_NONE, "huber": capi.LOSS_HUBER, "cauchy": capi.LOSS_CAUCHY}[loss]
    pl = {None: po.LOSS_NONE, "huber": po.LOSS_HUBER, "cauchy": po.LOSS_CAUCHY}[loss]
    for B in (1, 6, 20):
        base = _solve12(capi, als, fr, B, 1, nb, lt)
        assert base[3]["cus_per_alignment"] in (4, 8) and base[2][0]["flags"] == 0
        K = base[3]["cus_per_alignment"]
        for G in (None, 2, 4):
            if G is not None and (B * K * G > 512 or (K == 4 and G == 4)):
                continue
            r = _solve12(capi, als, fr, B, G, nb, lt)
            assert r[2][0]["flags"] == 0, "team time-out"
            if G is not None:
                assert r[3]["cus_per_alignment"] == K * G and r[3]["kernel"].endswith(f", {K}, 0, {G}>"), r[3]
            else:
                assert r[3]["cus_per_alignment"] > K, r[3]       # the rule forms groups in this regime
            assert np.abs(r[0][:, :13] - base[0][:, :13]).max() <= 1e-9 and np.array_equal(r[0][:, 14:16], base[0][:, 14:16]), (B, G)
            assert all(np.abs(x - y).max() <= 1e-9 for x, y in zip(r[1], base[1])), (B, G)
            for x, y in zip(r[2], base[2]):
                assert (x["num_successful_steps"], x["num_unsuccessful_steps"], x["termination"]) == (y["num_successful_steps"], y["num_unsuccessful_steps"], y["termination"])
                assert abs(x["final_cost"] - y["final_cost"]) <= 1e-12 * max(1.0, abs(y["final_cost"]))
        for b in range(min(B, 4)):
            a = als[b % len(als)]
            x32 = synth.Alignment(**{**a.__dict__, "frame": fr[b % len(als)].astype(np.float64)})
            o = po.Oracle(x32, num_blocks=nb, loss_type=pl, loss_param=0.3, max_num_iterations=10)
            ref = o.solve_lm(a.p0, a.q0, a.v0)
            i = base[2][b]
            assert (i["num_successful_steps"], i["num_unsuccessful_steps"], i["termination"]) == (ref["num_successful_steps"], ref["num_unsuccessful_steps"], ref["termination"])
            assert po.se3_distance(base[0][b, 0:3], base[0][b, 3:7], ref["p"], ref["q"]) <= 1e-6 and np.abs(base[0][b, 7:13] - ref["v"]).max() <= 1e-6
            er = o.eval12(base[0][b, 0:3], base[0][b, 3:7], base[0][b, 7:13], jac=False)["r_raw"]
            assert np.abs(base[1][b] - er).max() <= 1e-5 * np.abs(er).max()


def test_ref12_groups_short_budgets_and_warm_start(gpu, capi, synth, po, scene):
    als, fr = scene
    for iters in (1, 2, 3):
        base = _solve12(capi, als, fr, 3, 1, iters=iters)
        for G in (2, 4):
            r = _solve12(capi, als, fr, 3, G, iters=iters)
            assert np.abs(r[0][:, :13] - base[0][:, :13]).max() <= 1e-9 and np.array_equal(r[0][:, 14:16], base[0][:, 14:16]), (iters, G)
            assert [(x["num_successful_steps"], x["num_unsuccessful_steps"], x["termination"]) for x in r[2]] == \
                   [(x["num_successful_steps"], x["num_unsuccessful_steps"], x["termination"]) for x in base[2]]
