"""Loss scale on the device and post-solve point maintenance against the CPU oracles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_loss_param_on_device_matches_host_and_oracle(gpu, capi, synth, po):
    als = [synth.make_alignment(900 + b, H=120, W=160, N=n) for b, n in enumerate((1, 2, 63, 64, 500, 777, 1024, 1500))]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=4)
    h = capi.Handle(cfg, len(als), 1500, 120, 160)
    for b, a in enumerate(als):
        h.set_alignment(b, a)
    h.optimize_batch(0, 0, len(als))
    for method, lp in ((capi.LP_MAD, po.LP_MAD), (capi.LP_STD, po.LP_STD)):
        tau_dev = h.loss_param_batch(method)                          # residuals still in HBM: selected on the GPU
        for b, a in enumerate(als):
            p, q, v = h.get_state(b)
            er = po.Oracle(a).pose6_eval(p, q, v)["r"]
            tau_ref, _ = po.loss_param(er, lp)
            r_dev = h.residuals(b)                                    # now also copied to the host
            tau_exact, _ = po.loss_param(r_dev, lp)                   # oracle rule on the very same fp32 residuals
            assert tau_dev[b] == pytest.approx(tau_exact, rel=1e-12, abs=1e-300)
            assert tau_dev[b] == pytest.approx(tau_ref, rel=2e-4, abs=1e-12)
            assert h.loss_param(b, method) == pytest.approx(tau_exact, rel=1e-12, abs=1e-300)     # host path agrees
    h.close()


@pytest.mark.parametrize("delete", [True, False])
def test_update_points_vs_oracle(gpu, capi, synth, po, delete):
    import np_points_oracle as pto
    al = synth.make_alignment(61, H=120, W=160, N=900, margin=2)
    K = (al.fx, al.fy, al.cx, al.cy)
    # a pose that pushes a good fraction of the points out of the frame
    p = np.array([0.06, -0.03, 0.01])
    q = synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.05)
    ref = pto.get_coord(al.norm_coord, al.idp, al.coord, K, al.H, al.W, p, q, delete)
    if delete:
        assert 50 < al.N - len(ref["kept"]) < al.N - 50
    else:
        assert len(ref["kept"]) == al.N
    h = capi.Handle(capi.default_config(exec=capi.EXEC_HOST), 1, al.N, al.H, al.W)
    h.set_alignment(0, al)
    h.set_state(0, p, q, al.v0)
    out = h.update_points(0, delete)
    assert np.array_equal(out["kept"], ref["kept"])
    assert np.abs(out["coord"] - ref["coord"]).max() < 5e-5                      # pixels (fp32 displacement form)
    assert np.abs(out["tracks"] - ref["tracks"]).max() < 5e-5
    assert out["mean_sq_flow"] == pytest.approx(ref["mean_sq_flow"], rel=1e-5)
    assert pto.need_new_keyframe(out["mean_sq_flow"], al.H, al.W) == pto.need_new_keyframe(ref["mean_sq_flow"], al.H, al.W)
    # the compacted device planes now behave exactly like a keyframe uploaded with the kept points only
    keep = ref["kept"]
    al2 = type(al)(**{**al.__dict__, "norm_coord": al.norm_coord[keep], "grad": al.grad[keep], "idp": al.idp[keep],
                      "weights": al.weights[keep], "coord": al.coord[keep]})
    pe, qe = np.array([0.001, 0.002, -0.001]), synth.quat_from_axis_angle([0.3, -0.2, 0.9], 0.004)
    g = h.eval(0, pe, qe, al.v0, ncols=6)
    e = po.Oracle(al2).pose6_eval(pe, qe, al.v0)
    assert g["r"].shape == (len(keep),)
    assert np.abs(g["r"] - e["r"]).max() <= 1e-5 * np.abs(e["r"]).max()
    assert np.linalg.norm(g["JtJ"] - e["H"]) <= 1e-4 * np.linalg.norm(e["H"])
    # ... including a solve of the reference problem with per-block statistics rebuilt for the new point count
    h.set_config(capi.default_config(exec=capi.EXEC_HOST, solver=capi.SOLVER_REF12, num_blocks=3, max_num_iterations=6))
    pr, qr, vr, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    ref12 = po.Oracle(al2, num_blocks=3, max_num_iterations=6).solve_lm(al.p0, al.q0, al.v0)
    assert info["num_points"] == len(keep) and info["num_iterations"] == ref12["num_iterations"]
    assert po.se3_distance(pr, qr, ref12["p"], ref12["q"]) <= 1e-4
    h.close()


def test_update_points_all_in_frame_is_a_no_op(gpu, capi, synth):
    al = synth.make_alignment(62, H=120, W=160, N=300)
    h = capi.Handle(capi.default_config(), 1, al.N, al.H, al.W)
    h.set_alignment(0, al)
    out = h.update_points(0, True)                                    # identity pose: tracks are zero
    assert len(out["kept"]) == al.N and np.array_equal(out["kept"], np.arange(al.N))
    assert np.abs(out["tracks"]).max() < 1e-9 and out["mean_sq_flow"] < 1e-18
    assert np.abs(out["coord"] - al.coord).max() < 1e-9
    h.close()


def test_tracker_mirror_get_coord(gpu, capi, synth):
    import importlib
    import np_points_oracle as pto
    trk = importlib.import_module("slam-eds_amd.tracker")
    al = synth.make_alignment(63, H=120, W=160, N=400, margin=2)
    K = np.array([[al.fx, 0, al.cx], [0, al.fy, al.cy], [0, 0, 1.0]])
    kf = trk.KeyFrame(al.norm_coord.copy(), al.grad.copy(), al.weights.copy(), al.idp.copy(), K, al.H, al.W)
    t = trk.Tracker(kf, trk.Config())
    p, q = np.array([0.05, 0.02, 0.0]), synth.quat_from_axis_angle([0.0, 1.0, 0.1], 0.04)
    t.reset(kf, p, q, True)
    ref = pto.get_coord(al.norm_coord, al.idp, al.coord, (al.fx, al.fy, al.cx, al.cy), al.H, al.W, p, q, True)
    coord = t.getCoord(True)
    assert coord.shape == ref["coord"].shape and np.abs(coord - ref["coord"]).max() < 5e-5
    assert len(kf.inv_depth) == len(ref["kept"]) and np.array_equal(kf.inv_depth, al.idp[ref["kept"]])
    assert t.squared_norm_flow == pytest.approx(ref["mean_sq_flow"], rel=1e-5)
    assert t.needNewKeyframe(0.03) == pto.need_new_keyframe(ref["mean_sq_flow"], al.H, al.W, 0.03)
    t.close()


@pytest.mark.parametrize("N", [4097, 9000, 16000])
def test_point_maintenance_and_loss_scale_beyond_4096_points(gpu, capi, synth, po, N):
    """The pyramid configurations carry 8 000 - 16 000 points: culling sweeps 4 096 points at a time (order-preserving
    across sweeps), the MAD sort takes up to 16 384 keys in LDS."""
    import np_points_oracle as pto
    al = synth.make_alignment(64, H=480, W=640, N=N, margin=2)
    K = (al.fx, al.fy, al.cx, al.cy)
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=3), 1, N, al.H, al.W)
    h.set_alignment(0, al)
    h.optimize_batch(0, 0, 1)
    tau = h.loss_param_batch(capi.LP_MAD)[0]                                # residuals still on the device
    r = h.residuals(0)
    assert tau == pytest.approx(po.loss_param(r, po.LP_MAD)[0], rel=1e-12)
    p = np.array([0.08, -0.05, 0.01])
    q = synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.04)
    ref = pto.get_coord(al.norm_coord, al.idp, al.coord, K, al.H, al.W, p, q, True)
    assert 100 < N - len(ref["kept"]) < N - 100
    h.set_state(0, p, q, al.v0)
    out = h.update_points(0, True)
    assert np.array_equal(out["kept"], ref["kept"])
    assert np.abs(out["coord"] - ref["coord"]).max() < 1e-4
    assert out["mean_sq_flow"] == pytest.approx(ref["mean_sq_flow"], rel=1e-5)
    # the compacted planes behave like a fresh upload of the kept points
    keep = ref["kept"]
    al2 = type(al)(**{**al.__dict__, "norm_coord": al.norm_coord[keep], "grad": al.grad[keep], "idp": al.idp[keep],
                      "weights": al.weights[keep], "coord": al.coord[keep]})
    pe, qe = np.array([0.001, 0.002, -0.001]), synth.quat_from_axis_angle([0.3, -0.2, 0.9], 0.004)
    h.set_config(capi.default_config(exec=capi.EXEC_HOST))
    g = h.eval(0, pe, qe, al.v0, ncols=6)
    e = po.Oracle(al2).pose6_eval(pe, qe, al.v0)
    assert g["r"].shape == (len(keep),) and np.abs(g["r"] - e["r"]).max() <= 1e-5 * np.abs(e["r"]).max()
    h.close()


@pytest.mark.parametrize("want_points", [True, False])
def test_update_points_batch_vs_oracle_and_single_calls(gpu, capi, synth, po, want_points):
    """eds_trk_update_points_batch: 70 alignments (more than one launch of 64) of ragged point counts, each at its own pose — kept
    indices, coordinates, tracks, counts and mean squared flow per alignment against the oracle; the compacted planes then solve like
    keyframes uploaded with the kept points only; with the point arrays NULL only counts and flow come back."""
    import np_points_oracle as pto
    B = 70
    rng = np.random.default_rng(17)
    als = [synth.make_alignment(700 + b, H=120, W=160, N=int(rng.integers(64, 900)), margin=2) for b in range(8)]
    h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=4), B + 2, 900, 120, 160)
    poses, refs = [], []
    for b in range(B):
        al = als[b % 8]
        h.set_alignment(1 + b, al)
        p = np.array([0.06, -0.03, 0.01]) * rng.uniform(-1.0, 1.0, 3)
        q = synth.quat_from_axis_angle(rng.standard_normal(3), 0.05 * rng.uniform())
        h.set_state(1 + b, p, q, al.v0)
        poses.append((p, q))
        refs.append(pto.get_coord(al.norm_coord, al.idp, al.coord, (al.fx, al.fy, al.cx, al.cy), al.H, al.W, p, q, True))
    h.set_alignment(0, als[0]); h.set_alignment(B + 1, als[1])                 # neighbours that must stay as they are
    outs = h.update_points_batch(1, B, True, want_points=want_points)
    erased = 0
    for b in range(B):
        ref, out = refs[b], outs[b]
        assert out["n"] == len(ref["kept"]), b
        erased += als[b % 8].N - out["n"]
        assert out["mean_sq_flow"] == pytest.approx(ref["mean_sq_flow"], rel=1e-5)
        if want_points:
            assert np.array_equal(out["kept"], ref["kept"])
            assert np.abs(out["coord"] - ref["coord"]).max() < 5e-5
            assert np.abs(out["tracks"] - ref["tracks"]).max() < 5e-5
    assert erased > 100                                                        # the poses do push points out
    # the compacted planes: a solve equals the oracle's on the kept points (a few alignments)
    for b in (0, 13, 64, 69):
        al, keep = als[b % 8], refs[b]["kept"]
        if len(keep) < 32:
            continue
        al2 = type(al)(**{**al.__dict__, "norm_coord": al.norm_coord[keep], "grad": al.grad[keep], "idp": al.idp[keep],
                          "weights": al.weights[keep], "coord": al.coord[keep]})
        h.set_state(1 + b, al.p0, al.q0, al.v0)
        h.optimize_batch(0, 1 + b, 1)
        tab = h.results(1 + b, 1)[0]
        ref = po.Oracle(al2).pose6_lm(al.p0, al.q0, al.v0, iters=4, lambda0=0.01)
        assert po.se3_distance(tab[0:3], tab[3:7], ref["p"], ref["q"]) <= 1e-4
    # neighbours untouched
    for s_, al in ((0, als[0]), (B + 1, als[1])):
        out = h.update_points(s_, False)
        assert out["coord"].shape[0] == al.N
    h.close()


def test_batched_entry_points_with_one_slot_equal_the_single_slot_ones(gpu, capi, synth):
    """count = 1 of the batched getCoord / loss scale / event-frame entry points against their single-slot twins (same kernels, the
    batch dimension collapsed): identical outputs."""
    al = synth.make_alignment(77, H=120, W=160, N=700, margin=2)
    p = np.array([0.05, -0.02, 0.01]); q = synth.quat_from_axis_angle([0.2, 1.0, 0.1], 0.04)
    outs = []
    for batched in (False, True):
        h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=3), 2, al.N, al.H, al.W)
        h.set_alignment(1, al)
        h.set_state(1, al.p0, al.q0, al.v0)
        h.optimize_batch(0, 1, 1)
        tau = h.loss_param_batch(capi.LP_MAD, 1, 1)[0] if batched else h.loss_param(1, capi.LP_MAD)
        h.set_state(1, p, q, al.v0)
        o = h.update_points_batch(1, 1, True)[0] if batched else h.update_points(1, True)
        outs.append((tau, o))
        h.close()
    (t0, o0), (t1, o1) = outs
    assert t0 == pytest.approx(t1, rel=1e-12)
    assert np.array_equal(o0["kept"], o1["kept"]) and np.array_equal(o0["coord"], o1["coord"]) and np.array_equal(o0["tracks"], o1["tracks"])
    assert o0["mean_sq_flow"] == o1["mean_sq_flow"] and 0 < len(o0["kept"]) < al.N


def test_loss_param_on_device_with_all_equal_and_duplicated_residuals(gpu, capi, synth, po):
    """Radix select corner cases: most residuals equal (zero weights -> residual 0: no pass ever narrows to one candidate, median and
    MAD both inside the run), and heavy duplication (half the weights zero: the median sits at the edge of a run of equal keys)."""
    for N, zero_from in ((300, 40), (1001, 400), (2000, 1000)):      # (40 live points: the median AND the MAD lie in the run of zeros)
        al = synth.make_alignment(990 + N, H=120, W=160, N=N)
        w = al.weights.copy(); w[zero_from:] = 0.0
        h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=2), 1, N, 120, 160)
        h.set_keyframe(0, al.norm_coord, al.grad, al.idp, w, al.fx, al.fy, al.cx, al.cy)
        h.set_event_frame(0, al.frame)
        h.set_state(0, al.p0, al.q0, al.v0)
        h.optimize_batch(0, 0, 1)
        tau_dev = h.loss_param_batch(capi.LP_MAD, 0, 1)[0]
        r = h.residuals(0)
        assert (r == 0).sum() >= N - zero_from
        tau_exact, _ = po.loss_param(r, po.LP_MAD)
        assert tau_dev == pytest.approx(tau_exact, rel=1e-12, abs=1e-300)
        h.close()


@pytest.mark.parametrize("solver", ["lm6", "ref12"])
def test_residuals_and_loss_is_the_three_call_sequence(gpu, capi, synth, po, solver):
    """eds_trk_residuals_and_loss = Tracker.cpp:223-233 in one call: the residuals as get_residuals -> loss_param(MAD) -> get_residuals
    leaves them (the MAD's n_quantile_vector reorders kf->residuals in place) and the same tau, bit for bit; STD and CONSTANT too."""
    al = synth.make_alignment(41, H=120, W=160, N=777)
    sv = capi.SOLVER_LM6 if solver == "lm6" else capi.SOLVER_REF12
    cfg = capi.default_config(solver=sv, exec=capi.EXEC_DEVICE, max_num_iterations=6)
    for method, lp in ((capi.LP_MAD, po.LP_MAD), (capi.LP_STD, po.LP_STD), (capi.LP_CONSTANT, None)):
        ha, hb = capi.Handle(cfg, 1, al.N, al.H, al.W), capi.Handle(cfg, 1, al.N, al.H, al.W)
        for h in (ha, hb):
            h.set_alignment(0, al)
            h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
        r_first = ha.residuals(0)
        tau_a = ha.loss_param(0, method, current=0.25)
        r_a = ha.residuals(0)
        r_b, tau_b = hb.residuals_and_loss(0, method, current=0.25)
        assert np.array_equal(r_a, r_b) and tau_a == tau_b
        assert np.array_equal(np.sort(r_b), np.sort(r_first))                     # a permutation of the residuals at the solution
        if method == capi.LP_MAD:
            assert not np.array_equal(r_b, r_first)                               # ... and the MAD did reorder them
            assert tau_b == pytest.approx(po.loss_param(r_first, lp)[0], rel=1e-12)
        if method == capi.LP_CONSTANT:
            assert tau_b == 0.25 and np.array_equal(r_b, r_first)
        ha.close(); hb.close()


def test_set_idepth_in_one_launch_serves_device_and_host_readers(gpu, capi, synth, po):
    """set_idepth refreshes the rho plane and the Gram matrices with ONE launch and leaves the host copy of the Gram matrices stale;
    a device solve must see the new depths at once, and a host-side reader (eval, the host-driven loop) must fetch the matrices first:
    both equal a fresh handle that got the new depths through set_keyframe."""
    al = synth.make_alignment(43, H=120, W=160, N=900)
    rng = np.random.default_rng(5)
    idp2 = al.idp * rng.uniform(0.8, 1.25, al.N)
    al2 = synth.Alignment(**{**al.__dict__, "idp": idp2})
    for solver in (capi.SOLVER_LM6, capi.SOLVER_REF12):
        cfg = capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=6, num_blocks=2 if solver == capi.SOLVER_REF12 else 1)
        h, fresh = capi.Handle(cfg, 1, al.N, al.H, al.W), capi.Handle(cfg, 1, al.N, al.H, al.W)
        h.set_alignment(0, al); fresh.set_alignment(0, al2)
        h.optimize(0, p=al.p0, q=al.q0, v=al.v0)                                  # something in flight before the refresh
        h.set_idepth(0, idp2)
        got = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
        want = fresh.optimize(0, p=al.p0, q=al.q0, v=al.v0)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[3]["num_iterations"] == want[3]["num_iterations"]
        ncols = 12 if solver == capi.SOLVER_REF12 else 6
        h.set_idepth(0, al.idp); h.set_idepth(0, idp2)                            # twice in a row: the staging is re-used behind its event
        e1, e2 = h.eval(0, al.p0, al.q0, al.v0, ncols=ncols), fresh.eval(0, al.p0, al.q0, al.v0, ncols=ncols)
        assert np.array_equal(e1["r"], e2["r"]) and np.array_equal(e1["JtJ"], e2["JtJ"])
        h.close(); fresh.close()
