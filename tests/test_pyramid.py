"""Coarse-to-fine tracking on one image pyramid (BASELINE.json configs[3]: 4 levels 80x60 .. 640x480, 2 000 -> 16 000 points): the
device pyramid against the numpy oracle bit for bit, and the single-call solve against the oracle's level-by-level solve."""
import numpy as np
import pytest


def test_pyramid_oracle_properties():
    """2x2 box, fp32, order of additions; level sizes and intrinsics (HessianBlocks.cpp:173-176, CoarseTracker.cpp:103-111)."""
    import np_pyramid_oracle as pyo
    rng = np.random.default_rng(0)
    f = rng.standard_normal((61, 83)).astype(np.float32)
    d = pyo.box_down(f)
    assert d.shape == (30, 41) and d.dtype == np.float32
    r, c = 7, 11
    assert d[r, c] == np.float32(0.25) * (((f[2 * r, 2 * c] + f[2 * r, 2 * c + 1]) + f[2 * r + 1, 2 * c]) + f[2 * r + 1, 2 * c + 1])
    assert pyo.box_down(np.full((8, 8), 3.0, np.float32)).tolist() == np.full((4, 4), 3.0).tolist()      # constants stay
    levels = pyo.build_pyramid(rng.standard_normal((480, 640)), 4)
    assert [x.shape for x in levels] == [(480, 640), (240, 320), (120, 160), (60, 80)]
    fx, fy, cx, cy = pyo.level_intrinsics(2, 500.0, 500.0, 319.5, 239.5)
    assert (fx, fy) == (125.0, 125.0) and cx == (319.5 + 0.5) / 4 - 0.5 and cy == (239.5 + 0.5) / 4 - 0.5
    assert pyo.level_intrinsics(0, 500.0, 400.0, 10.0, 20.0) == (500.0, 400.0, 10.0, 20.0)
    # a pixel centre of level l maps to the centre of its 2^l x 2^l block: u_l = (u_0 + 0.5) / 2^l - 0.5
    u0 = 101.3
    assert fx * ((u0 - 319.5) / 500.0) + cx == pytest.approx((u0 + 0.5) / 4 - 0.5)


@pytest.mark.gpu
def test_device_pyramid_is_bit_exact(gpu, capi, synth):
    import np_pyramid_oracle as pyo
    for H, W, L in ((480, 640, 4), (123, 217, 3), (64, 64, 5)):
        rng = np.random.default_rng(H)
        frame = rng.standard_normal((H, W)) * 1e-2
        pyr = capi.Pyramid(capi.default_config(), [64] * L, H, W)
        pyr.set_event_frame(frame)
        ref = pyo.build_pyramid(frame, L)
        for l in range(L):
            assert pyr.level_size(l) == ref[l].shape
            assert np.array_equal(pyr.level_frame(l), ref[l].astype(np.float64)), (H, W, l)
        fx, fy, cx, cy = synth.intrinsics(H, W)
        for l in range(L):
            assert np.array_equal(capi.Pyramid.level_intrinsics(l, fx, fy, cx, cy), np.array(pyo.level_intrinsics(l, fx, fy, cx, cy)))
        pyr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["lm6", "ref12"])
def test_config3_coarse_to_fine_single_call(gpu, capi, synth, po, solver):
    """configs[3]: levels 3..0 at 80x60, 160x120, 320x240, 640x480 with 2 000, 4 000, 8 000, 16 000 points of ONE scene, the pose
    carried from level to level inside one eds_pyr_optimize call; every level's result equals the oracle's at that level."""
    import np_pyramid_oracle as pyo
    counts = [16000, 8000, 4000, 2000]                        # level 0 .. 3
    iters = [6, 6, 6, 6]
    al = synth.make_alignment(3234, H=480, W=640, N=16000, rot_deg=0.6, trans_norm=0.012, blur_ksize=15, blur_sigma=4.0)
    cfgk = dict(exec=capi.EXEC_DEVICE, max_num_iterations=6)
    if solver == "lm6":
        cfg = capi.default_config(solver=capi.SOLVER_LM6, **cfgk)
        okw = {}
    else:
        cfg = capi.default_config(solver=capi.SOLVER_REF12, num_blocks=4, loss_type=capi.LOSS_HUBER, loss_param=0.3, **cfgk)
        okw = dict(num_blocks=4, loss_type=po.LOSS_HUBER, loss_param=0.3)
    pyr = capi.Pyramid(cfg, counts, 480, 640)
    for l, n in enumerate(counts):
        pyr.set_keyframe(l, al.norm_coord[:n], al.grad[:n], al.idp[:n], al.weights[:n], al.fx, al.fy, al.cx, al.cy)
    pyr.set_event_frame(al.frame)
    p, q, v, infos = pyr.optimize(al.p0, al.q0, al.v0)
    rp, rq, rv, per_level = pyo.track(po, synth, al, counts, iters, solver=solver, **okw)
    assert po.se3_distance(p, q, rp, rq) <= 1e-4
    assert np.abs(v - rv).max() <= 1e-4
    for l in range(4):
        if solver == "lm6":
            assert infos[l]["num_iterations"] == per_level[l]["iterations"] and infos[l]["num_points"] == counts[l]
            assert infos[l]["num_successful_steps"] == int(per_level[l]["accepted"].sum())
        else:
            assert infos[l]["num_iterations"] == per_level[l]["num_iterations"]
            assert infos[l]["num_successful_steps"] == per_level[l]["num_successful_steps"]
            assert infos[l]["termination"] == per_level[l]["termination"]
    # coarse-to-fine does its job: closer to the truth than the start, and the finest level's residuals belong to the final pose
    assert po.se3_distance(p, q, al.p_true, al.q_true) < po.se3_distance(al.p0, al.q0, al.p_true, al.q_true)
    r0 = pyr.residuals(0)
    assert r0.shape == (16000,)
    pyr.close()


@pytest.mark.gpu
def test_batched_pyramids_match_single_ones_and_the_oracle(gpu, capi, synth, po):
    """eds_pyr_create_batch: B pyramids in one object, one launch per level for all of them (bench.py's configs[3] leg).  Every pyramid
    of the batch must end where the same pyramid ends when it is solved alone (eds_pyr_optimize), and the first one where the
    oracle's level-by-level solve ends."""
    import np_pyramid_oracle as pyo
    counts, iters, H, W = [6000, 3000, 1500], [6, 6, 6], 240, 320
    als = [synth.make_alignment(3300 + b, H=H, W=W, N=counts[0], rot_deg=0.6, trans_norm=0.012, blur_ksize=9, blur_sigma=2.5) for b in range(3)]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=6)
    pb = capi.Pyramid(cfg, counts, H, W, batch=len(als))
    singles = []
    for b, al in enumerate(als):
        one = capi.Pyramid(cfg, counts, H, W)
        for l, n in enumerate(counts):
            pb.set_keyframe_slot(b, l, al.norm_coord[:n], al.grad[:n], al.idp[:n], al.weights[:n], al.fx, al.fy, al.cx, al.cy)
            one.set_keyframe(l, al.norm_coord[:n], al.grad[:n], al.idp[:n], al.weights[:n], al.fx, al.fy, al.cx, al.cy)
        pb.set_event_frame_slot(b, al.frame)
        one.set_event_frame(al.frame)
        singles.append(one.optimize(al.p0, al.q0, al.v0))
        one.close()
    P, Q, V, infos = pb.optimize_batch(np.stack([a.p0 for a in als]), np.stack([a.q0 for a in als]), np.stack([a.v0 for a in als]))
    for b, al in enumerate(als):
        sp, sq, sv, sinfo = singles[b]
        assert po.se3_distance(P[b], Q[b], sp, sq) <= 1e-9, b
        for l in range(len(counts)):
            assert infos[l][b]["num_iterations"] == sinfo[l]["num_iterations"] and infos[l][b]["num_points"] == counts[l]
    rp, rq, rv, per_level = pyo.track(po, synth, als[0], counts, iters, solver="lm6")
    assert po.se3_distance(P[0], Q[0], rp, rq) <= 1e-4
    # a second call on the same object starts from what it is given, not from where the first ended
    P2, Q2, V2, _ = pb.optimize_batch(np.stack([a.p0 for a in als]), np.stack([a.q0 for a in als]), np.stack([a.v0 for a in als]), want_infos=False)
    assert np.array_equal(P2, P) and np.array_equal(Q2, Q)
    pb.close()
