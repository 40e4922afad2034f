"""GPU parity of the BATCHED workloads exactly as bench.py and BASELINE.json configs[4] run them.

* configs[4]: 64 independent alignments (seeds 5000 + b, 640x480, 2 000 points), sharded 8 per rank over 8 ranks
  (slam-eds_amd/batch.py), results gathered into the 64 x 16 table — here the 8 shards run one after the other on the
  one GPU of the test box (RCCL needs one device per rank; the collective itself is covered by tests/test_distributed.py),
  every row checked against the CPU oracle.
* the kernel behind bench.py's headline number, selected by `optimize`'s OWN rule (no EDS_LM6_KERNEL override): a launch of
  >= 1 536 alignments at 640x480 / 2 000 points, >= 32 distinct alignments against the oracle.

Tolerances as in tests/test_parity_gpu.py (fp32 kernels vs the fp64 oracle): solved pose within 1e-4 (SE(3) distance), LM6
accept pattern identical, REF12 iteration counts / termination identical, velocity within 1e-4.
"""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_POSE = 1e-4


@pytest.fixture(scope="module")
def batchmod():
    return importlib.import_module("slam-eds_amd.batch")


@pytest.fixture(scope="module")
def als64(synth):
    return [synth.make_alignment(5000 + b) for b in range(64)]            # seeds 5000 + b, 640x480, N = 2000


def test_config4_64_alignments_8_shards_lm6(gpu, capi, synth, po, batchmod, als64):
    assert not os.environ.get("EDS_LM6_KERNEL")
    total, world = 64, 8
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10)
    blocks, traces = [], []
    for rank in range(world):
        bt = batchmod.BatchTracker(cfg, total, 2000, 480, 640, rank=rank, world_size=world)
        assert (bt.first, bt.count) == (8 * rank, 8)
        mine = als64[bt.first: bt.first + bt.count]
        bt.load(mine)
        bt.reset_states(mine)
        bt.solve()
        blocks.append(bt.local_results())
        traces += [bt.handle.trace(i)["accepted"].copy() for i in range(bt.count)]
        bt.close()
    table = np.concatenate(blocks, axis=0)                                  # what the all-gather assembles (rank order)
    assert table.shape == (64, 16)
    worst = 0.0
    for b in range(total):
        al = als64[b]
        ref = po.Oracle(al).pose6_lm(al.p0, al.q0, al.v0, iters=10, lambda0=cfg.lambda0)
        d = po.se3_distance(table[b, 0:3], table[b, 3:7], ref["p"], ref["q"])
        worst = max(worst, d)
        assert d <= TOL_POSE, (b, d)
        assert np.array_equal(traces[b], ref["accepted"]), b
        assert table[b, 14] == ref["iterations"] and table[b, 15] == 1.0
        assert np.allclose(table[b, 7:13], al.v0)                          # pose-only solver: velocity untouched
    print(f"configs[4] LM6: worst SE(3) distance to the oracle {worst:.2e}")


def test_config4_ref12_subset(gpu, capi, synth, po, batchmod, als64):
    """The reference's own 12-parameter problem on a subset of configs[4] (8 alignments of the batch, both residual-block layouts)."""
    sub = [0, 9, 18, 27, 36, 45, 54, 63]
    for nb, loss in ((1, 0), (4, 1)):
        cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb,
                                  loss_type=loss, loss_param=0.3)
        bt = batchmod.BatchTracker(cfg, len(sub), 2000, 480, 640)
        mine = [als64[b] for b in sub]
        bt.load(mine); bt.reset_states(mine); bt.solve()
        table = bt.gather()
        for i, b in enumerate(sub):
            al = als64[b]
            ref = po.Oracle(al, num_blocks=nb, loss_type=loss, loss_param=0.3, max_num_iterations=10).solve_lm(al.p0, al.q0, al.v0)
            assert po.se3_distance(table[i, 0:3], table[i, 3:7], ref["p"], ref["q"]) <= TOL_POSE, (nb, b)
            assert np.abs(table[i, 7:13] - ref["v"]).max() <= 1e-4
            info = bt.handle.info(i)
            assert info["num_iterations"] == ref["num_iterations"] and info["num_successful_steps"] == ref["num_successful_steps"]
            assert info["termination"] == ref["termination"]
            assert table[i, 13] == pytest.approx(ref["final_cost"], rel=1e-5)
        bt.close()


def test_bench_shape_kernel_selected_by_optimize(gpu, capi, synth, po, als64):
    """1 536 alignments at the bench shape in ONE launch (round 1 switched kernels at this size; today the register-resident kernel
    with the quad-cooperative gather runs every batch size, and this is the shape the headline number is measured on) with no
    environment override; 32 distinct alignments, every distinct one checked against the oracle in two different slots."""
    for k in ("EDS_LM6_KERNEL", "EDS_FUSED_THREADS", "EDS_FUSED_PPT"):
        assert not os.environ.get(k), f"{k} must be unset: this test exercises optimize()'s own kernel choice"
    B, D = 1536, 32
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als64[:D]]
    for b in range(B):
        a = als64[b % D]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        h.set_event_frame(b, fr[b % D])
    S0 = (np.stack([als64[b % D].p0 for b in range(B)]), np.stack([als64[b % D].q0 for b in range(B)]), np.stack([als64[b % D].v0 for b in range(B)]))
    refs = []
    for d in range(D):
        a = als64[d]
        # the frame was handed over as fp32 (bench.py does the same): the oracle sees the same fp32-rounded frame
        a32 = synth.Alignment(**{**a.__dict__, "frame": fr[d].astype(np.float64)})
        ref = po.Oracle(a32).pose6_lm(a.p0, a.q0, a.v0, iters=10, lambda0=cfg.lambda0)
        refs.append((a32, ref, po.Oracle(a32).pose6_eval(ref["p"], ref["q"], a.v0)))
    worst = 0.0
    tables = []
    # the FIRST solve of these frames samples the tiles they were written in (what a live tracker runs), the SECOND one — the same
    # frames again — their strip copies: the headline instantiation <0, 4, 512, 3, 1>.  Both against the oracle, row by row.
    for kernel, layout in (("eds_fused6_kernel<0, 4, 512, 1, 1>", 1), ("eds_fused6_kernel<0, 4, 512, 3, 1>", 2)):
        h.set_states(0, *S0)
        h.optimize_batch(0, 0, B)
        li = h.last_launch()
        assert li["kernel"] == kernel and li["layout"] == layout and li["cus_per_alignment"] == 1, li
        table = h.results(0, B)
        tables.append(table.copy())
        assert np.all(table[:, 15] == 1.0)
        for d in range(D):
            a = als64[d]
            a32, ref, e = refs[d]
            for slot in (d, d + B - D):                                        # first and last replica of this alignment
                dist = po.se3_distance(table[slot, 0:3], table[slot, 3:7], ref["p"], ref["q"])
                worst = max(worst, dist)
                assert dist <= TOL_POSE, (kernel, slot, dist)
                assert np.array_equal(h.trace(slot)["accepted"], ref["accepted"]), (kernel, slot)
                assert table[slot, 14] == ref["iterations"]
            r = h.residuals(d)
            assert np.abs(r - e["r"]).max() <= 2e-5 * np.abs(e["r"]).max()      # residuals at the solution (Tracker.cpp:223-230)
        # replicas of one alignment must agree bit for bit (same inputs, same kernel, no cross-slot state)
        for slot in range(D, B):
            assert np.array_equal(table[slot, 0:7], table[slot % D, 0:7]), (kernel, slot)
    assert np.abs(tables[0][:, :7] - tables[1][:, :7]).max() < 1e-6 and np.array_equal(tables[0][:, 14], tables[1][:, 14])
    print(f"bench shape, {B} alignments in one launch: worst SE(3) distance to the oracle {worst:.2e}")
    h.close()


def test_ref12_batch_kernels_at_the_bench_shape_vs_oracle(gpu, capi, synth, po, als64):
    """The reference problem on a batch (two alignments per CU: eds_fused12_kernel<0, 256, 320, false, 1, Q>) at 640x480 / 2 000 points,
    320 alignments in one launch: a first solve on new frames (the lane gather on the tiles below 1 024 alignments), the same frames again
    (Q = 2: the quad gather on their strip copies — the instantiation bench.py's REF12 leg times).  Iteration counts, successful steps,
    termination, pose, velocity and cost of 16 distinct alignments in two slots each against the oracle's Ceres-LM restatement."""
    B, D = 320, 16
    cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=1)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als64[:D]]
    for b in range(B):
        a = als64[b % D]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        h.set_event_frame(b, fr[b % D])
    S0 = (np.stack([als64[b % D].p0 for b in range(B)]), np.stack([als64[b % D].q0 for b in range(B)]), np.stack([als64[b % D].v0 for b in range(B)]))
    refs = []
    for d in range(D):
        a = als64[d]
        a32 = synth.Alignment(**{**a.__dict__, "frame": fr[d].astype(np.float64)})
        refs.append(po.Oracle(a32, num_blocks=1, max_num_iterations=10).solve_lm(a.p0, a.q0, a.v0))
    tabs = []
    for kernel, layout in (("eds_fused12_kernel<0, 256, 320, false, 1, 0>", 1), ("eds_fused12_kernel<0, 256, 320, false, 1, 2>", 2)):
        h.set_states(0, *S0)
        h.optimize_batch(0, 0, B)
        li = h.last_launch()
        assert li["kernel"] == kernel and li["layout"] == layout, li
        tab = h.results(0, B)
        tabs.append(tab.copy())
        for d in range(D):
            ref = refs[d]
            for slot in (d, d + B - D):
                info = h.info(slot)
                assert info["num_iterations"] == ref["num_iterations"] and info["num_successful_steps"] == ref["num_successful_steps"], (kernel, slot)
                assert info["termination"] == ref["termination"] and bool(info["success"]) == ref["usable"]
                assert po.se3_distance(tab[slot, 0:3], tab[slot, 3:7], ref["p"], ref["q"]) <= TOL_POSE, (kernel, slot)
                assert np.abs(tab[slot, 7:13] - ref["v"]).max() <= 1e-4
                assert info["final_cost"] == pytest.approx(ref["final_cost"], rel=1e-4)
    assert np.abs(tabs[0][:, :13] - tabs[1][:, :13]).max() < 1e-5
    h.close()


@pytest.mark.parametrize("sampling", [0, 1])
@pytest.mark.parametrize("npts", [900, 2000])
def test_strip_kernels_warm_started_at_the_solution(gpu, capi, synth, po, sampling, npts):
    """The strip kernels consume a lane's points in groups behind COUNTED waits (s_waitcnt vmcnt(n), n = the row loads the wavefront really
    issued for the later group, eds_fused.hip): a cold solve issues every load.  Here the second and third solves START AT THE SOLUTION of
    the one before: the candidates move the points by fractions of a pixel, nearly every patch is still in the landing zone, whole load
    instructions are skipped and the counts run through their small values — the result must still be the oracle's from the same start
    (and the kernel the strips one: 2 and 4 points per lane)."""
    H, W, D = 120, 160, 4
    B = 40 if npts <= 1024 else 136          # (2 000 points: up to 128 alignments go out on teams of 4 CUs, which sample the tiles)
    als = [synth.make_alignment(1300 + i, H=H, W=W, N=npts - 11 * i, margin=2) for i in range(D)]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, sampling=sampling, max_num_iterations=6)
    h = capi.Handle(cfg, B, npts, H, W)
    for b in range(B):
        h.set_alignment(b, als[b % D])
    h.prepare_frames(0, B)
    ps, qs = np.array([1e-3, -2e-3, 5e-4]), synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    P, Q, V = np.stack([ps] * B), np.stack([qs] * B), np.stack([als[b % D].v0 for b in range(B)])
    for round_ in range(3):
        h.set_states(0, P, Q, V)
        h.optimize_batch(0, 0, B)
        tab = h.results(0, B)
        li = h.last_launch()
        assert li["layout"] == 2 and li["kernel"].startswith(f"eds_fused6_kernel<{sampling}, {2 if npts <= 1024 else 4}, 512, 3, 1>"), li
        for d, a in enumerate(als):
            ref = po.Oracle(a, sampling=sampling).pose6_lm(P[d], Q[d], a.v0, iters=6, lambda0=cfg.lambda0)
            for slot in (d, d + D * ((B - 1 - d) // D)):
                dist = po.se3_distance(tab[slot, 0:3], tab[slot, 3:7], ref["p"], ref["q"])
                # the bilinear sampler's derivative is one-sided at pixel boundaries: a warm start may sit on one — its poses are
                # compared at the fp32 level of the sampler, the bicubic ones at the suite's tolerance
                assert dist <= (TOL_POSE if sampling == 0 else 2e-3), (round_, li["kernel"], slot, dist)
                assert np.array_equal(tab[slot, 0:7], tab[d, 0:7]), (round_, slot)          # replicas bit for bit
        P, Q = tab[:, 0:3].copy(), tab[:, 3:7].copy()                                        # next round: from this solution
    h.close()


@pytest.mark.parametrize("sampling", [0, 1])
@pytest.mark.parametrize("count", [8, 40, 300])
@pytest.mark.parametrize("npts", [700, 1753, 3000])
def test_sampler_is_honoured_at_every_batch_size(gpu, capi, synth, po, sampling, count, npts):
    """The kernel rule switches gathers with the batch size (lane gather, pair-packed tiles, strips from 32 alignments) and with the
    points per thread (1, 2, 4): at every combination the configured SAMPLER must be the one that runs.  (Round 3's soak found
    bilinear batches of >= 32 alignments on 2 or 4 points per thread being solved by the bicubic strips kernel: 5e-3 from the
    oracle.  The first template argument of the reported kernel is the sampler; the oracle comparison is the proof.)  3 000 points
    run on teams of 4 CUs; from 32 alignments both samplers gather from the strips."""
    H, W = 120, 160
    als = [synth.make_alignment(900 + i, H=H, W=W, N=n, margin=2) for i, n in enumerate((npts, npts - 37, 130, npts // 2))]
    # start away from the identity: there every point sits exactly on a pixel centre, where the bilinear sampler's derivative is
    # one-sided and which side a coordinate falls on is a matter of the last bit (fp32 kernel vs fp64 oracle)
    ps, qs = np.array([1e-3, -2e-3, 5e-4]), synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    for tau in (0.0, 0.02):
        cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, sampling=sampling, max_num_iterations=5, huber_tau=tau)
        h = capi.Handle(cfg, count, npts, H, W)
        for b in range(count):
            h.set_alignment(b, als[b % 4])
        h.prepare_frames(0, count)          # (left to itself the library makes the strip copies when a frame is solved again)
        h.set_states(0, np.stack([ps] * count), np.stack([qs] * count), np.stack([als[b % 4].v0 for b in range(count)]))
        h.optimize_batch(0, 0, count)
        tab = h.results(0, count)
        kern = h.last_launch()["kernel"]
        assert kern.startswith(f"eds_fused6_kernel<{sampling},"), kern
        # the rule of eds_fused_solve: strips from 32 alignments, except where teams of 4 CUs x 512 points (one per lane) run
        expect_strips = count >= 32 and not (1024 < npts <= 2048 and count <= 128)
        assert (h.last_launch()["layout"] == 2) == expect_strips, (kern, h.last_launch()["layout"])
        for i, a in enumerate(als):
            ref = po.Oracle(a, sampling=sampling).pose6_lm(ps, qs, a.v0, iters=5, lambda0=cfg.lambda0, huber_tau=tau)
            for slot in (i, i + 4 * ((count - 1 - i) // 4)):
                assert po.se3_distance(tab[slot, 0:3], tab[slot, 3:7], ref["p"], ref["q"]) <= TOL_POSE, (kern, slot)
                assert tab[slot, 14] == ref["iterations"]
                assert np.array_equal(h.trace(slot)["accepted"], ref["accepted"]), (kern, slot)
        h.close()


@pytest.mark.parametrize("npts", [2049, 3000, 4097, 7000])
def test_wide_team_members_vs_oracle(gpu, capi, synth, po, npts):
    """Above 2 048 points a launch whose members of 1 024 points would be more than one workgroup per CU gives every member 2 048
    points instead (four per lane; eds_fused_solve): half the exchanges per point.  First solve on new frames (tiles), the same
    frames again (strips), with and without the per-point Huber weight, ragged counts (a last member with one point), against the
    oracle and against each other."""
    H, W, count = 240, 320, 140
    als = [synth.make_alignment(7600 + i, H=H, W=W, N=n) for i, n in enumerate((npts, npts - 1, max(2049, npts - 700), npts))]
    ps, qs = np.array([1e-3, -2e-3, 5e-4]), synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    K = 2 if npts <= 4096 else 4
    for tau in (0.0, 0.01):
        cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=6, huber_tau=tau)
        h = capi.Handle(cfg, count, npts, H, W)
        for b in range(count):
            h.set_alignment(b, als[b % 4])
        S = (np.stack([ps] * count), np.stack([qs] * count), np.stack([als[b % 4].v0 for b in range(count)]))
        tabs = []
        for layout, q in ((1, 2 if tau > 0 else 1), (2, 4 if tau > 0 else 3)):
            h.set_states(0, *S); h.optimize_batch(0, 0, count)
            li = h.last_launch()
            assert li["kernel"] == f"eds_fused6_kernel<0, 4, 512, {q}, {K}>" and li["layout"] == layout and li["cus_per_alignment"] == K, li
            tabs.append(h.results(0, count).copy())
        assert np.abs(tabs[0][:, :7] - tabs[1][:, :7]).max() < 1e-6
        for i, a in enumerate(als):
            ref = po.Oracle(a).pose6_lm(ps, qs, a.v0, iters=6, lambda0=cfg.lambda0, huber_tau=tau)
            for slot in (i, i + 4 * ((count - 1 - i) // 4)):
                assert po.se3_distance(tabs[1][slot, 0:3], tabs[1][slot, 3:7], ref["p"], ref["q"]) <= TOL_POSE, (npts, slot)
                assert tabs[1][slot, 14] == ref["iterations"] and np.array_equal(h.trace(slot)["accepted"], ref["accepted"])
            er = po.Oracle(a).pose6_eval(tabs[1][i, 0:3], tabs[1][i, 3:7], a.v0)["r"]
            r = h.residuals(i)
            assert r.shape == (a.N,) and np.abs(r - er).max() <= 1e-5 * np.abs(er).max()
        h.close()


def test_new_keyframe_invalidates_device_residuals(gpu, capi, synth):
    """after a device-mode solve, set_keyframe must not leave 'residuals still in HBM' set — get_residuals /
    loss_param before the next optimize then report EDS_ERR_STATE instead of the previous keyframe's plane."""
    a = synth.make_alignment(11, H=120, W=160, N=300)
    b = synth.make_alignment(12, H=120, W=160, N=300)
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=5), 1, 300, 120, 160)
    h.set_alignment(0, a)
    h.optimize(0)
    assert h.residuals(0).shape == (300,)
    h.optimize(0)                                   # residuals of this solve stay on the device
    h.set_keyframe(0, b.norm_coord, b.grad, b.idp, b.weights, b.fx, b.fy, b.cx, b.cy)
    with pytest.raises(capi.EdsError) as ei:
        h.residuals(0)
    assert ei.value.code == capi.ERR_STATE
    with pytest.raises(capi.EdsError):
        h.loss_param(0, capi.LP_MAD)
    with pytest.raises(capi.EdsError):
        h.loss_param_batch(capi.LP_MAD, 0, 1)
    h.close()


@pytest.mark.parametrize("team", [2, 4])
def test_team_kernel_vs_oracle_and_single_cu(gpu, capi, synth, po, monkeypatch, team):
    """Several CUs per alignment (eds_fused6_kernel TEAM = K: points split K-way, partial sums exchanged through tagged granules,
    every member running the solver on identical totals) against the oracle and against the one-CU kernel, on ragged point counts
    (members with few or no points), both samplers, with and without the per-point Huber weight."""
    als = [synth.make_alignment(7100 + b, H=240, W=320, N=n) for b, n in enumerate((513, 700, 1024, 1025, 1999, 2048))]
    ps, qs = np.array([1e-3, -2e-3, 5e-4]), synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    P0, Q0, V0 = np.stack([ps] * len(als)), np.stack([qs] * len(als)), np.stack([a.v0 for a in als])
    for sampling in (0, 1):
        for tau in (0.0, 0.01):
            res = {}
            for k in (1, team):
                monkeypatch.setenv("EDS_LM6_TEAM", str(k))
                h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, sampling=sampling, max_num_iterations=10,
                                                    huber_tau=tau), len(als), 2048, 240, 320)
                for b, a in enumerate(als):
                    h.set_alignment(b, a)
                h.set_states(0, P0, Q0, V0)
                h.optimize_batch(0, 0, len(als))
                res[k] = (h.results(0, len(als)), [h.residuals(b) for b in range(len(als))], [h.trace(b) for b in range(len(als))])
                h.close()
            (t1, r1, tr1), (tk, rk, trk) = res[1], res[team]
            for b, a in enumerate(als):
                ref = po.Oracle(a, sampling=sampling).pose6_lm(ps, qs, a.v0, iters=10, lambda0=0.01, huber_tau=tau)
                assert tk[b, 15] == 1.0 and tk[b, 14] == 10
                assert np.array_equal(trk[b]["accepted"], ref["accepted"]) and np.array_equal(trk[b]["accepted"], tr1[b]["accepted"])
                assert po.se3_distance(tk[b, 0:3], tk[b, 3:7], ref["p"], ref["q"]) <= TOL_POSE
                assert po.se3_distance(tk[b, 0:3], tk[b, 3:7], t1[b, 0:3], t1[b, 3:7]) <= 1e-6
                for k in range(len(trk[b]["increments"])):
                    assert np.abs(trk[b]["increments"][k] - tr1[b]["increments"][k]).max() <= 1e-4 * max(np.abs(tr1[b]["increments"][k]).max(), 1e-3)
                er = po.Oracle(a, sampling=sampling).pose6_eval(tk[b, 0:3], tk[b, 3:7], a.v0)["r"]
                assert rk[b].shape == (a.N,) and np.abs(rk[b] - er).max() <= 1e-5 * np.abs(er).max()


@pytest.mark.parametrize("team,sizes", [(4, (2049, 3000, 4096)), (8, (4097, 7000, 8000)), (16, (8193, 12345, 16000))])
def test_large_point_sets_run_on_teams_of_1024_points(gpu, capi, synth, po, monkeypatch, team, sizes):
    """More than 2 048 points (configs[2], the finer levels of configs[3]): with few alignments per launch optimize picks 4, 8 or 16
    CUs of 1 024 points each instead of one CU streaming them all.  The rule's own choice (no override) must agree with the oracle
    and with the one-CU streaming kernel (EDS_LM6_TEAM=1), on ragged counts (last member nearly empty or full), with and without
    the per-point Huber weight; residuals arrive through the pinned mirror."""
    als = [synth.make_alignment(7400 + b, H=480, W=640, N=n) for b, n in enumerate(sizes)]
    ps, qs = np.array([1e-3, -2e-3, 5e-4]), synth.quat_from_axis_angle([0.3, -0.5, 0.8], 2e-3)
    for tau in (0.0, 0.01):
        res = {}
        for k in (1, 0):
            if k: monkeypatch.setenv("EDS_LM6_TEAM", str(k))
            else: monkeypatch.delenv("EDS_LM6_TEAM", raising=False)
            h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=10, huber_tau=tau),
                            len(als), max(sizes), 480, 640)
            for b, a in enumerate(als):
                h.set_alignment(b, a)
            out = []
            for b, a in enumerate(als):          # one alignment per launch: the latency regime
                h.set_state(b, ps, qs, a.v0)
                h.optimize_batch(0, b, 1)
                out.append((h.results(b, 1)[0], h.residuals(b), h.trace(b)))
            res[k] = out
            h.close()
        for b, a in enumerate(als):
            (t1, r1, tr1), (tk, rk, trk) = res[1][b], res[0][b]
            ref = po.Oracle(a).pose6_lm(ps, qs, a.v0, iters=10, lambda0=0.01, huber_tau=tau)
            assert tk[15] == 1.0 and tk[14] == 10
            assert np.array_equal(trk["accepted"], ref["accepted"]) and np.array_equal(trk["accepted"], tr1["accepted"])
            assert po.se3_distance(tk[0:3], tk[3:7], ref["p"], ref["q"]) <= TOL_POSE
            assert po.se3_distance(tk[0:3], tk[3:7], t1[0:3], t1[3:7]) <= 1e-6
            er = po.Oracle(a).pose6_eval(tk[0:3], tk[3:7], a.v0)["r"]
            assert rk.shape == (a.N,) and np.abs(rk - er).max() <= 1e-5 * np.abs(er).max()
            assert np.abs(r1 - er).max() <= 1e-5 * np.abs(er).max()


def test_team_launches_back_to_back_and_in_sub_ranges(gpu, capi, synth, po, monkeypatch):
    """Granule tags carry the launch number: 40 team launches in a row on one handle (different sub-ranges, so a slot's mailbox
    is re-used by other alignments) keep returning the single-CU result."""
    monkeypatch.setenv("EDS_LM6_TEAM", "4")
    als = [synth.make_alignment(7200 + b, H=240, W=320, N=1500 + 100 * b) for b in range(5)]
    h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=8), len(als), 2048, 240, 320)
    for b, a in enumerate(als):
        h.set_alignment(b, a)
    refs = [po.Oracle(a).pose6_lm(a.p0, a.q0, a.v0, iters=8, lambda0=0.01) for a in als]
    rng = np.random.default_rng(3)
    for rep in range(40):
        first = int(rng.integers(0, len(als)))
        count = int(rng.integers(1, len(als) - first + 1))
        h.set_states(0, np.stack([a.p0 for a in als]), np.stack([a.q0 for a in als]), np.stack([a.v0 for a in als]))
        h.optimize_batch(0, first, count)
        tab = h.results(0, len(als))
        for b in range(first, first + count):
            assert po.se3_distance(tab[b, 0:3], tab[b, 3:7], refs[b]["p"], refs[b]["q"]) <= TOL_POSE, (rep, b)
            assert np.array_equal(h.trace(b)["accepted"], refs[b]["accepted"])
    h.close()


@pytest.mark.parametrize("team", [2, 4])
def test_ref12_team_kernel_vs_oracle(gpu, capi, synth, po, monkeypatch, team):
    """The reference problem with K CUs per alignment (eds_fused12_kernel TEAM = K: contiguous point slices, the per-block sums
    exchanged through tagged granules, every member running the LM state machine on identical totals) against the oracle's
    Ceres-LM restatement: same iteration / successful-step counts and termination, pose and velocity within tolerance."""
    monkeypatch.setenv("EDS_REF12_TEAM", str(team))
    als = [synth.make_alignment(7300 + b, H=240, W=320, N=n, start="ctor") for b, n in enumerate((600, 1024, 1500, 2000))]
    # (bicubic only: from the identity start the projections sit on pixel centres, where the bilinear gradient is discontinuous and fp32 /
    # fp64 trajectories part ways — tests/test_parity_gpu.py covers bilinear from a generic start)
    for nb, loss, sampling in ((1, 0, 0), (4, 1, 0), (3, 2, 0)):
        cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb, loss_type=loss,
                                  loss_param=0.3, sampling=sampling)
        h = capi.Handle(cfg, len(als), 2048, 240, 320)
        for b, a in enumerate(als):
            h.set_alignment(b, a)
        h.optimize_batch(0, 0, len(als))
        tab = h.results(0, len(als))
        for b, a in enumerate(als):
            ref = po.Oracle(a, num_blocks=nb, loss_type=loss, loss_param=0.3, max_num_iterations=10, sampling=sampling).solve_lm(a.p0, a.q0, a.v0)
            info = h.info(b)
            assert info["num_iterations"] == ref["num_iterations"] and info["num_successful_steps"] == ref["num_successful_steps"], (nb, b)
            assert info["termination"] == ref["termination"]
            assert po.se3_distance(tab[b, 0:3], tab[b, 3:7], ref["p"], ref["q"]) <= TOL_POSE
            assert np.abs(tab[b, 7:13] - ref["v"]).max() <= 1e-4
            assert tab[b, 13] == pytest.approx(ref["final_cost"], rel=1e-5)
            r = h.residuals(b)
            e = po.Oracle(a, num_blocks=nb, sampling=sampling).eval12(ref["p"], ref["q"], ref["v"], jac=False)["r_raw"]
            assert np.abs(r - e).max() <= 2e-5 * np.abs(e).max()
        h.close()


@pytest.mark.parametrize("n_points,nb,loss", [(5000, 1, 0), (8000, 4, 1), (12345, 1, 0), (16000, 3, 2)])
def test_ref12_large_point_sets_run_on_teams_of_8_and_16(gpu, capi, synth, po, monkeypatch, n_points, nb, loss):
    """The reference problem on more than 4 096 / 8 192 points in the latency regime: optimize picks 8 / 16 CUs per alignment
    (no override).  Iteration counts, termination, pose, velocity, cost and residuals against the oracle's Ceres-LM restatement, and
    the same solve on one CU (EDS_REF12_TEAM=1)."""
    a = synth.make_alignment(7500 + n_points, H=480, W=640, N=n_points, start="ctor")
    cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb, loss_type=loss, loss_param=0.3)
    out = {}
    for k in (0, 1):
        if k: monkeypatch.setenv("EDS_REF12_TEAM", str(k))
        else: monkeypatch.delenv("EDS_REF12_TEAM", raising=False)
        h = capi.Handle(cfg, 1, n_points, 480, 640)
        h.set_alignment(0, a)
        h.optimize_batch(0, 0, 1)
        out[k] = (h.results(0, 1)[0], h.info(0), h.residuals(0))
        h.close()
    ref = po.Oracle(a, num_blocks=nb, loss_type=loss, loss_param=0.3, max_num_iterations=10).solve_lm(a.p0, a.q0, a.v0)
    for k in (0, 1):
        tab, info, r = out[k]
        assert info["num_iterations"] == ref["num_iterations"] and info["num_successful_steps"] == ref["num_successful_steps"], k
        assert info["termination"] == ref["termination"]
        assert po.se3_distance(tab[0:3], tab[3:7], ref["p"], ref["q"]) <= TOL_POSE
        assert np.abs(tab[7:13] - ref["v"]).max() <= 1e-4
        assert tab[13] == pytest.approx(ref["final_cost"], rel=1e-5)
        e = po.Oracle(a, num_blocks=nb).eval12(ref["p"], ref["q"], ref["v"], jac=False)["r_raw"]
        assert np.abs(r - e).max() <= 2e-5 * np.abs(e).max()


def test_shared_event_frames(gpu, capi, synth, po):
    """eds_trk_share_event_frame: several alignments sample ONE slot's frame storage.  Same results, bit for bit, as with a copy of the
    frame in every slot (LM6 and the reference problem, batched and one at a time); a frame written into the source is seen by the
    sharers, a frame written into a sharer ends the sharing; a source must have a frame and must not share itself."""
    H, W, N = 240, 320, 1200
    als = [synth.make_alignment(7600 + k, H=H, W=W, N=N) for k in range(3)]
    B = 9
    for solver in (capi.SOLVER_LM6, capi.SOLVER_REF12):
        cfg = capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=6)
        tabs = {}
        for mode in ("copies", "shared"):
            h = capi.Handle(cfg, B, N, H, W)
            for b in range(B):
                a = als[b % 3]
                h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
                if mode == "copies" or b < 3:
                    h.set_event_frame(b, a.frame)
                else:
                    h.share_event_frame(b, b % 3)
                h.set_state(b, a.p0, a.q0, a.v0)
            h.optimize_batch(0, 0, B)
            tabs[mode] = h.results(0, B).copy()
            if mode == "shared":
                if solver == capi.SOLVER_LM6:
                    assert np.array_equal(tabs["shared"], tabs["copies"])              # no atomics on this path: bit-identical
                else:
                    assert np.abs(tabs["shared"][:, :13] - tabs["copies"][:, :13]).max() <= 1e-9 and np.array_equal(tabs["shared"][:, 14:], tabs["copies"][:, 14:])
                # one at a time (teams) as well
                a = als[1]
                h.set_state(7, a.p0, a.q0, a.v0)
                h.optimize_batch(0, 7, 1)
                assert po.se3_distance(h.results(7, 1)[0, 0:3], h.results(7, 1)[0, 3:7], tabs["copies"][7, 0:3], tabs["copies"][7, 3:7]) <= 1e-9
                assert np.array_equal(h.get_event_frame(7), h.get_event_frame(1))     # a sharer reports the frame it samples
                # a new frame in the SOURCE slot 1 is what its sharers 4 and 7 solve against from now on
                h.set_event_frame(1, als[2].frame)
                g = capi.Handle(cfg, 1, N, H, W)
                g.set_keyframe(0, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
                g.set_event_frame(0, als[2].frame)
                g.set_state(0, a.p0, a.q0, a.v0); g.optimize_batch(0, 0, 1)
                h.set_state(4, a.p0, a.q0, a.v0); h.optimize_batch(0, 4, 1)
                assert po.se3_distance(h.results(4, 1)[0, 0:3], h.results(4, 1)[0, 3:7], g.results(0, 1)[0, 0:3], g.results(0, 1)[0, 3:7]) <= 1e-9
                # a frame written INTO sharer 7 ends its sharing; 4 still follows slot 1
                h.set_event_frame(7, als[1].frame)
                h.set_state(7, a.p0, a.q0, a.v0); h.optimize_batch(0, 7, 1)
                assert po.se3_distance(h.results(7, 1)[0, 0:3], h.results(7, 1)[0, 3:7], tabs["copies"][7, 0:3], tabs["copies"][7, 3:7]) <= 1e-9
                g.close()
                with pytest.raises(capi.EdsError):
                    h.share_event_frame(8, 4)                                         # slot 4 shares slot 1's frame itself
                e = capi.Handle(cfg, 2, N, H, W)
                with pytest.raises(capi.EdsError):
                    e.share_event_frame(1, 0)                                         # slot 0 has no frame yet
                e.close()
            h.close()


def test_large_point_sets_in_large_batches_go_out_in_several_team_launches(gpu, capi, synth, po, monkeypatch):
    """More than 2 048 points at ANY batch size run on teams; a range that needs more workgroups than the mailboxes hold (4 096) is cut
    into several launches.  1 100 alignments of 2 100 points (4 CUs each: 4 400 workgroups, two launches): every result equals the
    one-CU streaming kernel's, a few are checked against the oracle."""
    H, W, N, B = 120, 160, 2100, 1100
    als = [synth.make_alignment(7700 + k, H=H, W=W, N=N) for k in range(5)]
    res = {}
    for k in (0, 1):
        if k: monkeypatch.setenv("EDS_LM6_TEAM", "1")
        else: monkeypatch.delenv("EDS_LM6_TEAM", raising=False)
        h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=5), B, N, H, W)
        for b in range(B):
            a = als[b % 5]
            h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
            if b < 5: h.set_event_frame(b, a.frame)
            else: h.share_event_frame(b, b % 5)
        h.set_states(0, np.stack([als[b % 5].p0 for b in range(B)]), np.stack([als[b % 5].q0 for b in range(B)]), np.stack([als[b % 5].v0 for b in range(B)]))
        h.optimize_batch(0, 0, B)
        res[k] = h.results(0, B).copy()
        h.close()
    assert (res[0][:, 15] == 1.0).all() and (res[0][:, 14] == 5).all()
    for b in range(B):
        assert po.se3_distance(res[0][b, 0:3], res[0][b, 3:7], res[1][b, 0:3], res[1][b, 3:7]) <= 1e-6, b
    for b in (0, 1023, 1024, 1099):                      # either side of the launch boundary
        a = als[b % 5]
        ref = po.Oracle(a).pose6_lm(a.p0, a.q0, a.v0, iters=5, lambda0=0.01)
        assert po.se3_distance(res[0][b, 0:3], res[0][b, 3:7], ref["p"], ref["q"]) <= TOL_POSE


def test_idepth_straight_from_the_depth_table(gpu, capi, synth):
    """eds_trk_set_idepth_strided: the inverse depths as column 0 of DepthPoints' N x 4 table [mu, sigma^2, a, b] — the same solve, bit for
    bit, as after eds_trk_set_idepth with a copy of that column."""
    a = synth.make_alignment(7800, H=120, W=160, N=900)
    rng = np.random.default_rng(1)
    table = np.column_stack([a.idp * rng.uniform(0.9, 1.1, a.N), rng.uniform(0, 1, a.N), rng.uniform(0, 1, a.N), rng.uniform(0, 1, a.N)])
    tabs = []
    for strided in (False, True):
        h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=5), 1, a.N, a.H, a.W)
        h.set_alignment(0, a)
        if strided: h.set_idepth_strided(0, table)
        else: h.set_idepth(0, np.ascontiguousarray(table[:, 0]))
        h.set_state(0, a.p0, a.q0, a.v0)
        h.optimize_batch(0, 0, 1)
        tabs.append(h.results(0, 1).copy())
        h.close()
    assert np.array_equal(tabs[0], tabs[1]) and tabs[0][0, 15] == 1.0


def test_ref12_new_frame_batch_shapes_at_their_operating_point_vs_oracle(gpu, capi, synth, po, als64):
    """Round 6: from 1 024 alignments on, the reference problem (one residual block, <= 2 000 points) on frames that are NEW for the solve
    launches the paired shape with 736 cache slots per alignment — eds_fused12_kernel<0, 256, 736, false, 1, 1>, quad gather on the tiles
    (csrc/eds_launch_rule.hpp).  1 024 alignments, 8 distinct, every slot its own frame: the rule's own choice by name, its rows against the
    oracle's Ceres-LM restatement (step accounting, termination, pose, velocity), and the other three batch shapes (EDS_REF12_KERNEL =
    paired | wide | full: no cache / 1 408 slots / a slot for every point, one alignment per CU) against it — the same solve to the last
    digits of the fp64 sums."""
    B, D = 1024, 8
    cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=1)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als64[:D]]
    for b in range(B):
        a = als64[b % D]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        h.set_event_frame(b, fr[b % D])
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")               # every solve as a frame's first solve
    S0 = (np.stack([als64[b % D].p0 for b in range(B)]), np.stack([als64[b % D].q0 for b in range(B)]), np.stack([als64[b % D].v0 for b in range(B)]))
    refs = []
    for d in range(D):
        a = als64[d]
        a32 = synth.Alignment(**{**a.__dict__, "frame": fr[d].astype(np.float64)})
        refs.append(po.Oracle(a32, num_blocks=1, max_num_iterations=10).solve_lm(a.p0, a.q0, a.v0))
    tabs = {}
    for shape, kernel in ((None, "eds_fused12_kernel<0, 256, 736, false, 1, 1>"), ("paired", "eds_fused12_kernel<0, 256, 320, false, 1, 1>"),
                          ("wide", "eds_fused12_kernel<0, 512, 1408, false, 1, 1>"), ("full", "eds_fused12_kernel<0, 512, 2000, false, 1, 1>")):
        h.set_knob("EDS_REF12_KERNEL", shape)
        h.set_states(0, *S0)
        h.optimize_batch(0, 0, B)
        assert h.last_launch()["kernel"] == kernel, (shape, h.last_launch()["kernel"])
        tab = np.array(h.results(0, B))
        tabs[shape] = tab
        for d in range(D):
            ref = refs[d]
            for slot in (d, d + B - D, d + 504):              # (504 = 63 x 8: the same alignment in the middle of the range)
                info = h.info(slot)
                assert (info["num_iterations"], info["num_successful_steps"], info["termination"]) == (ref["num_iterations"], ref["num_successful_steps"], ref["termination"]), (shape, slot)
                assert bool(info["success"]) == ref["usable"]
                assert po.se3_distance(tab[slot, 0:3], tab[slot, 3:7], ref["p"], ref["q"]) <= TOL_POSE and np.abs(tab[slot, 7:13] - ref["v"]).max() <= 1e-4, (shape, slot)
        er = po.Oracle(synth.Alignment(**{**als64[3].__dict__, "frame": fr[3].astype(np.float64)}), num_blocks=1).eval12(tab[3, 0:3], tab[3, 3:7], tab[3, 7:13], jac=False)["r_raw"]
        assert np.abs(h.residuals(3) - er).max() <= 1e-5 * np.abs(er).max(), shape          # kf->residuals at the solution (Tracker.cpp:223-230)
    for shape in ("paired", "wide", "full"):
        assert np.abs(tabs[shape][:, :13] - tabs[None][:, :13]).max() < 1e-9 and np.array_equal(tabs[shape][:, 14:], tabs[None][:, 14:]), shape
    h.close()
