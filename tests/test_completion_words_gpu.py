"""Round 6: a launch of up to 64 alignments is waited for through its workgroups' completion words in pinned memory instead of the HIP
runtime (csrc/eds_capi.hip: wait_stream; knob EDS_POLL_RESULTS).  Everything a caller reads right after the call — result table,
eds_trk_info, the residuals (pinned mirror for the first slots, the device plane for the others), the pose-only trace (copied to HBM BEHIND
the kernel's word, read by a null-stream copy) — must be exactly what the stream-waited path returns, call after call, for one
alignment on teams x candidate groups, a handful, and 64 in one launch; ABI 6's eds_trk_optimize_batch_wait and eds_trk_bench_batch
ride the same path."""
import importlib

import numpy as np
import pytest

capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")

pytestmark = pytest.mark.gpu
H, W, N = 240, 320, 1500


def _handle(solver, B, als, poll):
    h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, num_blocks=2 if solver == capi.SOLVER_REF12 else 1,
                                        loss_type=capi.LOSS_HUBER if solver == capi.SOLVER_REF12 else capi.LOSS_NONE, loss_param=0.3, max_num_iterations=8),
                    B, N, H, W)
    for b in range(B):
        h.set_alignment(b, als[b % len(als)])
    h.set_knob("EDS_POLL_RESULTS", "1" if poll else "0")
    return h


@pytest.mark.parametrize("solver", [capi.SOLVER_LM6, capi.SOLVER_REF12])
@pytest.mark.parametrize("B", [1, 5, 64])
def test_words_and_stream_wait_return_the_same(gpu, solver, B):
    als = [synth.make_alignment(9100 + i, H=H, W=W, N=N - 7 * i) for i in range(min(B, 6))]
    P0 = np.stack([als[b % len(als)].p0 for b in range(B)]); Q0 = np.stack([als[b % len(als)].q0 for b in range(B)]); V0 = np.stack([als[b % len(als)].v0 for b in range(B)])
    hw, hs = _handle(solver, B, als, True), _handle(solver, B, als, False)
    slots = sorted({0, B - 1, min(B - 1, 9)})             # (slot 9: beyond the pinned residual mirror's 8 rows)
    for rep in range(12):
        out = []
        for h in (hw, hs):
            h.set_states(0, P0, Q0, V0)
            h.optimize_batch(0, 0, B)                    # ABI 6: eds_trk_optimize_batch_wait
            tab = np.array(h.results(0, B))
            res = [h.residuals(s).copy() for s in slots]
            tr = [h.trace(s) for s in slots] if solver == capi.SOLVER_LM6 else None
            infos = [h.info(s) for s in slots]
            out.append((tab, res, tr, infos))
        (ta, ra, tra, ia), (tb, rb, trb, ib) = out
        assert all(i["success"] and i["flags"] == 0 for i in ia + ib)
        if solver == capi.SOLVER_LM6:
            assert np.array_equal(ta, tb), rep
            for x, y in zip(ra, rb):
                assert np.array_equal(x, y), rep
            for x, y in zip(tra, trb):
                assert np.array_equal(x["accepted"], y["accepted"]) and np.array_equal(x["costs"], y["costs"]) and np.array_equal(x["increments"], y["increments"]), rep
                assert len(x["accepted"]) == 8
        else:                                            # (REF12: fp64 LDS atomics, the last bits vary from run to run)
            np.testing.assert_allclose(ta[:, :13], tb[:, :13], rtol=0, atol=1e-9)
            assert np.array_equal(ta[:, 14:], tb[:, 14:]), rep
            for x, y in zip(ra, rb):
                np.testing.assert_allclose(x, y, rtol=0, atol=1e-9)
        for x, y in zip(ia, ib):
            assert (x["num_iterations"], x["num_successful_steps"], x["termination"]) == (y["num_iterations"], y["num_successful_steps"], y["termination"])
    hw.close(); hs.close()


def test_bench_batch_times_the_step_and_leaves_its_results(gpu):
    B = 16
    als = [synth.make_alignment(9200 + i, H=H, W=W, N=N) for i in range(4)]
    P0 = np.stack([als[b % 4].p0 for b in range(B)]); Q0 = np.stack([als[b % 4].q0 for b in range(B)]); V0 = np.stack([als[b % 4].v0 for b in range(B)])
    h = _handle(capi.SOLVER_LM6, B, als, True)
    h.set_knob("EDS_FUSED_LAYOUT", "tiles")              # (the same kernel for every solve: the second solve of a frame would otherwise switch to its strip copy)
    h.set_states(0, P0, Q0, V0); h.optimize_batch(0, 0, B)
    ref = np.array(h.results(0, B))
    t = h.bench_batch(P0, Q0, V0, reps=30)
    if any(h.info(b)["flags"] for b in range(B)):        # (a team time-out on a busy box: the one-CU re-run sums in another order)
        np.testing.assert_allclose(np.array(h.results(0, B))[:, :13], ref[:, :13], rtol=1e-6, atol=1e-6)
    else:
        assert np.array_equal(np.array(h.results(0, B)), ref)
    assert 0.0 < t["kernel_us"] <= t["solve_us"] <= t["step_us"] <= t["slowest_step_us"] < 1e6 and t["set_states_us"] < t["step_us"]
    with pytest.raises(capi.EdsError):
        h.bench_batch(P0, Q0, V0, reps=0)
    h.close()
