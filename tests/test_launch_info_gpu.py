"""eds_trk_last_launch / eds_trk_prepare_frames (include/eds_hip.h, round 3): the library says which kernel a solve launched — bench.py
prices that kernel instead of mirroring the selection rule — and the strip copies of the frames (csrc/eds_layout.hpp) follow every
change of a frame without the caller doing anything."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(capi, synth, B, N=2000, H=240, W=320, distinct=8, prepare=True, **cfg):
    als = [synth.make_alignment(8100 + i, H=H, W=W, N=N) for i in range(distinct)]
    h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, max_num_iterations=6, **cfg), B, N, H, W)
    for b in range(B):
        h.set_alignment(b, als[b % distinct])
    if prepare:
        h.prepare_frames(0, B)          # the strip copies up front (left to itself the library makes them when a frame is solved AGAIN)
    P = np.stack([als[b % distinct].p0 for b in range(B)]); Q = np.stack([als[b % distinct].q0 for b in range(B)]); V = np.stack([als[b % distinct].v0 for b in range(B)])
    return h, als, (P, Q, V)


def test_last_launch_names_the_kernel_that_ran(gpu, capi, synth):
    h, als, (P, Q, V) = _batch(capi, synth, 200, solver=capi.SOLVER_LM6)
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, 200)
    li = h.last_launch()
    assert li["kernel"] == "eds_fused6_kernel<0, 4, 512, 3, 1>" and li["layout"] == 2           # strips, LDS landing zone, one CU per alignment
    assert (li["workgroups"], li["cus_per_alignment"], li["first"], li["count"], li["timing_source"]) == (200, 1, 0, 200, 0)
    assert li["span_us"] > 0 and 0.0 < li["covered"] <= 1.0 and li["mean_workgroup_us"] <= li["span_us"] and li["tail_idle_us"] >= 0
    # a handful of alignments: several CUs each, timed from the kernels' own stamps, no digest
    h.set_states(0, P[:8], Q[:8], V[:8]); h.optimize_batch(0, 0, 8)
    li = h.last_launch()
    # (round 5: teams of four in four candidate groups — 16 CUs per alignment, eds_launch_rule.hpp)
    assert li["cus_per_alignment"] == 16 and li["workgroups"] == 128 and li["timing_source"] == 1 and li["span_us"] == 0.0
    assert li["kernel"].startswith("eds_fused6_kernel<0, 1, 512,") and li["kernel"].endswith(", 4, 4>")
    h.set_knob("EDS_LM6_GROUPS", "1")
    h.set_states(0, P[:8], Q[:8], V[:8]); h.optimize_batch(0, 0, 8)
    li = h.last_launch()
    assert li["cus_per_alignment"] == 4 and li["workgroups"] == 32 and li["kernel"].endswith(", 4>") and not li["kernel"].endswith(", 4, 4>")
    h.set_knob("EDS_LM6_GROUPS", None)
    # the reference problem, and the per-point Huber variant of the pose-only kernel
    h.set_config(capi.default_config(exec=capi.EXEC_DEVICE, max_num_iterations=6, solver=capi.SOLVER_REF12))
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, 200)
    assert h.last_launch()["kernel"] == "eds_fused12_kernel<0, 512, 1408, false, 1, 2>" and h.last_launch()["layout"] == 2
    h.set_config(capi.default_config(exec=capi.EXEC_DEVICE, max_num_iterations=6, solver=capi.SOLVER_LM6, huber_tau=0.01))
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, 200)
    assert h.last_launch()["kernel"] == "eds_fused6_kernel<0, 4, 512, 4, 1>"
    h.close()


def test_strip_copies_follow_the_frames(gpu, capi, synth):
    """A frame that changes between two solves (set_event_frame, a device-built frame, a shared frame) must be sampled as it is NOW:
    every slot's result equals what a fresh handle gives for the same inputs, bit for bit."""
    B = 64
    h, als, (P, Q, V) = _batch(capi, synth, B, solver=capi.SOLVER_LM6)
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)
    first = h.results(0, B).copy()
    assert h.prepare_frames(0, B) >= 0.0                                          # nothing stale: a no-op
    assert h.prepare_frames(0, B, force=True) > 0.0                               # measured conversion of all 64 frames
    # slot 5 gets slot 6's frame, slot 9 shares slot 2's storage
    h.set_event_frame(5, np.ascontiguousarray(als[6 % 8].frame, dtype=np.float32))
    h.share_event_frame(9, 2)
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)
    second = h.results(0, B).copy()
    same = [b for b in range(B) if b not in (5, 9)]
    assert np.array_equal(second[same], first[same])
    assert not np.array_equal(second[5], first[5]) and not np.array_equal(second[9], first[9])
    fresh = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, max_num_iterations=6, solver=capi.SOLVER_LM6), B, 2000, 240, 320)
    for b in range(B):
        a = als[b % 8]
        fresh.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        fresh.set_event_frame(b, np.ascontiguousarray(als[{5: 6, 9: 2}.get(b, b) % 8].frame, dtype=np.float32))
    fresh.prepare_frames(0, B)
    fresh.set_states(0, P, Q, V); fresh.optimize_batch(0, 0, B)
    assert np.array_equal(fresh.results(0, B), second)
    # and the tile kernels (the handle's EDS_FUSED_LAYOUT knob) agree with the strip kernels to the last bits of the fp32 sums
    fresh.set_knob("EDS_FUSED_LAYOUT", "tiles")
    fresh.set_states(0, P, Q, V); fresh.optimize_batch(0, 0, B)
    assert fresh.last_launch()["layout"] == 1
    tiles = fresh.results(0, B)
    fresh.set_knob("EDS_FUSED_LAYOUT", None)
    fresh.set_states(0, P, Q, V); fresh.optimize_batch(0, 0, B)
    assert np.array_equal(fresh.results(0, B), second)          # (64 x 2 000 points run on teams of 512 points per member: the tiles either way)
    assert np.abs(tiles[:, :7] - second[:, :7]).max() < 1e-6 and np.array_equal(tiles[:, 14], second[:, 14])
    h.close(); fresh.close()


def test_strip_copies_are_made_for_frames_that_are_solved_again(gpu, capi, synth):
    """The copies cost more than one solve gains from them (eds_strips.hip): the first solve on new frames samples the tiles, the second
    solve on the SAME frames makes the copies and uses them; a few new frames among many kept ones are converted at once, many new
    frames send the launch back to the tiles.  The results of the two layouts agree to the last bits of the fp32 sums."""
    B = 200
    h, als, (P, Q, V) = _batch(capi, synth, B, prepare=False, solver=capi.SOLVER_LM6)
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)
    assert h.last_launch()["layout"] == 1 and h.last_launch()["kernel"] == "eds_fused6_kernel<0, 4, 512, 1, 1>"     # tiles, pair-packed quad gather
    t1 = h.results(0, B).copy()
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)
    assert h.last_launch()["layout"] == 2 and h.last_launch()["kernel"] == "eds_fused6_kernel<0, 4, 512, 3, 1>"
    t2 = h.results(0, B).copy()
    assert np.abs(t1[:, :7] - t2[:, :7]).max() < 1e-6 and np.array_equal(t1[:, 14], t2[:, 14])
    f32 = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in (3, 17):                                                              # 2 of 200 frames are new: converted, the launch stays on strips
        h.set_event_frame(b, f32[(b + 1) % 8])
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)
    assert h.last_launch()["layout"] == 2
    t3 = h.results(0, B).copy()
    keep = [b for b in range(B) if b not in (3, 17)]
    assert np.array_equal(t3[keep], t2[keep]) and not np.array_equal(t3[3], t2[3])
    for b in range(0, B, 2):                                                       # half of them new: this launch samples the tiles
        h.set_event_frame(b, f32[b % 8])
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)
    assert h.last_launch()["layout"] == 1
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)                            # ... and the next one has its copies
    assert h.last_launch()["layout"] == 2
    odd = list(range(1, B, 2))
    assert np.array_equal(h.results(0, B)[[b for b in odd if b not in (3, 17)]], t2[[b for b in odd if b not in (3, 17)]])
    h.close()


@pytest.mark.parametrize("shape", [(480, 640), (45, 70), (33, 129), (720, 1280), (64, 64), (600, 37)])
def test_host_frame_upload_in_one_launch(gpu, capi, shape):
    """set_event_frame with a host frame (fp64 and fp32): ONE launch follows the host through the staging buffer (bands of 32 k rows,
    progress published in a pinned word).  What arrives must be the fp32-rounded frame, on sizes whose width is not a multiple of 4,
    whose height is not a multiple of the band, and on several frames in a row through the same staging buffer (the next frame must
    never show rows of the previous one); the margin replicates the border (sampled through the bilinear clamp in other tests)."""
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE), 2, 64, H, W)
    for rep in range(4):
        f64 = rng.standard_normal((H, W)) * (10.0 ** rng.integers(-3, 2))
        h.set_event_frame(rep & 1, f64)
        got = h.get_event_frame(rep & 1)
        assert np.array_equal(got, f64.astype(np.float32).astype(np.float64)), (shape, rep)
        f32 = rng.standard_normal((H, W)).astype(np.float32)
        h.set_event_frame(rep & 1, f32)
        assert np.array_equal(h.get_event_frame(rep & 1), f32.astype(np.float64)), (shape, rep)
    h.close()


def test_live_sequence_timed_inside_the_library(gpu, capi, synth, po):
    """eds_trk_bench_live: the live sequence looped inside one C call must leave the slot where a plain sequence of the same calls
    leaves it, and report plausible medians."""
    al = synth.make_alignment(5003, H=240, W=320, N=1500)
    cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=8, num_blocks=2)
    h = capi.Handle(cfg, 1, al.N, al.H, al.W)
    h.set_alignment(0, al)
    h.set_idepth(0, al.idp); h.set_event_frame(0, al.frame)          # the same calls by hand first
    p, q, v, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v0)
    r0, tau0 = h.residuals_and_loss(0, capi.LP_MAD)
    t = h.bench_live(0, al.p0, al.q0, al.v0, idp=al.idp, frame=al.frame, method=capi.LP_MAD, reps=5)
    tab = h.results(0, 1)[0]
    # (REF12 adds its wavefronts' tiles into the LDS sums with fp64 atomics: the order, and with it the last bits, vary from run to run)
    assert np.abs(tab[0:3] - p).max() < 1e-12 and np.abs(tab[3:7] - q).max() < 1e-12 and np.abs(tab[7:13] - v).max() < 1e-12
    assert 0 < t["kernel_us"] < t["optimize_us"] < t["total_us"] < 1e5
    assert abs(t["total_us"] - (t["set_idepth_us"] + t["set_event_frame_us"] + t["optimize_us"] + t["residuals_and_loss_us"])) < 0.5 * t["total_us"]
    with pytest.raises(capi.EdsError):
        h.bench_live(3, al.p0, al.q0, al.v0)
    h.close()


def test_two_handles_with_different_knobs_coexist(gpu, capi, synth, monkeypatch):
    """Tuning knobs belong to the handle (eds_trk_set_knob; the environment is read once, at eds_trk_create): two handles of one process
    keep different settings, changing the environment after creation changes nothing, unknown names are refused."""
    import os
    B = 40
    als = [synth.make_alignment(8100 + k, H=120, W=160, N=700) for k in range(4)]
    monkeypatch.setenv("EDS_LM6_TEAM", "1")
    monkeypatch.setenv("EDS_FUSED_GATHER", "lane")
    ha = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=6), B, 700, 120, 160)     # reads lane + no teams
    monkeypatch.delenv("EDS_FUSED_GATHER")
    monkeypatch.delenv("EDS_LM6_TEAM")
    hb = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=6), B, 700, 120, 160)     # the rule
    hb.set_knob("EDS_STRIPS_POLICY", "eager")
    for h in (ha, hb):
        for b in range(B):
            h.set_alignment(b, als[b % 4])
    S = (np.stack([als[b % 4].p0 for b in range(B)]), np.stack([als[b % 4].q0 for b in range(B)]), np.stack([als[b % 4].v0 for b in range(B)]))
    os.environ["EDS_FUSED_LAYOUT"] = "tiles"             # after creation: nobody reads it
    try:
        for rep in range(2):
            for h in (ha, hb):
                h.set_states(0, *S); h.optimize_batch(0, 0, B)
            la, lb = ha.last_launch(), hb.last_launch()
            assert la["kernel"] == "eds_fused6_kernel<0, 2, 512, 0, 1>" and la["layout"] == 1, la      # lane gather: never strips
            assert lb["kernel"] == "eds_fused6_kernel<0, 2, 512, 3, 1>" and lb["layout"] == 2, lb      # eager: strips from the first solve
            assert np.abs(ha.results(0, B)[:, :7] - hb.results(0, B)[:, :7]).max() < 1e-6
    finally:
        os.environ.pop("EDS_FUSED_LAYOUT", None)
    assert ha.strips_info()["bytes"] == 0 and hb.strips_info()["bytes"] > 0 and hb.strips_info()["row_phases"] == 4
    with pytest.raises(capi.EdsError):
        ha.set_knob("EDS_NO_SUCH_KNOB", "1")
    with pytest.raises(capi.EdsError):
        ha.set_knob("EDS_FRAME_LAYOUT", "rowmajor")
    # a budget the copies do not fit: refused once, remembered (no allocation attempt per solve), the tiles serve; a new budget re-arms
    hc = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=6), B, 700, 120, 160)
    for b in range(B):
        hc.set_alignment(b, als[b % 4])
    hc.set_knob("EDS_STRIPS_POLICY", "eager")
    hc.set_knob("EDS_STRIPS_BUDGET_PCT", "1")
    hc.set_states(0, *S); hc.optimize_batch(0, 0, B)
    info = hc.strips_info()
    if info["unavailable"]:                               # (only on a box whose free memory makes 1 % too little: not the case on 288 GB)
        assert hc.last_launch()["layout"] == 1
    else:
        assert info["bytes"] > 0 and hc.last_launch()["layout"] == 2
    assert np.abs(hc.results(0, B)[:, :7] - hb.results(0, B)[:, :7]).max() < 1e-6
    ha.close(); hb.close(); hc.close()
