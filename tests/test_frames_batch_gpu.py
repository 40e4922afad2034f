"""eds_trk_set_event_frames (ABI 5): many host frames in one call — narrowed on a few host threads into a ring of pinned staging slots,
stored by one kernel per frame — must leave every slot BIT-IDENTICAL to the one-frame path (eds_trk_set_event_frame: the frames
Tracker::optimize is handed, Tracker.hpp:80-81), whatever the count, the ring size (16 slots), the thread count, the dtype."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dma", [False, True])
@pytest.mark.parametrize("H,W,count,threads", [(120, 160, 40, None), (61, 83, 17, 3), (480, 640, 33, 16), (48, 64, 1, None), (120, 160, 16, 1)])
def test_batch_upload_equals_one_frame_uploads(gpu, capi, H, W, count, threads, dma):
    rng = np.random.default_rng(H * 1000 + count)
    frames = [rng.standard_normal((H, W)) * 1e-2 for _ in range(count)]
    frames[0][0, :] = np.array([1e-300, -1e-300, 1e38, -3.4e38] * (W // 4) + [0.0] * (W % 4))      # denormal / huge values narrow the same way
    B = count + 3
    cfg = capi.default_config(exec=capi.EXEC_DEVICE)
    ha, hb = capi.Handle(cfg, B, 64, H, W), capi.Handle(cfg, B, 64, H, W)
    if threads:
        hb.set_knob("EDS_UPLOAD_THREADS", str(threads))
    hb.set_knob("EDS_UPLOAD_DMA", "1" if dma else "0")
    for dt in (np.float64, np.float32):
        fr = [np.ascontiguousarray(f, dtype=dt) for f in frames]
        for i, f in enumerate(fr):
            ha.set_event_frame(2 + i, f)
        hb.set_event_frames(2, fr)
        for i in range(count):
            a, b = ha.get_event_frame(2 + i), hb.get_event_frame(2 + i)
            assert np.array_equal(a, b), (dt, i)
            assert np.array_equal(b, np.asarray(fr[i], dtype=np.float32).astype(np.float64))
        # again, other frames, straight behind the first batch (the ring is still being read by the last kernels)
        fr2 = [np.ascontiguousarray(f[::-1], dtype=dt) for f in frames]
        hb.set_event_frames(2, fr2)
        assert all(np.array_equal(hb.get_event_frame(2 + i), np.asarray(fr2[i], dtype=np.float32).astype(np.float64)) for i in (0, count // 2, count - 1))
    with pytest.raises(capi.EdsError):
        hb.set_event_frames(B - 1, [frames[0], frames[0]])                      # runs past the last slot
    ha.close(); hb.close()


def test_batch_upload_then_solve_equals_single_uploads(gpu, capi, synth, po):
    als = [synth.make_alignment(300 + i, H=120, W=160, N=400) for i in range(5)]
    cfg = capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=8)
    B = 20
    tabs = []
    for batch in (False, True):
        h = capi.Handle(cfg, B, 400, 120, 160)
        for b in range(B):
            a = als[b % 5]
            h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
            if not batch:
                h.set_event_frame(b, a.frame)
        if batch:
            h.set_event_frame(2, als[2].frame)
            h.share_event_frame(7, 2)                                           # a shared slot gets a frame of its own again
            h.set_event_frames(0, [als[b % 5].frame for b in range(B)])
        h.set_states(0, np.stack([als[b % 5].p0 for b in range(B)]), np.stack([als[b % 5].q0 for b in range(B)]), np.stack([als[b % 5].v0 for b in range(B)]))
        h.optimize_batch(0, 0, B)
        tabs.append(h.results(0, B).copy())
        h.close()
    assert np.array_equal(tabs[0], tabs[1])
    ref = po.Oracle(synth.Alignment(**{**als[3].__dict__, "frame": np.asarray(als[3].frame, dtype=np.float32).astype(np.float64)})).pose6_lm(
        als[3].p0, als[3].q0, als[3].v0, iters=8, lambda0=cfg.lambda0)
    assert po.se3_distance(tabs[1][3, 0:3], tabs[1][3, 3:7], ref["p"], ref["q"]) <= 1e-6
