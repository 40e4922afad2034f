"""include/eds_hip.h: "a handle owns one HIP stream and is not re-entrant; distinct handles may be used concurrently".  Four host threads,
each with its own handle (ctypes drops the GIL inside every call), run whole tracking steps at the same time — events -> frame, solve,
loss scale, getCoord — and every thread must get exactly what the same sequence gives when it runs alone (LM6: bit for bit)."""
import importlib
import threading

import numpy as np
import pytest

capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")

pytestmark = pytest.mark.gpu


def _sequence(seed, reps, out, barrier=None):
    try:
        al = synth.make_alignment(8200 + seed, H=240, W=320, N=1500 + 100 * seed)
        rng = np.random.default_rng(seed)
        fr = al.frame
        strong = np.argwhere(np.abs(fr) > 0.2 * np.abs(fr).max())
        pk = strong[rng.integers(0, len(strong), 20_000)]
        ev = (pk[:, 1].astype(np.uint16), pk[:, 0].astype(np.uint16), (fr[pk[:, 0], pk[:, 1]] > 0).astype(np.uint8))
        h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=8), 2, al.N, al.H, al.W)
        h.set_alignment(0, al); h.set_alignment(1, al)
        if barrier is not None:
            barrier.wait()
        res = []
        for k in range(reps):
            norm = h.build_event_frame(1, *ev)
            h.set_event_frame(0, al.frame)
            h.set_state(0, al.p0, al.q0, al.v0); h.set_state(1, al.p0, al.q0, al.v0)
            h.optimize_batch(0, 0, 2)
            tab = h.results(0, 2)
            tau = h.loss_param(0, capi.LP_MAD)
            pts = h.update_points(0, False)
            res.append((norm, tab.copy(), tau, pts["coord"].copy(), h.residuals(0).copy()))
        h.close()
        out[seed] = res
    except BaseException as e:                       # surfaces in the main thread
        out[seed] = e


def test_distinct_handles_from_four_threads(gpu):
    reps = 12
    alone = {}
    for s in range(4):
        _sequence(s, 2, alone)
        assert not isinstance(alone[s], BaseException), alone[s]
    together = {}
    barrier = threading.Barrier(4)
    threads = [threading.Thread(target=_sequence, args=(s, reps, together, barrier)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a thread did not finish"
    for s in range(4):
        assert not isinstance(together[s], BaseException), together[s]
        ref = alone[s][1]
        for k, got in enumerate(together[s]):
            assert got[0] == pytest.approx(ref[0], rel=1e-11)                     # frame norm (fp64 atomics: order varies)
            assert np.array_equal(got[1][0], ref[1][0]), (s, k)                   # slot 0: host-given frame -> bit-identical solve
            np.testing.assert_allclose(got[1][1], ref[1][1], rtol=1e-6, atol=1e-7)     # slot 1: device-built frame (norm in the last bits)
            assert got[2] == ref[2] and np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4]), (s, k)
