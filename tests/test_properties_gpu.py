"""Size-independent properties of the solve at BASELINE.json's full batch size (4 096 alignments x 2 000 points on 640x480, one
launch): what must hold whatever the inputs are, checked on every row of the batch — the oracle checks of the other files sample rows.

* replicas: the same alignment in different slots gives the same bits (no cross-slot state; LM6 has no atomics);
* re-solve: restarted from its own result the solver never ends on a higher cost and moves less and less;
* point order: a permutation of the keyframe's points changes only the order of fp32 sums;
* batch / layout: a row of the batch equals the same alignment solved alone (one CU team kernel), on tiles and on strips;
* the reference problem: num_iterations = successful + unsuccessful, final cost <= initial cost, unit quaternion and velocity out.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
B, N, H, W, D = 4096, 2000, 480, 640, 16


@pytest.fixture(scope="module")
def als(synth):
    return [synth.make_alignment(6100 + i, H=H, W=W, N=N) for i in range(D)]


def _load(capi, als, solver, **kw):
    h = capi.Handle(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10, **kw), B, N, H, W)
    fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in range(B):
        a = als[b % D]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy)
        h.set_event_frame(b, fr[b % D])
    S = tuple(np.stack([getattr(als[b % D], k) for b in range(B)]) for k in ("p0", "q0", "v0"))
    return h, S


def test_lm6_full_batch_properties(gpu, capi, synth, po, als):
    h, (P, Q, V) = _load(capi, als, capi.SOLVER_LM6)
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)                 # first solve on new frames: tiles
    t_tiles = h.results(0, B).copy(); assert h.last_launch()["layout"] == 1
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)                 # the same frames again: strips
    t = h.results(0, B).copy(); assert h.last_launch()["layout"] == 2
    assert np.all(t[:, 15] == 1.0) and np.all(t[:, 14] == 10)
    # replicas bit-identical, both layouts; the layouts agree to the last bits of the fp32 sums
    for tab in (t, t_tiles):
        assert all(np.array_equal(tab[b, :14], tab[b % D, :14]) for b in range(D, B))
    assert np.abs(t[:, :7] - t_tiles[:, :7]).max() < 1e-6
    # unit quaternions out
    assert np.abs(np.linalg.norm(t[:, 3:7], axis=1) - 1.0).max() < 1e-12
    # restarted from its own result the solver never ends on a higher cost, and the distance it still moves per restart shrinks
    prev, moved = t, []
    for rnd in range(8):
        h.set_states(0, prev[:, 0:3], prev[:, 3:7], V); h.optimize_batch(0, 0, B)
        cur = h.results(0, B).copy()
        assert np.all(cur[:, 13] <= prev[:, 13] * (1 + 1e-6)), rnd
        moved.append(max(po.se3_distance(prev[b, 0:3], prev[b, 3:7], cur[b, 0:3], cur[b, 3:7]) for b in range(D)))
        prev = cur
    assert moved[-1] < 0.5 * max(moved[:3]), moved                     # (these noisy scenes creep along a flat valley: 2e-4 per restart at the end)
    # one row against the same alignment solved alone (several CUs per alignment, another kernel shape)
    for b in (0, 5, D - 1):
        hs = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), 1, N, H, W)
        hs.set_alignment(0, als[b]); hs.set_event_frame(0, np.ascontiguousarray(als[b].frame, dtype=np.float32))
        hs.set_state(0, als[b].p0, als[b].q0, als[b].v0); hs.optimize_batch(0, 0, 1)
        ts = hs.results(0, 1)[0]
        assert po.se3_distance(t[b, 0:3], t[b, 3:7], ts[0:3], ts[3:7]) < 1e-6 and ts[14] == t[b, 14]
        hs.close()
    # point order: slot 1 gets alignment 1's points in a random order
    rng = np.random.default_rng(3)
    a = als[1]; perm = rng.permutation(N)
    h.set_keyframe(1, a.norm_coord[perm], a.grad[perm], a.idp[perm], a.weights[perm], a.fx, a.fy, a.cx, a.cy)
    h.set_states(0, P, Q, V); h.optimize_batch(0, 0, B)
    t3 = h.results(0, B)
    assert po.se3_distance(t3[1, 0:3], t3[1, 3:7], t[1, 0:3], t[1, 3:7]) < 1e-6 and t3[1, 14] == t[1, 14]
    assert np.array_equal(t3[2:, :14], t[2:, :14])                      # nobody else noticed
    r = h.residuals(1)
    r_ref = po.Oracle(synth.Alignment(**{**a.__dict__, "frame": np.ascontiguousarray(a.frame, dtype=np.float32).astype(np.float64)})).pose6_eval(t3[1, 0:3], t3[1, 3:7], a.v0)["r"]
    assert np.abs(r - r_ref[perm]).max() <= 2e-5 * np.abs(r_ref).max()  # residuals come back in the caller's point order
    h.close()


def test_ref12_full_batch_properties(gpu, capi, synth, po, als):
    h, (P, Q, V) = _load(capi, als, capi.SOLVER_REF12, num_blocks=1)
    V0 = np.tile(np.full(6, 0.001) / np.linalg.norm(np.full(6, 0.001)), (B, 1))      # Tracker.cpp:45-46
    h.set_states(0, P, Q, V0); h.optimize_batch(0, 0, B)
    h.set_states(0, P, Q, V0); h.optimize_batch(0, 0, B)                # (second solve: strips)
    assert h.last_launch()["layout"] == 2
    t = h.results(0, B)
    assert np.all(t[:, 15] == 1.0)
    assert np.abs(np.linalg.norm(t[:, 3:7], axis=1) - 1.0).max() < 1e-12 and np.abs(np.linalg.norm(t[:, 7:13], axis=1) - 1.0).max() < 1e-12
    for b in list(range(0, B, 257)) + [B - 1]:
        info = h.info(b)
        assert info["num_iterations"] == info["num_successful_steps"] + info["num_unsuccessful_steps"]      # Tracker.cpp:211
        assert info["final_cost"] <= info["initial_cost"] and info["success"]
        assert info["num_iterations"] == h.info(b % D)["num_iterations"] and info["termination"] == h.info(b % D)["termination"]
    # replicas agree to the last bits of sums that are added with fp64 atomics in varying order
    assert max(np.abs(t[b, :13] - t[b % D, :13]).max() for b in range(D, B, 7)) < 1e-9
    h.close()
