"""Another point distribution than SURVEY 8d's uniform one: the points strung along contours, the way a keyframe
selected by gradient strength (KeyFrame.cpp:740-823) looks.  Neighbouring patches then overlap — shared cache lines, equal cells, patches
that coincide after a move — which is exactly where a gather / patch-cache kernel could go wrong.  Full size (640x480, 2 000 points):
the residual/Jacobian pass, the LM6 batch kernels on tiles and strips, one alignment on teams x candidate groups, and REF12, all against
the oracle."""
import importlib

import numpy as np
import pytest


def _rounded(synth, a):
    return synth.Alignment(**{**a.__dict__, "frame": np.ascontiguousarray(a.frame, dtype=np.float32).astype(np.float64)})


def test_edge_layout_is_what_it_says(synth):
    a, u = synth.make_alignment(6200, layout="edges"), synth.make_alignment(6200)

    def lines(al):              # 128-byte lines (8x4-pixel tiles) the 4x4 patches touch at the keyframe position
        s = set()
        for x, y in al.coord.astype(int):
            s.update(((y + dy) // 4, (x + dx) // 8) for dy in range(-1, 3) for dx in range(-1, 3))
        return len(s)
    assert a.N == 2000 and len({(int(x), int(y)) for x, y in a.coord}) == 2000          # distinct pixels
    assert lines(a) < 0.75 * lines(u)                                                     # clustered: a third fewer lines under the same number of patches
    with pytest.raises(ValueError):
        synth.make_alignment(1, layout="blobs")


@pytest.mark.gpu
def test_edge_layout_rows_and_solves_vs_oracle(gpu, capi, synth, po):
    als = [synth.make_alignment(6200 + i, layout="edges") for i in range(4)]
    B = 136                                     # > 128: one CU per alignment (the batch kernels; up to 128 go out on teams of two)
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), B, 2000, 480, 640)
    f32 = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
    for b in range(B):
        a = als[b % 4]
        h.set_keyframe(b, a.norm_coord, a.grad, a.idp, a.weights, a.fx, a.fy, a.cx, a.cy); h.set_event_frame(b, f32[b % 4])
    # the stand-alone residual/Jacobian pass + reduction
    q = synth.quat_from_axis_angle([0.2, -0.4, 0.9], 0.002); p = np.array([0.001, -0.0005, 0.0008])
    g = h.eval(1, p, q, als[1].v0, ncols=6)
    e = po.Oracle(_rounded(synth, als[1])).pose6_eval(p, q, als[1].v0)
    assert np.max(np.abs(g["r"] - e["r"])) <= 1e-5 * np.max(np.abs(e["r"])) and np.linalg.norm(g["J"] - e["J"]) <= 1e-4 * np.linalg.norm(e["J"])
    assert np.linalg.norm(g["JtJ"] - e["H"]) <= 1e-4 * np.linalg.norm(e["H"])
    refs = [po.Oracle(_rounded(synth, a)).pose6_lm(a.p0, a.q0, a.v0, iters=10, lambda0=0.01) for a in als]
    S0 = (np.stack([als[b % 4].p0 for b in range(B)]), np.stack([als[b % 4].q0 for b in range(B)]), np.stack([als[b % 4].v0 for b in range(B)]))
    kernels = []
    for layout in ("tiles", "strips"):          # first solve: the tiles; the same frames again: their strip copies
        h.set_states(0, *S0); h.optimize_batch(0, 0, B)
        tab = h.results(0, B); kernels.append(h.last_launch()["kernel"])
        for b in range(B):
            ref = refs[b % 4]
            assert tab[b, 15] == 1.0 and tab[b, 14] == ref["iterations"], (layout, b)
            assert po.se3_distance(tab[b, 0:3], tab[b, 3:7], ref["p"], ref["q"]) <= 1e-6, (layout, b)
        assert np.array_equal(h.trace(5)["accepted"], refs[1]["accepted"])
    assert kernels[0] != kernels[1] and all(k.startswith("eds_fused6_kernel<0, 4, 512") for k in kernels), kernels
    h.close()
    # the latency regime: one alignment, teams x candidate groups (LM6), and the reference problem
    a = als[2]
    h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), 1, 2000, 480, 640)
    h.set_alignment(0, a)
    pg, qg, _, info = h.optimize(0, p=a.p0, q=a.q0, v=a.v0)
    assert h.last_launch()["cus_per_alignment"] > 1 and info["num_iterations"] == refs[2]["iterations"]
    assert po.se3_distance(pg, qg, refs[2]["p"], refs[2]["q"]) <= 1e-6
    for nb, loss in ((1, capi.LOSS_NONE), (4, capi.LOSS_HUBER)):
        h.set_config(capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=nb, loss_type=loss, loss_param=0.3))
        pg, qg, vg, info = h.optimize(0, p=a.p0, q=a.q0, v=a.v0)
        ref = po.Oracle(_rounded(synth, a), num_blocks=nb, loss_type=po.LOSS_HUBER if loss == capi.LOSS_HUBER else po.LOSS_NONE, loss_param=0.3,
                        max_num_iterations=10).solve_lm(a.p0, a.q0, a.v0)
        assert (info["num_successful_steps"], info["num_unsuccessful_steps"], info["termination"]) == (ref["num_successful_steps"], ref["num_unsuccessful_steps"], ref["termination"]), nb
        assert po.se3_distance(pg, qg, ref["p"], ref["q"]) <= 1e-6 and np.abs(vg - ref["v"]).max() <= 1e-6, nb
    h.close()
