"""The failure path of the team launches (eds_fused.hip / eds_fused12.hip): a team whose members do not all arrive reports a time-out
after EDS_TEAM_TIMEOUT_TICKS (50 ms) instead of hanging, and the collect step re-runs the range with one CU per alignment and never
forms teams on that handle again.  EDS_TEAM_TEST_DROP_MEMBER launches the team grid one workgroup short — exactly the situation the
bound exists for — without touching the kernels."""
import importlib
import os
import time

import numpy as np
import pytest

capi = importlib.import_module("slam-eds_amd.capi")
synth = importlib.import_module("slam-eds_amd.synth")

pytestmark = pytest.mark.gpu


def _solve(solver, B, drop, seeds):
    cfg = capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, num_blocks=1, max_num_iterations=10)
    h = capi.Handle(cfg, B, 2000, 480, 640)
    als = [synth.make_alignment(s) for s in seeds]
    for b, a in enumerate(als):
        h.set_alignment(b, a)
    p0 = np.stack([a.p0 for a in als]); q0 = np.stack([a.q0 for a in als]); v0 = np.stack([a.v0 for a in als])
    walls, tables = [], []
    for rep in range(3):
        h.set_states(0, p0, q0, v0)
        if drop and rep == 0:
            os.environ["EDS_TEAM_TEST_DROP_MEMBER"] = "1"
        t = time.perf_counter()
        try:
            h.optimize_batch(0, 0, B)
        finally:
            os.environ.pop("EDS_TEAM_TEST_DROP_MEMBER", None)
        walls.append(time.perf_counter() - t)
        tables.append(np.array(h.results(0, B)))
    infos = [h.info(b) for b in range(B)]
    h.close()
    return walls, tables, infos


@pytest.mark.parametrize("solver", [capi.SOLVER_LM6, capi.SOLVER_REF12])
def test_incomplete_team_times_out_and_falls_back(solver):
    seeds = [7100, 7101, 7102]
    walls_ok, tab_ok, _ = _solve(solver, 3, False, seeds)
    walls, tab, infos = _solve(solver, 3, True, seeds)
    # the short launch waited for the bound, then the range was solved again: well above a normal call, far below a hang
    assert 0.045 < walls[0] < 2.0, walls
    assert walls_ok[0] < 0.02
    # every alignment has a usable result, the one whose team was incomplete included
    assert all(i["success"] for i in infos)
    # the fallback (one CU per alignment) agrees with the team solve of the undisturbed handle to the last digits of fp64 sums
    np.testing.assert_allclose(tab[0], tab_ok[0], rtol=1e-6, atol=1e-6)
    # later calls on the handle no longer form teams: same results as its own fallback solve, at normal speed (LM6: bit for bit; the REF12
    # kernel adds its per-wavefront tiles with fp64 LDS atomics, whose order varies from run to run)
    if solver == capi.SOLVER_LM6:
        assert np.array_equal(tab[1], tab[0]) and np.array_equal(tab[2], tab[0])
    else:
        np.testing.assert_allclose(tab[1], tab[0], rtol=1e-6, atol=1e-6); np.testing.assert_allclose(tab[2], tab[0], rtol=1e-6, atol=1e-6)
    assert walls[2] < 0.02, walls
