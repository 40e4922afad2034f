"""The failure path of the team launches (eds_fused.hip / eds_fused12.hip): a team whose members do not all arrive reports a time-out
after EDS_TEAM_TIMEOUT_TICKS (5 ms) instead of hanging, the collect step re-runs the range with one CU per alignment, teams pause on
that handle for EDS_TEAM_COOLDOWN solves (eds_trk_info.flags says so) and then come back by themselves.  EDS_TEAM_TEST_DROP_MEMBER launches the team grid one workgroup short — exactly the situation the
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
            h.set_knob("EDS_TEAM_TEST_DROP_MEMBER", "1")       # (a knob of THIS handle: the library reads no environment after create)
        t = time.perf_counter()
        try:
            h.optimize_batch(0, 0, B)
        finally:
            h.set_knob("EDS_TEAM_TEST_DROP_MEMBER", None)
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
    assert 0.0045 < walls[0] < 2.0, walls
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
    # the event is visible to the caller: the call that timed out carries EDS_INFO_TEAM_TIMEOUT ... (infos = the LAST call: paused)
    assert all(i["flags"] & capi.INFO_TEAMS_PAUSED for i in infos)


@pytest.mark.parametrize("solver", [capi.SOLVER_LM6, capi.SOLVER_REF12])
def test_teams_pause_then_come_back(solver):
    """One time-out must not cost the handle its teams for good: EDS_TEAM_COOLDOWN solves with one CU per alignment (flagged), then
    team launches again — same kernel time as before the time-out, results unchanged, flags clear; and a SECOND time-out after the
    re-arm works exactly like the first (the ticket counter was reset, so the teams of later launches form correctly)."""
    cfg = capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, num_blocks=1, max_num_iterations=10)
    B = 2
    h = capi.Handle(cfg, B, 2000, 480, 640)
    als = [synth.make_alignment(s) for s in (7200, 7201)]
    for b, a in enumerate(als):
        h.set_alignment(b, a)
    p0 = np.stack([a.p0 for a in als]); q0 = np.stack([a.q0 for a in als]); v0 = np.stack([a.v0 for a in als])

    def call(drop=False):
        h.set_states(0, p0, q0, v0)
        if drop:
            h.set_knob("EDS_TEAM_TEST_DROP_MEMBER", "1")
        try:
            t = time.perf_counter(); h.optimize_batch(0, 0, B); w = time.perf_counter() - t
        finally:
            h.set_knob("EDS_TEAM_TEST_DROP_MEMBER", None)
        return w, np.array(h.results(0, B)), h.info(0)

    for _ in range(3):
        w_team, tab_team, info_team = call()
    assert info_team["flags"] == 0 and w_team < 0.02
    for round_ in range(2):                                    # a time-out, the pause, the come-back — twice
        w, tab, info = call(drop=True)
        assert w > 0.0045 and info["flags"] & capi.INFO_TEAM_TIMEOUT and info["success"]
        np.testing.assert_allclose(tab, tab_team, rtol=1e-6, atol=1e-6)
        pause = capi.TEAM_COOLDOWN * (2 ** round_)             # a time-out right after a re-arm doubles the pause
        dev_paused, w_paused = [], []
        for k in range(pause):
            w, tab, info = call()
            assert info["flags"] == capi.INFO_TEAMS_PAUSED, (round_, k, info["flags"])
            assert info["success"]
            dev_paused.append(info["device_time_us"]); w_paused.append(w)
        # normal speed during the pause (the median: a shared box may stall any single call), and no time-out wait inside the kernels
        assert np.median(w_paused) < 0.005 and max(dev_paused) < 5000.0, (round_, sorted(w_paused)[-3:], max(dev_paused))
        w, tab, info = call()                                  # re-armed
        assert info["flags"] == 0 and info["success"], (round_, info["flags"])
        np.testing.assert_allclose(tab, tab_team, rtol=1e-6, atol=1e-6)
        # several CUs per alignment again: the kernel is faster than the one-CU kernel of the pause (2 000 points: ~0.07 vs ~0.10 ms LM6)
        assert info["device_time_us"] < 0.95 * np.median(dev_paused), (info["device_time_us"], np.median(dev_paused))
    h.close()
