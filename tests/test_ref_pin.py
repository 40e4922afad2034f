"""Pins the oracle against the REAL reference where the reference can be built (Next #7).

oracle/ref/ is a recipe: a Makefile that probes for Ceres <= 2.1 + Eigen + OpenCV + yaml-cpp + Rock base-types and, where they exist,
compiles oracle/ref/ref_driver.cpp — which includes /root/reference/src/tracking/PhotometricError.hpp unmodified and replays
Tracker::optimize (Tracker.cpp:104-241) — into oracle/_ref/ref_driver.  This image has none of those libraries, so here the probe
says "parity unpinned" and the comparison below is skipped with exactly that reason; on a box that has them the same test asserts
oracle == reference on the golden cases."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref"))
import refcase  # noqa: E402


def test_recipe_probes_and_never_fakes_a_build():
    msg = refcase.build() if os.path.isdir("/root/reference/src") else "parity unpinned: no reference tree"
    assert refcase.available() or msg.startswith("parity unpinned"), msg
    # nothing of the reference stays under the repository: no stand-in headers, no link into /root/reference
    assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "include"))
    drv = open(os.path.join(ROOT, "oracle", "ref", "ref_driver.cpp")).read()
    assert "#include <eds/tracking/PhotometricError.hpp>" in drv and len(drv.splitlines()) <= 100


@pytest.mark.parametrize("golden,loss", [("small_n64.npz", 0), ("small_n200_nb3.npz", 0), ("small_n200_nb3.npz", 1), ("small_n64.npz", 2)])
def test_oracle_equals_the_reference_on_the_golden_cases(synth, po, golden, loss):
    if not refcase.available():
        pytest.skip("PARITY UNPINNED: oracle/_ref/ref_driver cannot be built in this image (Ceres <= 2.1, Eigen, OpenCV, yaml-cpp, Rock base-types absent)")
    g = np.load(os.path.join(ROOT, "tests", "golden", golden))
    al = synth.Alignment(**{**synth.make_alignment(int(g["seed"]), H=int(g["H"]), W=int(g["W"]), N=int(g["N"])).__dict__,
                            "norm_coord": g["norm_coord"], "grad": g["grad"], "idp": g["idp"], "weights": g["weights"], "frame": g["frame"]})
    nb = int(g["num_blocks"])
    ref = refcase.run(al, g["start_p"], g["start_q"], al.v0, num_threads=nb, loss=loss, loss_param=0.3, max_num_iterations=10)
    ours = po.Oracle(al, num_blocks=nb, loss_type=loss, loss_param=0.3, max_num_iterations=10).solve_lm(g["start_p"], g["start_q"], al.v0)
    assert ref["usable"] and ours["usable"]
    assert (ref["num_successful_steps"], ref["num_unsuccessful_steps"]) == (ours["num_successful_steps"], ours["num_unsuccessful_steps"])
    assert po.se3_distance(ours["p"], ours["q"], ref["p"], ref["q"]) <= 1e-8 and np.abs(ours["v"] - ref["v"]).max() <= 1e-8
    assert abs(ours["final_cost"] - ref["final_cost"]) <= 1e-10 * max(1.0, abs(ref["final_cost"]))
    r = po.Oracle(al, num_blocks=nb).eval12(ref["p"], ref["q"], ref["v"], jac=False)["r_raw"]
    assert np.abs(r - ref["residuals"]).max() <= 1e-10 * np.abs(ref["residuals"]).max()
