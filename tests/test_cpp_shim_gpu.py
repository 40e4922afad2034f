"""The C++ shim `slam-eds_amd/csrc/Tracker.hpp` (reference Tracker.hpp member signatures over the C ABI),
driven from a plain g++ program the way the external EDS component drives eds::tracking::Tracker."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "cpp", "shim_demo.cpp")
EXE = os.path.join(HERE, "cpp", "shim_demo")


def build_demo(capi):
    deps = [SRC, os.path.join(ROOT, "slam-eds_amd", "csrc", "Tracker.hpp"), os.path.join(ROOT, "include", "eds_hip.h"), capi.LIB_PATH]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", SRC, "-o", EXE, "-L", capi.CSRC, "-leds_hip",
                               "-Wl,-rpath," + capi.CSRC])
    return EXE


def test_shim_compiles_against_c_abi_only(capi):
    """No HIP / torch headers are needed on the caller's side: plain g++ + include/eds_hip.h."""
    build_demo(capi)
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "libeds_hip.so" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nb,loss", [(1, 0), (4, 1)])
def test_shim_optimize_matches_oracle(gpu, capi, synth, po, tmp_path, nb, loss):
    exe = build_demo(capi)
    al = synth.make_alignment(808, H=240, W=320, N=700, start="ctor")
    path = tmp_path / "al.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("3i", al.N, al.H, al.W))
        f.write(struct.pack("4d", al.fx, al.fy, al.cx, al.cy))
        for a in (al.norm_coord, al.grad, al.idp, al.weights, al.frame, al.v0):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    res = subprocess.run([exe, str(path), str(nb), str(loss), "12"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    r = json.loads(res.stdout.strip().splitlines()[-1])
    ref = po.Oracle(al, num_blocks=nb, loss_type=loss, loss_param=0.3, max_num_iterations=12).solve_lm(al.p0, al.q0, al.v0)
    assert r["ok"] == 1 and ref["usable"]
    assert r["iterations"] == ref["num_iterations"] and r["num_points"] == al.N and r["residuals"] == al.N
    assert np.abs(np.array(r["t"]) - ref["p"]).max() <= 1e-4
    assert np.abs(np.array(r["R"]).reshape(3, 3) - po.quat_to_R(ref["q"] / np.linalg.norm(ref["q"]))).max() <= 1e-4
    assert np.abs(np.array(r["v"]) - ref["v"]).max() <= 1e-4
    assert r["inverse_err"] < 1e-12                      # optimize returns T_kf_ef = getTransform().inverse()
    r_fin = po.Oracle(al, num_blocks=nb).eval12(ref["p"], ref["q"], ref["v"], jac=False)["r_raw"]
    assert r["tau"] == pytest.approx(po.loss_param(r_fin, po.LP_MAD)[0], rel=1e-3)       # loss_params rewritten (Tracker.cpp:233)
    # Tracker::getCoord(true) + needNewKeyframe through the shim, against the point-maintenance oracle
    import np_points_oracle as pto
    pm = np.array([0.06, -0.03, 0.01])
    qm = synth.quat_from_axis_angle([0.1, 1.0, 0.2], 0.05)
    refp = pto.get_coord(al.norm_coord, al.idp, al.coord, (al.fx, al.fy, al.cx, al.cy), al.H, al.W, pm, qm, True)
    assert r["consistent"] == 1 and r["kept"] == len(refp["kept"]) and 0 < r["kept"] < al.N
    assert r["first_x"] == pytest.approx(refp["coord"][0, 0], abs=1e-4) and r["last_y"] == pytest.approx(refp["coord"][-1, 1], abs=1e-4)
    assert r["sq_flow"] == pytest.approx(refp["mean_sq_flow"], rel=1e-4)
    assert bool(r["need_kf"]) == pto.need_new_keyframe(refp["mean_sq_flow"], al.H, al.W, 0.03)
    assert r["first_idp"] == al.idp[refp["kept"][0]]
    # second / fourth solve on the unchanged KeyFrame (device copy reused, then inverse depth changed and restored) reproduce the first
    assert r["rep_err"] <= 1e-9
    # getTransform(bool&): false + identity for the first two calls, true + the (mean-filtered) pose on the third (Tracker.cpp:251-260)
    assert r["filt_flags"] == 4 | 8 and r["filt_err"] <= 1e-12
    print(f"shim live call (frame upload + solve + residuals + MAD, keyframe reused): {r['live_call_us']:.0f} us, of which solve {r['live_solve_us']:.0f} us")


def test_shim_signature_drift_breaks_the_build(tmp_path):
    """The pointer-to-member pins of tests/cpp/shim_eds_types_check.cpp really bite: a copy of the shim whose getVelocity()
    returns by value (the round-1 drift) must fail to compile."""
    shim = open(os.path.join(ROOT, "slam-eds_amd", "csrc", "Tracker.hpp")).read()
    good = "Eigen::Matrix<double, 6, 1>& getVelocity() { return vx; }"
    assert good in shim
    d = tmp_path / "slam-eds_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "Tracker.hpp").write_text(shim.replace(good, "Eigen::Matrix<double, 6, 1> getVelocity() { return vx; }")
                                   .replace('#include "../../include/eds_hip.h"', f'#include "{os.path.join(ROOT, "include", "eds_hip.h")}"'))
    t = tmp_path / "tests" / "cpp"
    t.mkdir(parents=True)
    (t / "check.cpp").write_text(open(os.path.join(HERE, "cpp", "shim_eds_types_check.cpp")).read())
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(HERE, "cpp", "mock_eds"), str(t / "check.cpp")],
                         capture_output=True, text=True)
    assert res.returncode != 0 and "getVelocity" in res.stderr


def test_shim_eds_types_branch_compiles_against_mock_headers():
    """The EDS_HIP_WITH_EDS_TYPES branch of Tracker.hpp (what an EDS build would compile: Eigen / OpenCV / Rock types and the
    real KeyFrame) cannot be built here for lack of those libraries; tests/cpp/mock_eds declares the members it touches
    with the reference's names and signatures so that the branch is at least compile-checked."""
    src = os.path.join(HERE, "cpp", "shim_eds_types_check.cpp")
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(HERE, "cpp", "mock_eds"), src],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_shim_reports_failures_as_false_not_as_exceptions(capi):
    """Reference path: no exceptions (Tracker.cpp:104-241 returns bool).  The shim maps every failure of the library underneath to
    `false` / an empty vector and keeps status + message (hipLastStatus / hipLastError).  Runs without a GPU: the device does not exist."""
    exe = build_demo(capi)
    res = subprocess.run([exe, "--no-device"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    r = json.loads(res.stdout.strip().splitlines()[-1])
    status = r.pop("status")
    assert status in (capi.ERR_NO_DEVICE, capi.ERR_INVALID)           # no device at all (this container) / ordinal out of range (a GPU box)
    assert r == {"threw": 0, "good": 0, "T_untouched": 1, "coords": 0, "message": "set"}
    assert "throw" not in open(os.path.join(ROOT, "slam-eds_amd", "csrc", "Tracker.hpp")).read().split("#pragma once")[1]


def test_reference_members_unit_links_against_the_shim(tmp_path, capi):
    """EDS_HIP_REFERENCE_MEMBERS: the shim declares trackPoints, trackPointsPyr, trackPointsAlongEpiline, getEMatrix, getFMatrix and
    getFilteredPose with the reference's signatures (pinned in shim_eds_types_check.cpp) and a translation unit of the EDS tree defines
    them (tests/cpp/shim_reference_members.cpp stands in for the reference's Tracker.cpp:378-648): both compile against the mock EDS
    types and link into one object — every member declared is defined exactly once, the private members the reference bodies touch are
    reachable under the reference's names."""
    inc = os.path.join(HERE, "cpp", "mock_eds")
    objs = []
    for name in ("shim_eds_types_check", "shim_reference_members"):
        o = str(tmp_path / (name + ".o"))
        res = subprocess.run(["g++", "-std=c++17", "-fPIC", "-Wall", "-c", "-I", inc, os.path.join(HERE, "cpp", name + ".cpp"), "-o", o], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        objs.append(o)
    so = str(tmp_path / "libshim_check.so")
    res = subprocess.run(["g++", "-shared", "-Wl,--no-undefined", "-o", so] + objs + ["-L", capi.CSRC, "-leds_hip", "-Wl,-rpath," + capi.CSRC], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    syms = subprocess.run(["nm", "-C", "--defined-only", so], capture_output=True, text=True).stdout
    for m in ("trackPoints(", "trackPointsPyr(", "trackPointsAlongEpiline(", "getEMatrix()", "getFMatrix()", "getFilteredPose("):
        assert f"eds::tracking::Tracker::{m}" in syms, m
    # without the switch the six are not declared at all (an EDS tree that calls them must opt in)
    probe = tmp_path / "probe.cpp"
    probe.write_text('#define EDS_HIP_WITH_EDS_TYPES\n#include "' + os.path.join(ROOT, "slam-eds_amd", "csrc", "Tracker.hpp") +
                     '"\ncv::Mat f(eds::tracking::Tracker& t) { return t.getEMatrix(); }\n')
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", inc, str(probe)], capture_output=True, text=True)
    assert res.returncode != 0 and "getEMatrix" in res.stderr
