"""Case files of oracle/_ref/ref_driver (oracle/ref/ref_driver.cpp) — test infrastructure.

write_case() lays an alignment out as the driver reads it, run() calls the driver and reads its answer back.  The driver exists only
where oracle/ref/Makefile's probe found Ceres <= 2.1 + Eigen + OpenCV + yaml-cpp + Rock base-types; available() says whether."""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(os.path.dirname(HERE), "_ref", "ref_driver")


def available() -> bool:
    return os.path.isfile(DRIVER) and os.access(DRIVER, os.X_OK)


def build() -> str:
    """Runs the recipe; returns its last line ('built ...', or 'parity unpinned: ...' where the probe fails — not an error)."""
    r = subprocess.run(["make", "-s", "-C", HERE], capture_output=True, text=True)
    lines = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and not l.startswith("make")]
    return lines[-1] if lines else ""


def write_case(path, al, p, q, v, num_threads=1, loss=0, loss_param=1.0, max_num_iterations=10, function_tolerance=1e-6):
    with open(path, "wb") as f:
        np.array([al.N, al.H, al.W, num_threads, loss, max_num_iterations], dtype=np.int32).tofile(f)
        np.array([al.fx, al.fy, al.cx, al.cy, loss_param, function_tolerance], dtype=np.float64).tofile(f)
        for a in (al.norm_coord, al.grad, al.idp, al.weights, al.frame, p, q, v):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)


def run(al, p, q, v, **kw) -> dict:
    with tempfile.TemporaryDirectory() as d:
        cin, cout = os.path.join(d, "case.bin"), os.path.join(d, "out.bin")
        write_case(cin, al, p, q, v, **kw)
        subprocess.run([DRIVER, cin, cout], check=True, stdout=subprocess.DEVNULL)
        o = np.fromfile(cout, dtype=np.float64)
    return dict(p=o[0:3], q=o[3:7], v=o[7:13], usable=bool(o[13]), num_successful_steps=int(o[14]), num_unsuccessful_steps=int(o[15]),
                termination_type=int(o[16]), initial_cost=o[17], final_cost=o[18], residuals=o[19:19 + al.N])
