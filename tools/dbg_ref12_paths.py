"""Host-driven loop vs persistent kernel on randomised REF12 problems: prints termination / step counts side by side."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po
rng = np.random.default_rng(11)
for k in range(24):
    N = int(rng.integers(300, 2040))
    al = synth.make_alignment(8100 + k, H=240, W=320, N=N, rot_deg=float(rng.uniform(0.1, 1.5)), trans_norm=float(rng.uniform(0.002, 0.03)),
                              start="ctor" if k % 3 == 0 else "truth_velocity")
    nb, loss, lp, iters = int(rng.integers(1, 9)), int(rng.integers(0, 3)), float(rng.uniform(0.05, 1.0)), int(rng.integers(3, 25))
    row = []
    for ex in ("host", "device"):
        os.environ["EDS_REF12_EXEC"] = ex
        h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_REF12, num_blocks=nb, loss_type=loss, loss_param=lp,
                                            max_num_iterations=iters, function_tolerance=1e-5), 1, al.N, al.H, al.W)
        h.set_alignment(0, al)
        try:
            p, q, v, info = h.optimize(0)
            row.append(f"{ex}: term {info['termination']} it {info['num_iterations']:2d} ok {info['num_successful_steps']:2d} cost {info['final_cost']:.9e}")
        except capi.EdsError as e:
            row.append(f"{ex}: FAIL {e.code}")
        h.close()
    ref = po.Oracle(al, num_blocks=nb, loss_type=loss, loss_param=lp, max_num_iterations=iters, function_tolerance=1e-5).solve_lm(al.p0, al.q0, al.v0)
    print(f"case {k:2d} N {N:4d} nb {nb} loss {loss} cap {iters:2d} | " + " | ".join(row) + f" | oracle: term {ref['termination']} it {ref['num_iterations']:2d} ok {ref['num_successful_steps']:2d} cost {ref['final_cost']:.9e}")
