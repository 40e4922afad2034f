"""Replays trial 60 of `tools/fuzz_pyramid.py 8808 120` (round 6's long soak: the one REF12 pyramid whose two paths ended 7.8e-6 apart with
equal step accounting and equal costs at every level): the SAME eds_pyr_optimize call repeated — its own answers differ by 1e-6 .. 6e+3 from run to run,
with and without the completion-word wait.  A coarse level of 269 points on noise: the cost is flat in the pose, and the run-to-run last bits of REF12's
fp64 LDS atomics decide where the solve walks.  Ill-posed input, not a discrepancy between paths."""
import importlib, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
import pyoracle as po
rng = np.random.default_rng(8808)
for t in range(61):
    L = int(rng.integers(2, 5)); H = int(rng.integers(30, 70)) << (L - 1); W = int(rng.integers(40, 90)) << (L - 1)
    N0 = int(rng.integers(300, min(9000, (H - 40) * (W - 40) // 2))); ref12 = bool(rng.integers(0, 2))
    nb = int(rng.integers(1, 4)) if ref12 else 1
counts = [max(64, N0 >> l) for l in range(L)]
print(t, H, W, L, counts, ref12, nb)
al = synth.make_alignment(9500 + t, H=H, W=W, N=N0, rot_deg=0.4, trans_norm=0.008, blur_ksize=11, blur_sigma=3.0, start="ctor")
cfg = capi.default_config(solver=capi.SOLVER_REF12, exec=capi.EXEC_DEVICE, max_num_iterations=5, num_blocks=nb)
for poll in ("1", "0"):
    os.environ["EDS_POLL_RESULTS"] = poll
    pyr = capi.Pyramid(cfg, counts, H, W)
    for l, n in enumerate(counts):
        pyr.set_keyframe(l, al.norm_coord[:n], al.grad[:n], al.idp[:n], al.weights[:n], al.fx, al.fy, al.cx, al.cy)
    pyr.set_event_frame(al.frame)
    res = [pyr.optimize(al.p0, al.q0, al.v0)[:3] for _ in range(6)]
    print("poll", poll, "pyramid run-to-run distances:", ["%.1e" % po.se3_distance(res[0][0], res[0][1], r[0], r[1]) for r in res[1:]])
    pyr.close()
