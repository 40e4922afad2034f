"""configs[3]: one coarse-to-fine call (4 levels, 2 000 -> 16 000 points, 10 iterations each) — wall time of eds_pyr_set_event_frame
(upload + 3 down-samplings) and of eds_pyr_optimize (4 solves, pose carried on)."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
counts = [16000, 8000, 4000, 2000]
al = synth.make_alignment(3234, H=480, W=640, N=16000, rot_deg=0.6, trans_norm=0.012, blur_ksize=15, blur_sigma=4.0)
for solver, name in ((capi.SOLVER_LM6, "lm6"), (capi.SOLVER_REF12, "ref12")):
    pyr = capi.Pyramid(capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10), counts, 480, 640)
    for l, n in enumerate(counts):
        pyr.set_keyframe(l, al.norm_coord[:n], al.grad[:n], al.idp[:n], al.weights[:n], al.fx, al.fy, al.cx, al.cy)
    tf, to = [], []
    for _ in range(10):
        t0 = time.perf_counter(); pyr.set_event_frame(al.frame)
        t1 = time.perf_counter(); p, q, v, infos = pyr.optimize(al.p0, al.q0, al.v0)
        t2 = time.perf_counter(); tf.append(t1 - t0); to.append(t2 - t1)
    print("%-5s frame+pyramid %.1f us  optimize %.1f us  (kernels: %s us)" % (name, np.median(tf[3:]) * 1e6, np.median(to[3:]) * 1e6,
          " + ".join("%.0f" % i["device_time_us"] for i in reversed(infos))))
    pyr.close()
