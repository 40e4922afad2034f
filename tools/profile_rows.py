"""Workload for the rocprofv3 profile of the rows around the solve: event frames (all levels from one vote), MAD
loss scale, point maintenance, keyframe set-up, the image pyramid.  No numpy oracle timing here (profiling target only).
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rows --output-format csv -- python3 tools/profile_rows.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
H, W = 480, 640
rng = np.random.default_rng(0)
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE), 64, 2000, H, W)
n = 100_000
x = rng.integers(0, W, n).astype(np.uint16); y = rng.integers(0, H, n).astype(np.uint16); pol = rng.integers(0, 2, n).astype(np.uint8)
for _ in range(10): h.build_event_frames(0, 3, x, y, pol)
t = time.perf_counter()
for _ in range(20): h.build_event_frames(0, 3, x, y, pol)
print(f"3 levels from one vote, 100 k events: {(time.perf_counter() - t) / 20 * 1e6:.1f} us")
t = time.perf_counter()
for _ in range(20): h.build_event_frame(0, x, y, pol)
print(f"1 level, 100 k events: {(time.perf_counter() - t) / 20 * 1e6:.1f} us")
als = [synth.make_alignment(5000 + i) for i in range(8)]
for b in range(64): h.set_alignment(b, als[b % 8])
for _ in range(5):
    h.optimize_batch(0, 0, 64); h.loss_param_batch(capi.LP_MAD)
for b in range(64): h.update_points(b, True)
# the batched forms (configs[4]: one event slice per tracker): 64 slices of 20 k events, getCoord of all 64
sl = [(rng.integers(0, W, 20_000).astype(np.uint16), rng.integers(0, H, 20_000).astype(np.uint16), rng.integers(0, 2, 20_000).astype(np.uint8)) for _ in range(64)]
for _ in range(6): h.build_event_frame_batch(0, sl)
for b in range(64): h.set_alignment(b, als[b % 8])
h.optimize_batch(0, 0, 64)
for _ in range(6): h.update_points_batch(0, 64, False, want_points=False)
h.close()
img = rng.standard_normal((H, W))
for _ in range(3): img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 0) + np.roll(img, -1, 1)) / 5.0
img = np.round(255 * (img - img.min()) / (img.max() - img.min())).astype(np.uint8)
K = (0.78 * W, 0.78 * W, (W - 1) / 2, (H - 1) / 2)
dxy = np.stack([rng.uniform(0, W - 1, 3000), rng.uniform(0, H - 1, 3000)], axis=1); didp = rng.uniform(0.2, 1.0, 3000)
hk = capi.Handle(capi.default_config(), 1, H * W, H, W)
for _ in range(5): hk.build_keyframe(0, img, K, method=capi.KF_MAX, num_points=2000, depth_xy=dxy, depth_idp=didp)
rgb2 = np.repeat(np.repeat(np.stack([img, img, img], axis=2), 2, axis=0), 2, axis=1)
for _ in range(5): hk.build_keyframe(0, rgb2, K, method=capi.KF_MAX, num_points=2000, depth_xy=dxy, depth_idp=didp)
hk.close()
pyr = capi.Pyramid(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=6), [2000, 2000, 2000, 2000], H, W)
for _ in range(5): pyr.set_event_frame(als[0].frame)
pyr.close()
