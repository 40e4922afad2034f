"""Timing of the §8f rows on the GPU: event-frame construction, device loss scale, point maintenance."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
import np_frame_oracle as fo
H, W = 480, 640
h = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE), 256, 2000, H, W)
rng = np.random.default_rng(0)
for n in (10_000, 100_000, 1_000_000):
    x = rng.integers(0, W, n).astype(np.uint16); y = rng.integers(0, H, n).astype(np.uint16); pol = rng.integers(0, 2, n).astype(np.uint8)
    h.build_event_frame(0, x, y, pol)
    t = time.perf_counter()
    for _ in range(20): h.build_event_frame(0, x, y, pol)
    dt = (time.perf_counter() - t) / 20
    t = time.perf_counter(); fo.event_frame(x, y, pol, H, W); dcpu = time.perf_counter() - t
    print(f"event frame 640x480 from {n:8d} events: GPU {dt*1e6:8.1f} us per frame incl. event upload ({n/dt/1e6:7.1f} M events/s) | numpy oracle {dcpu*1e3:7.1f} ms")
als = [synth.make_alignment(5000 + i) for i in range(8)]
for b in range(256):
    h.set_alignment(b, als[b % 8])
h.optimize_batch(0, 0, 256)
t = time.perf_counter()
for _ in range(10): tau = h.loss_param_batch(capi.LP_MAD)
print(f"MAD scale of 256 x 2000 residuals on device: {(time.perf_counter()-t)/10*1e6:.1f} us per batch ({(time.perf_counter()-t)/10/256*1e6:.2f} us per alignment)")
h.optimize_batch(0, 0, 256)
t = time.perf_counter(); taus = [h.loss_param(b, capi.LP_MAD) for b in range(256)]; d1 = time.perf_counter() - t
r = [h.residuals(b) for b in range(256)]
t = time.perf_counter(); taus = [h.loss_param(b, capi.LP_MAD) for b in range(256)]; d2 = time.perf_counter() - t
print(f"  single-slot calls: device-resident {d1/256*1e6:.1f} us each; host nth_element path {d2/256*1e6:.1f} us each")
t = time.perf_counter()
for b in range(256): out = h.update_points(b, True)
print(f"point maintenance (getCoord + culling), 2000 points: {(time.perf_counter()-t)/256*1e6:.1f} us per alignment incl. readback of coords/tracks")
# keyframe set-up
import np_keyframe_oracle as ko
h.close()
img = rng.standard_normal((H, W))
for _ in range(3): img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 0) + np.roll(img, -1, 1)) / 5.0
img = np.round(255 * (img - img.min()) / (img.max() - img.min())).astype(np.uint8)
K = (0.78 * W, 0.78 * W, (W - 1) / 2, (H - 1) / 2)
for md in (3000, 30000):
    dxy = np.stack([rng.uniform(0, W - 1, md), rng.uniform(0, H - 1, md)], axis=1); didp = rng.uniform(0.2, 1.0, md)
    hk = capi.Handle(capi.default_config(), 1, H * W, H, W)
    for method, npts, name in ((capi.KF_MAX, 2000, "MAX 2000"), (capi.KF_MAX, 30000, "MAX 30000"), (capi.KF_MEDIAN, 0, "MEDIAN")):
        out = hk.build_keyframe(0, img, K, method=method, num_points=npts, depth_xy=dxy, depth_idp=didp)
        t = time.perf_counter()
        for _ in range(5): out = hk.build_keyframe(0, img, K, method=method, num_points=npts, depth_xy=dxy, depth_idp=didp)
        dt = (time.perf_counter() - t) / 5
        t = time.perf_counter(); ref = ko.keyframe(img, K, method, npts, depth_xy=dxy, depth_idp=didp); dcpu = time.perf_counter() - t
        print(f"keyframe set-up 640x480 {name:10s} depth map {md:6d}: GPU {dt*1e3:7.2f} ms incl. image upload + readback of {len(out['idp']):6d} points | numpy oracle {dcpu*1e3:8.1f} ms")
    hk.close()
# batched event frames (configs[4]: one frame per alignment): 64 slices in one call vs 64 single calls
hb = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE), 64, 2000, H, W)
for n in (10_000, 100_000):
    sl = [(rng.integers(0, W, n).astype(np.uint16), rng.integers(0, H, n).astype(np.uint16), rng.integers(0, 2, n).astype(np.uint8)) for b in range(64)]
    offs = np.arange(65, dtype=np.int32) * n
    cx = np.concatenate([s_[0] for s_ in sl]); cy = np.concatenate([s_[1] for s_ in sl]); cp = np.concatenate([s_[2] for s_ in sl])
    import ctypes as C
    norms = np.zeros(64)
    call = lambda: capi.lib().eds_trk_build_event_frame_batch(hb._h, 0, 64, offs.ctypes.data_as(C.POINTER(C.c_int32)), cx.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                               cy.ctypes.data_as(C.POINTER(C.c_uint16)), cp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 0.5, 1,
                                                               norms.ctypes.data_as(C.POINTER(C.c_double)))
    call()
    t = time.perf_counter()
    for _ in range(5): call()
    tb = (time.perf_counter() - t) / 5
    for b in range(64): hb.build_event_frame(b, *sl[b])
    t = time.perf_counter()
    for b in range(64): hb.build_event_frame(b, *sl[b])
    ts = time.perf_counter() - t
    print(f"64 event frames of {n:7d} events: one batched call {tb*1e6:8.1f} us ({tb/64*1e6:5.1f} us per frame) | 64 single calls {ts*1e6:8.1f} us ({ts/64*1e6:5.1f} each)")
hb.close()
# one tracking step of 64 trackers on one GPU (configs[4] end to end): 64 event slices -> frames, LM6 solves, MAD scales, getCoord / culling
hs = capi.Handle(capi.default_config(solver=capi.SOLVER_LM6, exec=capi.EXEC_DEVICE, max_num_iterations=10), 64, 2000, H, W)
als64 = [synth.make_alignment(5000 + i) for i in range(8)]
for b in range(64): hs.set_alignment(b, als64[b % 8])
sl = []
for b in range(64):
    fr = als64[b % 8].frame
    strong = np.argwhere(np.abs(fr) > 0.25 * np.abs(fr).max())
    pick = strong[rng.integers(0, len(strong), 20_000)]
    sl.append((pick[:, 1].astype(np.uint16), pick[:, 0].astype(np.uint16), (fr[pick[:, 0], pick[:, 1]] > 0).astype(np.uint8)))
offs = (np.arange(65) * 20_000).astype(np.int32)
cx = np.concatenate([s_[0] for s_ in sl]); cy = np.concatenate([s_[1] for s_ in sl]); cp = np.concatenate([s_[2] for s_ in sl])
P0 = np.stack([als64[b % 8].p0 for b in range(64)]); Q0 = np.stack([als64[b % 8].q0 for b in range(64)]); V0 = np.stack([als64[b % 8].v0 for b in range(64)])
import ctypes as C
def step64():
    capi.lib().eds_trk_build_event_frame_batch(hs._h, 0, 64, offs.ctypes.data_as(C.POINTER(C.c_int32)), cx.ctypes.data_as(C.POINTER(C.c_uint16)),
                                               cy.ctypes.data_as(C.POINTER(C.c_uint16)), cp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 0.5, 1, None)
    hs.set_states(0, P0, Q0, V0)
    hs.optimize_batch(0, 0, 64)
    hs.loss_param_batch(capi.LP_MAD, 0, 64)
    hs.update_points_batch(0, 64, False, want_points=False)
for _ in range(3): step64()
t = time.perf_counter()
for _ in range(10): step64()
dt = (time.perf_counter() - t) / 10
print(f"one step of 64 trackers (20 k events each -> frames, LM6 x10, MAD, getCoord criterion): {dt*1e6:.1f} us = {dt/64*1e6:.1f} us per tracker")
hs.close()
