"""Where the batched event-frame construction spends its time: 64 slices x 20 000 events on 640x480 (bench.py latency.B64_step_ms builds them
like this), with the events on the strong pixels of a frame (an event camera's edges: many events per pixel) or spread uniformly, with
and without the exponential window weight."""
import importlib, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi"); synth = importlib.import_module("slam-eds_amd.synth")
B, NE, H, W = 64, 20000, 480, 640
al = synth.make_alignment(5000, H=H, W=W, N=2000)
h = capi.Handle(capi.default_config(exec=capi.EXEC_DEVICE), B, 2000, H, W)
rng = np.random.default_rng(0)
strong = np.argwhere(np.abs(al.frame) > 0.25 * np.abs(al.frame).max())
for name, pick in (("strong pixels", strong[rng.integers(0, len(strong), B * NE)]), ("uniform", np.stack([rng.integers(0, H, B * NE), rng.integers(0, W, B * NE)], axis=1))):
    cx = pick[:, 1].astype(np.uint16); cy = pick[:, 0].astype(np.uint16); cp = rng.integers(0, 2, B * NE).astype(np.uint8)
    offs = (np.arange(B + 1) * NE).astype(np.int32)
    print(f"{name}: {len(np.unique(pick[:NE, 0] * W + pick[:NE, 1]))} distinct pixels in the first slice of {NE} events")
    for use_exp in (1, 0):
        ts = []
        for rep in range(8):
            t0 = time.perf_counter()
            rc = capi.lib().eds_trk_build_event_frame_batch(h._h, 0, B, offs.ctypes.data_as(C.POINTER(C.c_int32)), cx.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                            cy.ctypes.data_as(C.POINTER(C.c_uint16)), cp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 0.5, use_exp, None)
            assert rc == 0
            ts.append((time.perf_counter() - t0) * 1e6)
        print(f"   exp weights {use_exp}: {np.median(ts[2:]):.0f} us per batch of {B} frames")
# the part that does not depend on the events (image clears, blur, level + norm, tile store, launches, one wait): slices of 16 events
cx = np.full(B * 16, 100, np.uint16); cy = np.full(B * 16, 100, np.uint16); cp = np.ones(B * 16, np.uint8); offs = (np.arange(B + 1) * 16).astype(np.int32)
ts = []
for rep in range(8):
    t0 = time.perf_counter()
    capi.lib().eds_trk_build_event_frame_batch(h._h, 0, B, offs.ctypes.data_as(C.POINTER(C.c_int32)), cx.ctypes.data_as(C.POINTER(C.c_uint16)),
                                               cy.ctypes.data_as(C.POINTER(C.c_uint16)), cp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 0.5, 1, None)
    ts.append((time.perf_counter() - t0) * 1e6)
print(f"16 events per slice: {np.median(ts[2:]):.0f} us per batch of {B} frames")
h.close()
