"""The headline batch (4 096 x 2 000 points, LM6, frames new for the solve, then on strip copies; REF12 the same) through RAW ctypes calls
that every ABI since 4 has — so that two builds of the library with different ABIs can be A/B'd inside one gpurun call:
    python3 tools/ab_headline_raw.py slam-eds_amd/csrc/libeds_hip_r5.so slam-eds_amd/csrc/libeds_hip.so"""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
capi = importlib.import_module("slam-eds_amd.capi")          # (structures and constants only: the libraries are loaded by hand)
synth = importlib.import_module("slam-eds_amd.synth")
B, N, H, W, D = 4096, 2000, 480, 640, 16
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as pool:
    als = list(pool.map(lambda i: synth.make_alignment(5000 + i), range(D)))
fr = [np.ascontiguousarray(a.frame, dtype=np.float32) for a in als]
dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
P0 = f64(np.stack([als[b % D].p0 for b in range(B)])); Q0 = f64(np.stack([als[b % D].q0 for b in range(B)])); V0 = f64(np.stack([als[b % D].v0 for b in range(B)]))


def run(path):
    L = C.CDLL(os.path.join(ROOT, path))
    L.eds_last_error.restype = C.c_char_p
    out = []
    for solver, name in ((capi.SOLVER_LM6, "LM6"), (capi.SOLVER_REF12, "REF12")):
        cfg = capi.default_config(solver=solver, exec=capi.EXEC_DEVICE, max_num_iterations=10, num_blocks=1)
        h = C.c_void_p()
        assert L.eds_trk_create(C.byref(cfg), B, N, H, W, C.byref(h)) == 0, L.eds_last_error()
        for b in range(B):
            a = als[b % D]
            arrs = [f64(a.norm_coord), f64(a.grad), f64(a.idp), f64(a.weights)]
            assert L.eds_trk_set_keyframe(h, b, N, *[x.ctypes.data_as(dp) for x in arrs], C.c_double(a.fx), C.c_double(a.fy), C.c_double(a.cx), C.c_double(a.cy)) == 0
            assert L.eds_trk_set_event_frame_f32(h, b, fr[b % D].ctypes.data_as(fp)) == 0
        for layout in ("tiles", "strips"):
            assert L.eds_trk_set_knob(h, b"EDS_FUSED_LAYOUT", b"tiles" if layout == "tiles" else None) == 0
            ks = []
            for _ in range(7):
                assert L.eds_trk_set_states(h, 0, B, P0.ctypes.data_as(dp), Q0.ctypes.data_as(dp), V0.ctypes.data_as(dp)) == 0
                assert L.eds_trk_optimize_batch(h, 0, 0, B) == 0 and L.eds_trk_sync(h) == 0, L.eds_last_error()
                info = capi.Info(); L.eds_trk_get_info(h, 0, C.byref(info)); ks.append(info.device_time_us)
            out.append(f"{name} {layout}: kernel {np.median(ks[2:]):8.1f} us")
        L.eds_trk_destroy(h)
    print(f"{path:44s} " + "   ".join(out), flush=True)


for rep in range(2):
    for path in sys.argv[1:]:
        run(path)
