"""ctypes binding of the C-ABI library ``csrc/libeds_hip.so`` (header: ``include/eds_hip.h``).

This is plumbing only: every number is produced by the HIP kernels behind the C ABI.
There is no CPU fallback — if the library is missing or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("EDS_HIP_LIB") or os.path.join(CSRC, "libeds_hip.so")   # env override: diagnostic builds only

# enums of include/eds_hip.h
EDS_OK = 0
ERR_INVALID, ERR_HIP, ERR_NOT_USABLE, ERR_STATE, ERR_NO_DEVICE = -1, -2, -3, -4, -5
SAMPLE_BICUBIC, SAMPLE_BILINEAR = 0, 1
SOLVER_GN6, SOLVER_LM6, SOLVER_REF12 = 0, 1, 2
EXEC_HOST, EXEC_DEVICE = 0, 1
LOSS_NONE, LOSS_HUBER, LOSS_CAUCHY = 0, 1, 2
LP_CONSTANT, LP_MAD, LP_STD = 0, 1, 2
KF_MAX, KF_MEDIAN = 0, 1
IMG_U8, IMG_F32, IMG_F64 = 0, 1, 2
MAX_LEVELS = 8

# every symbol include/eds_hip.h declares (tests check the .so exports all of them)
EXPORTS = (
    "eds_abi_version", "eds_device_count", "eds_last_error", "eds_trk_cfg_default", "eds_trk_cfg_size",
    "eds_trk_info_size",
    "eds_trk_create", "eds_trk_destroy", "eds_trk_set_config", "eds_trk_get_config",
    "eds_trk_set_keyframe", "eds_trk_set_idepth", "eds_trk_set_idepth_strided", "eds_trk_set_event_frame", "eds_trk_set_event_frame_f32", "eds_trk_set_event_frames", "eds_trk_set_event_frames_f32",
    "eds_trk_set_undistort_map", "eds_trk_set_undistort_map_sized", "eds_trk_build_event_frame", "eds_trk_build_event_frames", "eds_trk_build_event_frame_batch", "eds_trk_build_event_frames_aos", "eds_trk_build_event_frames_aos_timed", "eds_event_times_aos",
    "eds_trk_get_event_frame", "eds_trk_share_event_frame",
    "eds_trk_set_state", "eds_trk_get_state", "eds_trk_set_states", "eds_trk_get_states", "eds_trk_get_results",
    "eds_trk_eval", "eds_trk_optimize", "eds_trk_optimize_batch", "eds_trk_optimize_batch_wait",
    "eds_trk_sync", "eds_trk_get_info", "eds_trk_get_trace", "eds_trk_get_residuals", "eds_trk_loss_param", "eds_trk_residuals_and_loss",
    "eds_trk_loss_param_batch", "eds_trk_update_points", "eds_trk_update_points_batch",
    "eds_kf_select_default", "eds_trk_build_keyframe", "eds_trk_build_keyframe_image", "eds_trk_get_keyframe_points",
    "eds_trk_timer_start", "eds_trk_timer_stop", "eds_trk_bench_eval", "eds_trk_bench_live", "eds_trk_bench_batch", "eds_trk_last_launch", "eds_trk_prepare_frames",
    "eds_trk_set_knob", "eds_trk_get_strips_info", "eds_trk_bench_kernel_cold", "eds_trk_hbm_probe", "eds_trk_kernel_instances",
    "eds_pyr_create", "eds_pyr_destroy", "eds_pyr_set_config", "eds_pyr_level_intrinsics", "eds_pyr_set_keyframe",
    "eds_pyr_set_event_frame", "eds_pyr_build_event_frame", "eds_pyr_level_size", "eds_pyr_get_level_frame", "eds_pyr_optimize",
    "eds_pyr_get_residuals", "eds_pyr_create_batch", "eds_pyr_set_keyframe_slot", "eds_pyr_set_event_frame_slot", "eds_pyr_optimize_batch",
)

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class KfSelect(C.Structure):
    """``eds_kf_select`` — the arguments of KeyFrame::create that steer the point set-up (KeyFrame.cpp:333-341)."""
    _fields_ = [("method", C.c_int32), ("cell", C.c_int32), ("num_points", C.c_int32), ("sobel_ksize", C.c_int32),
                ("min_depth", C.c_double), ("max_depth", C.c_double), ("weight_threshold", C.c_double)]


class Cfg(C.Structure):
    """``eds_trk_cfg`` — mirrors eds::tracking::Config (reference tracking/Config.hpp:40-58)."""
    _fields_ = [("device", C.c_int32), ("sampling", C.c_int32), ("solver", C.c_int32), ("exec", C.c_int32),
                ("num_blocks", C.c_int32), ("loss_type", C.c_int32), ("loss_param", C.c_double),
                ("huber_tau", C.c_double), ("lambda0", C.c_double), ("num_levels", C.c_int32),
                ("max_num_iterations", C.c_int32 * MAX_LEVELS), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double), ("nc", C.c_int32),
                ("reserved", C.c_int32 * 7)]


class Info(C.Structure):
    """``eds_trk_info`` — mirrors eds::tracking::TrackerInfo (reference tracking/Config.hpp:60-68)."""
    _fields_ = [("meas_time_us", C.c_double), ("num_points", C.c_uint32), ("num_iterations", C.c_int32),
                ("time_seconds", C.c_double), ("success", C.c_uint8), ("flags", C.c_uint8), ("pad_", C.c_uint8 * 2),
                ("termination", C.c_int32), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("device_time_us", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad_"}


class EventTimes(C.Structure):
    """``eds_event_times`` — EventFrame::create's time bookkeeping (EventFrame.cpp:313-335)."""
    _fields_ = [("first_time", C.c_int64), ("last_time", C.c_int64), ("time", C.c_int64), ("delta_time", C.c_int64),
                ("last_valid", C.c_int32), ("reserved", C.c_int32)]


def event_times(events, prev_last_time: int = 0) -> dict:
    """first / last / middle-element time stamp and their difference of a structured event array with an int64 field `ts`;
    raises EdsError(ERR_INVALID) when events[0].ts > the last time stamp, like the reference's throw.  `prev_last_time`: the previous
    slice's last_time (the reference object keeps it: a one-event slice carries it over — include/eds_hip.h)."""
    ev = np.ascontiguousarray(events)
    t = EventTimes()
    t.last_time = int(prev_last_time)
    _check(lib().eds_event_times_aos(int(ev.shape[0]), ev.ctypes.data_as(C.c_void_p), int(ev.dtype.itemsize), int(ev.dtype.fields["ts"][1]), C.byref(t)))
    return {k: getattr(t, k) for k, _ in t._fields_ if k != "reserved"}


class LaunchInfo(C.Structure):
    """``eds_trk_launch_info`` — what the last on-device solve launched (include/eds_hip.h)."""
    _fields_ = [("kernel", C.c_char * 96), ("workgroups", C.c_int32), ("cus_per_alignment", C.c_int32), ("first", C.c_int32),
                ("count", C.c_int32), ("layout", C.c_int32), ("timing_source", C.c_int32), ("span_us", C.c_double),
                ("mean_workgroup_us", C.c_double), ("covered", C.c_double), ("tail_idle_us", C.c_double)]


INFO_TEAM_TIMEOUT, INFO_TEAMS_PAUSED = 1, 2       # eds_trk_info.flags (include/eds_hip.h)
TEAM_COOLDOWN = 16                                # EDS_TEAM_COOLDOWN (csrc/eds_fused.hpp): solves without teams after a time-out


class EdsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libeds_hip error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile libeds_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "eds_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    rccl_lib = os.path.join(CSRC, "libeds_hip_rccl.so")       # include/eds_hip_rccl.h: the RCCL gather for a C / C++ caller (its own library)
    rccl_src = [os.path.join(CSRC, "eds_gather.hip"), os.path.join(_HERE, "..", "include", "eds_hip_rccl.h")]
    stale_rccl = (not os.path.exists(rccl_lib)) or any(os.path.getmtime(s) > os.path.getmtime(rccl_lib) for s in rccl_src)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-j4", "-s", "libeds_hip.so"])
    if force or stale_rccl:
        subprocess.check_call(["make", "-C", CSRC, "-s", "libeds_hip_rccl.so"])
    return LIB_PATH


_lib = None
# PyTorch-ROCm ships its own libamdhip64.so.7 and libeds_hip.so is linked against /opt/rocm's: one process holds ONE copy of a SONAME,
# whichever is loaded first.  torch works only on its own, libeds_hip on either — so a process that uses torch.cuda as well must
# import torch before the first call into this module (bench.py and BatchTracker do; batch.gather_results checks).
torch_loaded_first = None


def lib():
    """Loads the library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EdsError(ERR_NO_DEVICE, f"{LIB_PATH} is missing — run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        global torch_loaded_first
        import sys
        torch_loaded_first = "torch" in sys.modules
        L = C.CDLL(LIB_PATH)
        L.eds_last_error.restype = C.c_char_p
        L.eds_trk_create.argtypes = [C.POINTER(Cfg), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.eds_trk_destroy.argtypes = [C.c_void_p]
        L.eds_trk_destroy.restype = None
        L.eds_trk_cfg_default.argtypes = [C.POINTER(Cfg)]
        L.eds_trk_cfg_default.restype = None
        L.eds_trk_set_config.argtypes = [C.c_void_p, C.POINTER(Cfg)]
        L.eds_trk_get_config.argtypes = [C.c_void_p, C.POINTER(Cfg)]
        L.eds_trk_set_keyframe.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_double,
                                           C.c_double, C.c_double]
        L.eds_trk_set_idepth.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
        L.eds_trk_set_idepth_strided.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int]
        L.eds_trk_set_event_frame.argtypes = [C.c_void_p, C.c_int, _dp]
        L.eds_trk_set_event_frame_f32.argtypes = [C.c_void_p, C.c_int, _fp]
        L.eds_trk_set_event_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.eds_trk_set_event_frames_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.eds_trk_set_state.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
        L.eds_trk_get_state.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
        L.eds_trk_set_undistort_map.argtypes = [C.c_void_p, _fp, _fp]
        L.eds_trk_build_event_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                                C.POINTER(C.c_uint8), C.c_int, C.c_double, C.c_int, _dp]
        L.eds_trk_get_event_frame.argtypes = [C.c_void_p, C.c_int, _dp]
        L.eds_trk_set_undistort_map_sized.argtypes = [C.c_void_p, _fp, _fp, C.c_int, C.c_int]
        L.eds_trk_build_event_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                                 C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, C.c_int, _dp]
        L.eds_trk_build_event_frames_aos.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.c_double, C.c_int, _dp]
        L.eds_trk_build_event_frames_aos_timed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                           C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _dp, C.POINTER(EventTimes)]
        L.eds_event_times_aos.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(EventTimes)]
        L.eds_trk_build_event_frame_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                                      C.POINTER(C.c_uint8), C.c_int, C.c_double, C.c_int, _dp]
        L.eds_trk_share_event_frame.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.eds_trk_set_states.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]
        L.eds_trk_get_states.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]
        L.eds_trk_get_results.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
        L.eds_trk_eval.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.eds_trk_optimize.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(Info)]
        L.eds_trk_optimize_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.eds_trk_optimize_batch_wait.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.eds_trk_sync.argtypes = [C.c_void_p]
        L.eds_trk_get_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(Info)]
        L.eds_trk_get_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip]
        L.eds_trk_get_residuals.argtypes = [C.c_void_p, C.c_int, _dp]
        L.eds_trk_loss_param.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
        L.eds_trk_loss_param_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp]
        L.eds_trk_update_points.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip, _ip, _dp]
        L.eds_trk_update_points_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _ip, _ip, _dp]
        L.eds_kf_select_default.argtypes = [C.POINTER(KfSelect)]
        L.eds_kf_select_default.restype = None
        L.eds_trk_build_keyframe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(KfSelect), C.c_int, _dp, _dp,
                                             C.c_double, C.c_double, C.c_double, C.c_double, _ip]
        L.eds_trk_get_keyframe_points.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.eds_trk_build_keyframe_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(KfSelect),
                                                   C.c_int, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _ip]
        L.eds_trk_timer_start.argtypes = [C.c_void_p]
        L.eds_trk_timer_stop.argtypes = [C.c_void_p, _fp]
        L.eds_trk_bench_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.eds_trk_bench_live.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp]
        L.eds_trk_bench_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp]
        L.eds_trk_bench_kernel_cold.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.eds_trk_hbm_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _fp, _fp]
        L.eds_pyr_create.argtypes = [C.POINTER(Cfg), C.c_int, _ip, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.eds_pyr_destroy.argtypes = [C.c_void_p]
        L.eds_pyr_destroy.restype = None
        L.eds_pyr_set_config.argtypes = [C.c_void_p, C.POINTER(Cfg)]
        L.eds_pyr_level_intrinsics.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        L.eds_pyr_set_keyframe.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.eds_pyr_set_event_frame.argtypes = [C.c_void_p, _dp]
        L.eds_pyr_build_event_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(C.c_uint8),
                                                C.c_double, C.c_int, _dp]
        L.eds_pyr_level_size.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
        L.eds_pyr_get_level_frame.argtypes = [C.c_void_p, C.c_int, _dp]
        L.eds_pyr_optimize.argtypes = [C.c_void_p, _dp, _dp, _dp, C.POINTER(Info)]
        L.eds_pyr_create_batch.argtypes = [C.POINTER(Cfg), C.c_int, C.c_int, _ip, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.eds_pyr_set_keyframe_slot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.eds_pyr_set_event_frame_slot.argtypes = [C.c_void_p, C.c_int, _dp]
        L.eds_pyr_optimize_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(Info)]
        L.eds_trk_last_launch.argtypes = [C.c_void_p, C.POINTER(LaunchInfo)]
        L.eds_trk_set_knob.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.eds_trk_get_strips_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), _ip, _ip]
        L.eds_pyr_get_residuals.argtypes = [C.c_void_p, C.c_int, _dp]
        if L.eds_trk_cfg_size() != C.sizeof(Cfg) or L.eds_trk_info_size() != C.sizeof(Info):
            raise EdsError(ERR_INVALID, "ctypes struct layout disagrees with include/eds_hip.h")
        _lib = L
    return _lib


def last_error() -> str:
    return (lib().eds_last_error() or b"").decode()


def _check(rc):
    if rc != EDS_OK:
        raise EdsError(rc, last_error())


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def default_config(**kw) -> Cfg:
    cfg = Cfg()
    lib().eds_trk_cfg_default(C.byref(cfg))
    for k, v in kw.items():
        if k == "max_num_iterations":
            v = [int(v)] * MAX_LEVELS if np.isscalar(v) else list(v) + [list(v)[-1]] * (MAX_LEVELS - len(v))
            cfg.max_num_iterations = (C.c_int32 * MAX_LEVELS)(*v[:MAX_LEVELS])
        else:
            setattr(cfg, k, v)
    return cfg


class Handle:
    """RAII wrapper of ``eds_trk*``: ``batch`` alignment slots on one GPU / one HIP stream."""

    def __init__(self, cfg: Cfg, batch: int, max_points: int, H: int, W: int):
        self._h = C.c_void_p()
        self.batch, self.max_points, self.H, self.W = int(batch), int(max_points), int(H), int(W)
        _check(lib().eds_trk_create(C.byref(cfg), self.batch, self.max_points, self.H, self.W, C.byref(self._h)))
        self._N = [0] * self.batch

    def close(self):
        if self._h:
            lib().eds_trk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------------------
    def set_config(self, cfg: Cfg):
        _check(lib().eds_trk_set_config(self._h, C.byref(cfg)))

    def get_config(self) -> Cfg:
        cfg = Cfg()
        _check(lib().eds_trk_get_config(self._h, C.byref(cfg)))
        return cfg

    # -- inputs --------------------------------------------------------------------------
    def set_keyframe(self, slot, norm_coord, grad, idp, weights, fx, fy, cx, cy):
        nc, g, d, w = _f64(norm_coord), _f64(grad), _f64(idp), _f64(weights)
        N = int(d.shape[0])
        if nc.shape != (N, 2) or g.shape != (N, 2) or w.shape != (N,):
            raise EdsError(ERR_INVALID, "keyframe arrays have inconsistent shapes")
        _check(lib().eds_trk_set_keyframe(self._h, slot, N, _p(nc), _p(g), _p(d), _p(w), fx, fy, cx, cy))
        self._N[slot] = N

    def set_alignment(self, slot, al):
        """Convenience: upload an ``Alignment`` (synth.py) and seed its start state."""
        self.set_keyframe(slot, al.norm_coord, al.grad, al.idp, al.weights, al.fx, al.fy, al.cx, al.cy)
        self.set_event_frame(slot, al.frame)
        self.set_state(slot, al.p0, al.q0, al.v0)

    def set_idepth(self, slot, idp):
        d = _f64(idp)
        _check(lib().eds_trk_set_idepth(self._h, slot, int(d.shape[0]), _p(d)))

    def set_idepth_strided(self, slot, table, column=0):
        """Inverse depths as column `column` of a row-major N x k table of doubles (DepthPoints keeps N x 4)."""
        t = np.ascontiguousarray(table, dtype=np.float64)
        _check(lib().eds_trk_set_idepth_strided(self._h, slot, int(t.shape[0]), t[:, column:].ctypes.data_as(_dp), int(t.shape[1])))

    def set_event_frame(self, slot, frame):
        fr = np.asarray(frame)
        if fr.size != self.H * self.W:
            raise EdsError(ERR_INVALID, "event frame size != H*W")
        if fr.dtype == np.float32:
            fr = np.ascontiguousarray(fr)
            _check(lib().eds_trk_set_event_frame_f32(self._h, slot, fr.ctypes.data_as(_fp)))
        else:
            fr = _f64(fr)
            _check(lib().eds_trk_set_event_frame(self._h, slot, _p(fr)))

    def set_event_frames(self, first, frames):
        """Many host frames in ONE call (eds_trk_set_event_frames / _f32): frames[i] -> slot first + i; all fp64 or all fp32, each H x W."""
        arrs = [np.ascontiguousarray(f) for f in frames]
        f32 = all(a.dtype == np.float32 for a in arrs)
        if not f32:
            arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
        for a in arrs:
            if a.size != self.H * self.W:
                raise ValueError("frame size does not match the handle")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        fn = lib().eds_trk_set_event_frames_f32 if f32 else lib().eds_trk_set_event_frames
        _check(fn(self._h, int(first), len(arrs), ptrs))

    def set_undistort_map(self, mapx=None, mapy=None):
        if mapx is None:
            _check(lib().eds_trk_set_undistort_map(self._h, None, None))
            return
        mx, my = np.ascontiguousarray(mapx, dtype=np.float32), np.ascontiguousarray(mapy, dtype=np.float32)
        if mx.size != self.H * self.W or my.size != self.H * self.W:
            raise EdsError(ERR_INVALID, "undistortion map size != H*W")
        _check(lib().eds_trk_set_undistort_map(self._h, mx.ctypes.data_as(_fp), my.ctypes.data_as(_fp)))

    def build_event_frame(self, slot, x, y, polarity, level=0, blur_sigma=0.5, use_exp_weights=True):
        """EventFrame::create on the device; returns the Frobenius norm the frame was divided by."""
        ex = np.ascontiguousarray(x, dtype=np.uint16)
        ey = np.ascontiguousarray(y, dtype=np.uint16)
        pol = np.ascontiguousarray(polarity, dtype=np.uint8)
        norm = C.c_double(0.0)
        _check(lib().eds_trk_build_event_frame(self._h, slot, int(ex.shape[0]), ex.ctypes.data_as(C.POINTER(C.c_uint16)),
                                               ey.ctypes.data_as(C.POINTER(C.c_uint16)), pol.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               int(level), float(blur_sigma), int(bool(use_exp_weights)),
                                               C.cast(C.byref(norm), _dp)))
        return norm.value

    def build_event_frames(self, first_slot, num_levels, x, y, polarity, sensor_size=None, blur_sigma=0.5, use_exp_weights=True):
        """All `num_levels` frames of one event slice from a single vote (EventFrame::create), level i into slot first_slot + i;
        sensor_size = (H, W) of the events / LUT when it differs from the handle's frame size (out_scale != 1).  Returns the norms."""
        x = np.ascontiguousarray(x, dtype=np.uint16); y = np.ascontiguousarray(y, dtype=np.uint16)
        pol = np.ascontiguousarray(polarity, dtype=np.uint8)
        sH, sW = (0, 0) if sensor_size is None else sensor_size
        norms = np.zeros(num_levels)
        _check(lib().eds_trk_build_event_frames(self._h, int(first_slot), int(num_levels), int(x.shape[0]),
                                                x.ctypes.data_as(C.POINTER(C.c_uint16)), y.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                pol.ctypes.data_as(C.POINTER(C.c_uint8)), int(sH), int(sW), float(blur_sigma),
                                                int(bool(use_exp_weights)), _p(norms)))
        return norms

    def build_event_frames_aos(self, first_slot, num_levels, events, sensor_size=None, blur_sigma=0.5, use_exp_weights=True):
        """`events`: a numpy structured array with fields x, y (uint16) and polarity (1 byte) — e.g. the memory of a
        std::vector<base::samples::Event>."""
        ev = np.ascontiguousarray(events)
        f = ev.dtype.fields
        sH, sW = sensor_size if sensor_size is not None else (0, 0)
        norms = np.zeros(num_levels)
        _check(lib().eds_trk_build_event_frames_aos(self._h, int(first_slot), int(num_levels), int(ev.shape[0]), ev.ctypes.data_as(C.c_void_p),
                                                    int(ev.dtype.itemsize), int(f["x"][1]), int(f["y"][1]), int(f["polarity"][1]), int(sH), int(sW),
                                                    float(blur_sigma), int(bool(use_exp_weights)), _p(norms)))
        return norms

    def build_event_frames_aos_timed(self, first_slot, num_levels, events, sensor_size=None, blur_sigma=0.5, use_exp_weights=True):
        """As build_event_frames_aos on records that also carry `ts` (int64): returns (norms, times dict); the time check comes first."""
        ev = np.ascontiguousarray(events)
        f = ev.dtype.fields
        sH, sW = sensor_size if sensor_size is not None else (0, 0)
        norms = np.zeros(num_levels)
        t = EventTimes()
        _check(lib().eds_trk_build_event_frames_aos_timed(self._h, int(first_slot), int(num_levels), int(ev.shape[0]), ev.ctypes.data_as(C.c_void_p),
                                                          int(ev.dtype.itemsize), int(f["x"][1]), int(f["y"][1]), int(f["polarity"][1]), int(f["ts"][1]),
                                                          int(sH), int(sW), float(blur_sigma), int(bool(use_exp_weights)), _p(norms), C.byref(t)))
        return norms, {k: getattr(t, k) for k, _ in t._fields_ if k != "reserved"}

    def build_event_frame_batch(self, first_slot, slices, level=0, blur_sigma=0.5, use_exp_weights=True):
        """`slices`: one (x, y, polarity) triple per slot, first_slot onwards; returns the norms."""
        offs = np.zeros(len(slices) + 1, dtype=np.int32)
        for b, (x, _, _) in enumerate(slices):
            offs[b + 1] = offs[b] + len(x)
        cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(sl[k], dtype=dt) for sl in slices]) if len(slices) else np.zeros(0, dt), dtype=dt)
        x, y, pol = cat(0, np.uint16), cat(1, np.uint16), cat(2, np.uint8)
        norms = np.zeros(len(slices))
        _check(lib().eds_trk_build_event_frame_batch(self._h, int(first_slot), len(slices), offs.ctypes.data_as(_ip),
                                                     x.ctypes.data_as(C.POINTER(C.c_uint16)), y.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                     pol.ctypes.data_as(C.POINTER(C.c_uint8)), int(level), float(blur_sigma),
                                                     int(bool(use_exp_weights)), _p(norms)))
        return norms

    def set_undistort_map_sized(self, mapx, mapy, sensor_size):
        mx = np.ascontiguousarray(mapx, dtype=np.float32); my = np.ascontiguousarray(mapy, dtype=np.float32)
        assert mx.shape == tuple(sensor_size) and my.shape == tuple(sensor_size)
        _check(lib().eds_trk_set_undistort_map_sized(self._h, mx.ctypes.data_as(_fp), my.ctypes.data_as(_fp), int(sensor_size[0]), int(sensor_size[1])))

    def share_event_frame(self, slot, src_slot):
        _check(lib().eds_trk_share_event_frame(self._h, int(slot), int(src_slot)))

    def get_event_frame(self, slot):
        fr = np.zeros((self.H, self.W))
        _check(lib().eds_trk_get_event_frame(self._h, slot, _p(fr)))
        return fr

    def set_state(self, slot, p=None, q=None, v=None):
        p = None if p is None else _f64(p)
        q = None if q is None else _f64(q)
        v = None if v is None else _f64(v)
        _check(lib().eds_trk_set_state(self._h, slot, _p(p), _p(q), _p(v)))

    def get_state(self, slot):
        p, q, v = np.zeros(3), np.zeros(4), np.zeros(6)
        _check(lib().eds_trk_get_state(self._h, slot, _p(p), _p(q), _p(v)))
        return p, q, v

    def set_states(self, first, p=None, q=None, v=None):
        """Bulk seed of slots [first, first+count): p count x 3, q count x 4, v count x 6."""
        arrs = [None if a is None else _f64(a) for a in (p, q, v)]
        count = next(a.shape[0] for a in arrs if a is not None)
        _check(lib().eds_trk_set_states(self._h, first, count, *[_p(a) for a in arrs]))

    def get_states(self, first=0, count=None):
        count = self.batch - first if count is None else count
        p, q, v = np.zeros((count, 3)), np.zeros((count, 4)), np.zeros((count, 6))
        _check(lib().eds_trk_get_states(self._h, first, count, _p(p), _p(q), _p(v)))
        return p, q, v

    def results(self, first=0, count=None, out=None):
        """count x 16 table: p[3] q[4] v[6] final_cost iterations success.  `out`: a C-contiguous float64 array of that shape to fill
        (a caller that steps in a loop keeps one: a fresh half-megabyte array per call is an mmap and its page faults)."""
        count = self.batch - first if count is None else count
        if out is None:
            out = np.empty((count, 16))         # (the call fills every column)
        elif out.shape != (count, 16) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise EdsError(ERR_INVALID, "results(out=...): a C-contiguous float64 array of shape (count, 16)")
        _check(lib().eds_trk_get_results(self._h, first, count, _p(out)))
        return out

    # -- evaluation / solve --------------------------------------------------------------
    def eval(self, slot, p, q, v, ncols=6, want_jacobian=True):
        N = self._N[slot]
        p, q, v = _f64(p), _f64(q), _f64(v)
        r = np.zeros(N)
        J = np.zeros((N, ncols)) if want_jacobian else None
        JtJ, Jtr, cost = np.zeros((ncols, ncols)), np.zeros(ncols), C.c_double(0.0)
        _check(lib().eds_trk_eval(self._h, slot, _p(p), _p(q), _p(v), ncols, _p(r), _p(J), _p(JtJ), _p(Jtr),
                                  C.cast(C.byref(cost), _dp)))
        return dict(r=r, J=J, JtJ=JtJ, Jtr=Jtr, cost=cost.value)

    def optimize(self, slot, level=0, p=None, q=None, v=None):
        """Tracker::optimize for one slot.  Returns (p, q, v, info dict); raises EdsError(ERR_NOT_USABLE)."""
        if p is None or q is None or v is None:         # (the stored state only where the caller leaves one out: a call less on the live path)
            sp, sq, sv = self.get_state(slot)
        p = sp if p is None else _f64(p).copy()
        q = sq if q is None else _f64(q).copy()
        v = sv if v is None else _f64(v).copy()
        info = Info()
        _check(lib().eds_trk_optimize(self._h, slot, level, _p(p), _p(q), _p(v), C.byref(info)))
        return p, q, v, info.as_dict()

    def optimize_batch(self, level=0, first=0, count=None, sync=True):
        count = self.batch - first if count is None else count
        if sync:            # ABI 6: launch + wait + collect in one call (no interpreter between them)
            _check(lib().eds_trk_optimize_batch_wait(self._h, level, first, count))
        else:
            _check(lib().eds_trk_optimize_batch(self._h, level, first, count))

    def sync(self):
        _check(lib().eds_trk_sync(self._h))

    def info(self, slot):
        info = Info()
        _check(lib().eds_trk_get_info(self._h, slot, C.byref(info)))
        return info.as_dict()

    def trace(self, slot, max_iters=128):
        inc, cost, acc = np.zeros((max_iters, 6)), np.zeros(max_iters), np.zeros(max_iters, dtype=np.int32)
        n = lib().eds_trk_get_trace(self._h, slot, max_iters, _p(inc), _p(cost), acc.ctypes.data_as(_ip))
        if n < 0:
            _check(n)
        return dict(increments=inc[:n], costs=cost[:n], accepted=acc[:n])

    def residuals(self, slot):
        r = np.zeros(self._N[slot])
        _check(lib().eds_trk_get_residuals(self._h, slot, _p(r)))
        return r

    def loss_param(self, slot, method, current=0.0):
        tau = C.c_double(current)
        _check(lib().eds_trk_loss_param(self._h, slot, int(method), C.cast(C.byref(tau), _dp)))
        return tau.value

    def residuals_and_loss(self, slot, method, current=0.0):
        """Tracker.cpp:223-233 in one call: (kf->residuals as the MAD selection leaves them, loss scale)."""
        r = np.zeros(self._N[slot])
        tau = C.c_double(current)
        _check(lib().eds_trk_residuals_and_loss(self._h, slot, int(method), _p(r), C.cast(C.byref(tau), _dp)))
        return r, tau.value

    def loss_param_batch(self, method, first=0, count=None):
        count = self.batch - first if count is None else count
        tau = np.zeros(count)
        _check(lib().eds_trk_loss_param_batch(self._h, first, count, int(method), _p(tau)))
        return tau

    def update_points(self, slot, delete_out_points=True):
        """Tracker::getCoord(delete_out_point): returns dict(coord, tracks, kept, mean_sq_flow)."""
        N = self._N[slot]
        coord, tracks = np.zeros((N, 2)), np.zeros((N, 2))
        kept = np.zeros(N, dtype=np.int32)
        n, flow = C.c_int32(0), C.c_double(0.0)
        _check(lib().eds_trk_update_points(self._h, slot, int(bool(delete_out_points)), _p(coord), _p(tracks),
                                           kept.ctypes.data_as(_ip), C.cast(C.byref(n), _ip), C.cast(C.byref(flow), _dp)))
        self._N[slot] = n.value
        return dict(coord=coord[:n.value], tracks=tracks[:n.value], kept=kept[:n.value], mean_sq_flow=flow.value)

    def update_points_batch(self, first=0, count=None, delete_out_points=True, want_points=True):
        """Tracker::getCoord(delete_out_point) for a range of slots in one call: list of dicts like update_points (coord / tracks /
        kept are None with want_points=False: only culling, counts and the mean squared flow)."""
        count = self.batch - first if count is None else count
        stride = max(self._N[first:first + count] + [1])
        n = np.zeros(count, dtype=np.int32); flow = np.zeros(count)
        if want_points:
            coord, tracks = np.zeros((count, stride, 2)), np.zeros((count, stride, 2))
            kept = np.zeros((count, stride), dtype=np.int32)
            _check(lib().eds_trk_update_points_batch(self._h, first, count, int(bool(delete_out_points)), stride, _p(coord), _p(tracks),
                                                     kept.ctypes.data_as(_ip), n.ctypes.data_as(_ip), _p(flow)))
        else:
            _check(lib().eds_trk_update_points_batch(self._h, first, count, int(bool(delete_out_points)), stride, None, None, None,
                                                     n.ctypes.data_as(_ip), _p(flow)))
        out = []
        for b in range(count):
            self._N[first + b] = int(n[b])
            out.append(dict(coord=coord[b, :n[b]] if want_points else None, tracks=tracks[b, :n[b]] if want_points else None,
                            kept=kept[b, :n[b]] if want_points else None, mean_sq_flow=float(flow[b]), n=int(n[b])))
        return out

    # -- keyframe set-up on the device ------------------------------------------------------
    def build_keyframe(self, slot, img, K, method=KF_MEDIAN, num_points=0, cell=20, depth_xy=None, depth_idp=None,
                       min_depth=1.0, max_depth=3.0, weight_threshold=0.7, sobel_ksize=3):
        """KeyFrame::create's tracker-facing part on the device; returns dict(coord, norm_coord, grad, idp, weights)."""
        img = np.ascontiguousarray(img)
        if img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] not in (1, 3)):
            raise EdsError(ERR_INVALID, "image must be H x W (grey) or H x W x 3 (RGB)")
        channels = 1 if img.ndim == 2 else int(img.shape[2])
        img_H, img_W = int(img.shape[0]), int(img.shape[1])      # != (self.H, self.W): out_scale != 1, resized on the device
        if img.dtype == np.uint8:
            ty = IMG_U8
        elif img.dtype == np.float32:
            ty = IMG_F32
        else:
            ty, img = IMG_F64, np.ascontiguousarray(img, dtype=np.float64)
        sel = KfSelect()
        lib().eds_kf_select_default(C.byref(sel))
        sel.method, sel.cell, sel.num_points, sel.sobel_ksize = int(method), int(cell), int(num_points), int(sobel_ksize)
        sel.min_depth, sel.max_depth, sel.weight_threshold = float(min_depth), float(max_depth), float(weight_threshold)
        nd = 0 if depth_xy is None else len(depth_xy)
        dxy = _f64(depth_xy) if nd else None
        didp = _f64(depth_idp) if nd else None
        n = C.c_int32(0)
        fx, fy, cx, cy = [float(k) for k in K]
        if channels == 1 and (img_H, img_W) == (self.H, self.W):
            _check(lib().eds_trk_build_keyframe(self._h, slot, ty, img.ctypes.data_as(C.c_void_p), C.byref(sel), nd,
                                                _p(dxy) if nd else None, _p(didp) if nd else None, fx, fy, cx, cy,
                                                C.cast(C.byref(n), _ip)))
        else:
            _check(lib().eds_trk_build_keyframe_image(self._h, slot, ty, img.ctypes.data_as(C.c_void_p), img_H, img_W, channels,
                                                      C.byref(sel), nd, _p(dxy) if nd else None, _p(didp) if nd else None,
                                                      fx, fy, cx, cy, C.cast(C.byref(n), _ip)))
        N = n.value
        self._N[slot] = N
        out = dict(coord=np.zeros((N, 2)), norm_coord=np.zeros((N, 2)), grad=np.zeros((N, 2)), idp=np.zeros(N), weights=np.zeros(N))
        _check(lib().eds_trk_get_keyframe_points(self._h, slot, _p(out["coord"]), _p(out["norm_coord"]), _p(out["grad"]),
                                                 _p(out["idp"]), _p(out["weights"])))
        return out

    # -- measurement ---------------------------------------------------------------------
    def timer_start(self):
        _check(lib().eds_trk_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float(0.0)
        _check(lib().eds_trk_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def prepare_frames(self, first=0, count=None, force=False) -> float:
        """Converts the (stale, or with force all) frames of the range to the strip layout now; returns the device time in ms."""
        count = self.batch - first if count is None else count
        ms = C.c_float(0.0)
        _check(lib().eds_trk_prepare_frames(self._h, int(first), int(count), int(bool(force)), C.byref(ms)))
        return ms.value

    def set_knob(self, name: str, value=None) -> None:
        """One tuning knob of THIS handle (``eds_trk_set_knob``): same names and values as the environment variables the handle read at
        creation; ``None`` / ``""`` restores the default."""
        _check(lib().eds_trk_set_knob(self._h, name.encode(), None if value is None else str(value).encode()))

    def strips_info(self) -> dict:
        b, ph, un = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        _check(lib().eds_trk_get_strips_info(self._h, C.byref(b), C.byref(ph), C.byref(un)))
        return dict(bytes=int(b.value), row_phases=int(ph.value), unavailable=bool(un.value))

    def last_launch(self) -> dict:
        li = LaunchInfo()
        _check(lib().eds_trk_last_launch(self._h, C.byref(li)))
        d = {k: getattr(li, k) for k, _ in li._fields_}
        d["kernel"] = li.kernel.decode()
        return d

    def bench_live(self, slot, p0, q0, v0, level=0, idp=None, frame=None, method=-1, reps=50) -> dict:
        """``eds_trk_bench_live``: the live sequence timed inside the library (medians, microseconds)."""
        p0, q0, v0 = _f64(p0), _f64(q0), _f64(v0)
        idp = None if idp is None else _f64(idp)
        frame = None if frame is None else _f64(frame)
        out = np.zeros(6)
        _check(lib().eds_trk_bench_live(self._h, int(slot), int(level), None if idp is None else _p(idp), None if frame is None else _p(frame),
                                        _p(p0), _p(q0), _p(v0), int(method), int(reps), _p(out)))
        return dict(zip(("total_us", "set_idepth_us", "set_event_frame_us", "optimize_us", "residuals_and_loss_us", "kernel_us"), out.tolist()))

    def bench_batch(self, P, Q, V, first=0, level=0, reps=50) -> dict:
        """``eds_trk_bench_batch``: reps x {set_states; optimize_batch_wait} timed inside the library (medians + the slowest step, microseconds)."""
        P, Q, V = _f64(P), _f64(Q), _f64(V)
        out = np.zeros(5)
        _check(lib().eds_trk_bench_batch(self._h, int(level), int(first), int(P.shape[0]), _p(P), _p(Q), _p(V), int(reps), _p(out)))
        return dict(zip(("step_us", "set_states_us", "solve_us", "kernel_us", "slowest_step_us"), out.tolist()))

    def bench_eval(self, first, count, ncols=6, with_reduction=False, reps=20) -> float:
        ms = C.c_float(0.0)
        _check(lib().eds_trk_bench_eval(self._h, first, count, ncols, int(with_reduction), reps, C.byref(ms)))
        return ms.value


    def bench_kernel_cold(self, first, count, ncols=6, which=0, reps=10) -> float:
        """One streaming kernel (which: 0 residual/Jacobian, 1 reduction) timed cold — 1 GiB streamed through the caches in front of
        every repetition; mean ms per launch (HIP events on the handle's stream)."""
        ms = C.c_float(0.0)
        _check(lib().eds_trk_bench_kernel_cold(self._h, first, count, ncols, int(which), reps, C.byref(ms)))
        return ms.value

    def hbm_probe(self, nbytes=1 << 30, reps=10) -> dict:
        """What the box's HBM streams through the library's own plain kernel: read-only pass and copy (read + write counted), GB/s."""
        r, c = C.c_float(0.0), C.c_float(0.0)
        _check(lib().eds_trk_hbm_probe(self._h, int(nbytes), reps, C.byref(r), C.byref(c)))
        return {"read_GBps": r.value, "copy_GBps": c.value, "bytes": int(nbytes), "reps": reps}


def kernel_instances(family: int):
    """The compiled instantiations of eds_fused6_kernel (family 0: S, P, T, Q, K, G) / eds_fused12_kernel (1: S, T, CAP, NC, K, Q)."""
    L = lib()
    n = L.eds_trk_kernel_instances(int(family), -1, None)
    out = []
    for i in range(max(n, 0)):
        a = (C.c_int32 * 6)()
        L.eds_trk_kernel_instances(int(family), i, a)
        out.append(tuple(int(x) for x in a))
    return out


class Pyramid:
    """RAII wrapper of ``eds_pyr*``: coarse-to-fine tracking of one alignment on an image pyramid (BASELINE.json configs[3])."""

    def __init__(self, cfg: Cfg, max_points, H: int, W: int, batch: int = 1):
        self.levels = len(max_points)
        self.H, self.W, self.batch = int(H), int(W), int(batch)
        mp = np.ascontiguousarray(max_points, dtype=np.int32)
        self._h = C.c_void_p()
        self._N = [0] * self.levels
        _check(lib().eds_pyr_create_batch(C.byref(cfg), self.batch, self.levels, mp.ctypes.data_as(_ip), self.H, self.W, C.byref(self._h)))

    # -- batched pyramids (batch > 1): the same calls per pyramid (slot), one launch per level for all of them ----------------
    def set_keyframe_slot(self, slot, level, norm_coord, grad, idp, weights, fx, fy, cx, cy):
        nc, g, d, w = _f64(norm_coord), _f64(grad), _f64(idp), _f64(weights)
        self._N[level] = max(self._N[level], int(d.shape[0]))
        _check(lib().eds_pyr_set_keyframe_slot(self._h, int(slot), int(level), int(d.shape[0]), _p(nc), _p(g), _p(d), _p(w), fx, fy, cx, cy))

    def set_event_frame_slot(self, slot, frame):
        f = _f64(frame)
        assert f.size == self.H * self.W
        _check(lib().eds_pyr_set_event_frame_slot(self._h, int(slot), _p(f)))

    def optimize_batch(self, P, Q, V, first=0, want_infos=True):
        """P, Q, V: count x 3 / 4 / 6 start states of pyramids [first, first + count).  Returns (P, Q, V, infos[level][k])."""
        P, Q, V = _f64(P).copy(), _f64(Q).copy(), _f64(V).copy()
        count = P.shape[0]
        infos = (Info * (self.levels * count))() if want_infos else None
        _check(lib().eds_pyr_optimize_batch(self._h, int(first), count, _p(P), _p(Q), _p(V), infos))
        out = [[infos[l * count + k].as_dict() for k in range(count)] for l in range(self.levels)] if want_infos else None
        return P, Q, V, out

    def close(self):
        if self._h:
            lib().eds_pyr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_config(self, cfg: Cfg):
        _check(lib().eds_pyr_set_config(self._h, C.byref(cfg)))

    @staticmethod
    def level_intrinsics(level, fx, fy, cx, cy):
        K = np.zeros(4)
        _check(lib().eds_pyr_level_intrinsics(int(level), fx, fy, cx, cy, _p(K)))
        return K

    def set_keyframe(self, level, norm_coord, grad, idp, weights, fx, fy, cx, cy):
        nc, g, d, w = _f64(norm_coord), _f64(grad), _f64(idp), _f64(weights)
        self._N[level] = int(d.shape[0])
        _check(lib().eds_pyr_set_keyframe(self._h, int(level), self._N[level], _p(nc), _p(g), _p(d), _p(w), fx, fy, cx, cy))

    def set_event_frame(self, frame):
        f = _f64(frame)
        assert f.size == self.H * self.W
        _check(lib().eds_pyr_set_event_frame(self._h, _p(f)))

    def build_event_frame(self, x, y, polarity, blur_sigma=0.5, use_exp_weights=True):
        x = np.ascontiguousarray(x, dtype=np.uint16); y = np.ascontiguousarray(y, dtype=np.uint16)
        pol = np.ascontiguousarray(polarity, dtype=np.uint8)
        norm = C.c_double(0.0)
        _check(lib().eds_pyr_build_event_frame(self._h, int(x.shape[0]), x.ctypes.data_as(C.POINTER(C.c_uint16)),
                                               y.ctypes.data_as(C.POINTER(C.c_uint16)), pol.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               float(blur_sigma), int(bool(use_exp_weights)), C.cast(C.byref(norm), _dp)))
        return norm.value

    def level_size(self, level):
        h, w = C.c_int32(0), C.c_int32(0)
        _check(lib().eds_pyr_level_size(self._h, int(level), C.byref(h), C.byref(w)))
        return h.value, w.value

    def level_frame(self, level):
        h, w = self.level_size(level)
        out = np.zeros((h, w))
        _check(lib().eds_pyr_get_level_frame(self._h, int(level), _p(out)))
        return out

    def optimize(self, p, q, v):
        """One call, coarsest level first.  Returns (p, q, v, [info per level, finest first])."""
        p, q, v = _f64(p).copy(), _f64(q).copy(), _f64(v).copy()
        infos = (Info * self.levels)()
        _check(lib().eds_pyr_optimize(self._h, _p(p), _p(q), _p(v), infos))
        return p, q, v, [infos[l].as_dict() for l in range(self.levels)]

    def residuals(self, level):
        r = np.zeros(self._N[level])
        _check(lib().eds_pyr_get_residuals(self._h, int(level), _p(r)))
        return r


def device_count() -> int:
    return int(lib().eds_device_count())
