"""Python mirror of ``eds::tracking::Tracker`` (reference src/tracking/Tracker.hpp:36-114) over the
C ABI — same member names, argument meaning and error behaviour for the alignment path
(``optimize``, ``getLossParams``, ``getTransform``, ``set``/``reset``, ``getInfo``).  The KLT /
epipolar helpers of the reference class (Tracker.cpp:378-654) are outside the hot path and are
not mirrored.  The C++ twin with the reference's exact signatures is ``csrc/Tracker.hpp``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import capi

# eds::tracking::LOSS_PARAM_METHOD (Tracker.hpp:34)
CONSTANT, MAD, STD = capi.LP_CONSTANT, capi.LP_MAD, capi.LP_STD
# eds::tracking::LOSS_FUNCTION (tracking/Config.hpp:36)
NONE, HUBER, CAUCHY = capi.LOSS_NONE, capi.LOSS_HUBER, capi.LOSS_CAUCHY
# eds::tracking::CANDIDATE_POINT_METHOD
SELECT_MAX, SELECT_MEDIAN = capi.KF_MAX, capi.KF_MEDIAN


@dataclass
class SolverOptions:
    """eds::tracking::SolverOptions (tracking/Config.hpp:40-47)."""
    linear_solver_type: str = "SPARSE_NORMAL_CHOLESKY"   # accepted for compatibility; the 12x12 solve is dense
    num_threads: int = 1                                 # = number of residual blocks (Tracker.cpp:178-195)
    max_num_iterations: List[int] = field(default_factory=lambda: [10])
    function_tolerance: float = 1e-6
    minimizer_progress_to_stdout: bool = False


@dataclass
class Config:
    """eds::tracking::Config (tracking/Config.hpp:49-58) plus the GPU-side switches."""
    percent_points: float = 0.0
    type: str = "ceres"
    loss_type: int = NONE
    loss_params: List[float] = field(default_factory=lambda: [1.0])
    options: SolverOptions = field(default_factory=SolverOptions)
    # extensions (not in the reference struct)
    solver: int = capi.SOLVER_REF12          # REF12 reproduces the reference; GN6/LM6 are the pose-only solvers
    sampling: int = capi.SAMPLE_BICUBIC
    exec: int = capi.EXEC_DEVICE
    huber_tau: float = 0.0
    nc: bool = False                         # PhotometricErrorNC instead of PhotometricError (Tracker.cpp:25-27 toggle)
    lambda0: float = 0.01
    device: int = 0


@dataclass
class KeyFrame:
    """The members of eds::tracking::KeyFrame the tracker reads (KeyFrame.hpp:60-96)."""
    norm_coord: np.ndarray          # N x 2
    grad: np.ndarray                # N x 2
    weights: np.ndarray             # N
    inv_depth: np.ndarray           # N   (DepthPoints::getIDepth, mapping/DepthPoints.cpp:230-237)
    K_ref: np.ndarray               # 3 x 3
    rows: int                       # kf->img.rows
    cols: int                       # kf->img.cols
    residuals: np.ndarray = field(default_factory=lambda: np.zeros(0))
    coord: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))   # pixel coordinates (KeyFrame.hpp:80)

    @classmethod
    def create(cls, img, K_ref, depth_xy=None, depth_idp=None, points_selection_method: int = SELECT_MEDIAN,
               min_depth: float = 1.0, max_depth: float = 3.0, percent_points: float = 0.0, device: int = 0) -> "KeyFrame":
        """The tracker-facing part of KeyFrame::create (KeyFrame.cpp:333-463) on the GPU: log image, Sobel, per-cell
        point selection, depth association against the depth map (``depth_xy`` pixels, ``depth_idp``), cleanPoints(0.7).
        Like the reference, a point target outside (0, rows*cols) falls back to MEDIAN (:406-411)."""
        img = np.asarray(img)
        rows, cols = img.shape
        K = np.asarray(K_ref, dtype=np.float64)
        target = rows * cols * (percent_points / 100.0)
        method, num = (points_selection_method, int(target)) if 0 < target < rows * cols else (SELECT_MEDIAN, 0)
        h = capi.Handle(capi.default_config(device=device), 1, rows * cols, rows, cols)
        try:
            out = h.build_keyframe(0, img, (K[0, 0], K[1, 1], K[0, 2], K[1, 2]), method=method, num_points=num,
                                   depth_xy=depth_xy, depth_idp=depth_idp, min_depth=min_depth, max_depth=max_depth)
        finally:
            h.close()
        return cls(out["norm_coord"], out["grad"], out["weights"], out["idp"], K, rows, cols, coord=out["coord"])


@dataclass
class TrackerInfo:
    """eds::tracking::TrackerInfo (tracking/Config.hpp:60-68)."""
    meas_time_us: float = 0.0
    num_points: int = 0
    num_iterations: int = 0
    time_seconds: float = 0.0
    success: bool = False


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _R_to_quat(R):
    """Eigen::Quaterniond(Matrix3d) (Shepperd's method), xyzw."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w, x, y, z = 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        v = [0.0, 0.0, 0.0]
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        w = (R[k, j] - R[j, k]) / s
        x, y, z = v
    return np.array([x, y, z, w])


class Tracker:
    """Event-to-model alignment driver; one instance owns one GPU handle (one alignment slot)."""

    def __init__(self, kf_or_config, config: Optional[Config] = None):
        if config is None:                       # Tracker(const Config&)          Tracker.hpp:65
            kf, config = None, kf_or_config
        else:                                    # Tracker(shared_ptr<KeyFrame>, const Config&)  :62
            kf = kf_or_config
        self.config = config                     # public member, Tracker.hpp:40
        self.kf: Optional[KeyFrame] = kf
        self.px = np.zeros(3)                                  # Tracker.cpp:43
        self.qx = np.array([0.0, 0.0, 0.0, 1.0])               # :44 (xyzw)
        self.vx = np.full(6, 0.001) / np.linalg.norm(np.full(6, 0.001))   # :45-46
        self.info = TrackerInfo()
        self._h: Optional[capi.Handle] = None
        self._kf_uploaded = None

    # -- state -------------------------------------------------------------------------
    def reset(self, kf: KeyFrame, px, qx, keep_velo_or_velo=True):
        """Tracker.cpp:49-72 — both overloads (bool keep_velo | Vector6d velo)."""
        self.kf = kf
        self.px = np.asarray(px, dtype=np.float64).copy()
        self.qx = np.asarray(qx, dtype=np.float64).copy()
        if isinstance(keep_velo_or_velo, (bool, np.bool_)):
            if not keep_velo_or_velo:
                self.vx = np.full(6, 0.001) / np.linalg.norm(np.full(6, 0.001))
        else:
            self.vx = np.asarray(keep_velo_or_velo, dtype=np.float64).copy()
        self._kf_uploaded = None

    def set(self, T_kf_ef):
        """Tracker.cpp:74-79: stores the INVERSE (the tracker works with T_ef_kf)."""
        Ti = np.linalg.inv(np.asarray(T_kf_ef, dtype=np.float64))
        self.px = Ti[:3, 3].copy()
        self.qx = _R_to_quat(Ti[:3, :3])

    def getTransform(self):
        """Tracker.cpp:243-249: T_ef_kf = SE3(qx, px) as a 4x4."""
        T = np.eye(4)
        T[:3, :3] = _quat_to_R(self.qx / np.linalg.norm(self.qx))
        T[:3, 3] = self.px
        return T

    def getVelocity(self):
        return self.vx

    def linearVelocity(self):
        return self.vx[:3].copy()

    def angularVelocity(self):
        return self.vx[3:].copy()

    def getInfo(self) -> TrackerInfo:
        return self.info

    # -- solve -------------------------------------------------------------------------
    def _cfg(self) -> capi.Cfg:
        c, o = self.config, self.config.options
        return capi.default_config(device=c.device, sampling=c.sampling, solver=c.solver, exec=c.exec,
                                   num_blocks=max(1, int(o.num_threads)), loss_type=int(c.loss_type),
                                   loss_param=float(c.loss_params[0]) if c.loss_params else 1.0,
                                   huber_tau=float(c.huber_tau), lambda0=float(c.lambda0), nc=int(bool(c.nc)),
                                   num_levels=len(o.max_num_iterations), max_num_iterations=list(o.max_num_iterations),
                                   function_tolerance=float(o.function_tolerance))

    def _ensure_handle(self):
        kf = self.kf
        if kf is None:
            raise capi.EdsError(capi.ERR_STATE, "Tracker has no keyframe")
        N = int(np.asarray(kf.inv_depth).shape[0])
        if self._h is None or self._h.max_points < N or (self._h.H, self._h.W) != (kf.rows, kf.cols):
            if self._h is not None:
                self._h.close()
            cap = max(N, 2048)
            self._h = capi.Handle(self._cfg(), 1, cap, kf.rows, kf.cols)
            self._kf_uploaded = None
        else:
            self._h.set_config(self._cfg())

    def optimize(self, id: int, event_frame, T_kf_ef=None, *, px=None, qx=None, vx=None,
                 loss_param_method: int = MAD):
        """Tracker::optimize (Tracker.cpp:104-241; overloads :81-102 via px/qx/vx keywords).

        Returns ``(ok, T_kf_ef)``; on ``ok == False`` nothing is updated (T_kf_ef is returned
        unchanged), exactly like the reference's ``return false`` branch.
        """
        if px is not None:
            self.px = np.asarray(px, dtype=np.float64).copy()
        if qx is not None:
            self.qx = np.asarray(qx, dtype=np.float64).copy()
        if vx is not None:
            self.vx = np.asarray(vx, dtype=np.float64).copy()
        self._ensure_handle()
        kf, h = self.kf, self._h
        K = np.asarray(kf.K_ref, dtype=np.float64)
        # the reference re-reads every vector on each call (Tracker.cpp:164-167,189-191)
        h.set_keyframe(0, kf.norm_coord, kf.grad, kf.inv_depth, kf.weights, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        h.set_event_frame(0, event_frame)
        try:
            p, q, v, info = h.optimize(0, level=id, p=self.px, q=self.qx, v=self.vx)
        except capi.EdsError as e:
            if e.code != capi.ERR_NOT_USABLE:
                raise
            i = h.info(0)
            self.info = TrackerInfo(i["meas_time_us"], i["num_points"], i["num_iterations"], i["time_seconds"], False)
            return False, T_kf_ef
        self.px, self.qx, self.vx = p, q, v
        self.info = TrackerInfo(info["meas_time_us"], info["num_points"], info["num_iterations"],
                                info["time_seconds"], bool(info["success"]))
        kf.residuals = h.residuals(0)                               # Tracker.cpp:223-230
        self.config.loss_params = self.getLossParams(loss_param_method)   # :233
        return True, np.linalg.inv(self.getTransform())             # :220

    def getLossParams(self, method: int = CONSTANT):
        """Tracker.cpp:281-317.  MAD partially reorders kf.residuals in place like the reference."""
        if method == CONSTANT or self._h is None:
            return list(self.config.loss_params)
        tau = self._h.loss_param(0, method, self.config.loss_params[0] if self.config.loss_params else 0.0)
        self.kf.residuals = self._h.residuals(0)
        return [tau]

    def getCoord(self, delete_out_point: bool = False):
        """Tracker::getCoord (Tracker.cpp:319-376): warped active points in the event frame.  With
        ``delete_out_point`` the points that left the frame are erased from every index-aligned KeyFrame vector
        (KeyFrame::erasePoint, KeyFrame.cpp:1060-1106) — on the device and, through the returned index list, here."""
        self._ensure_handle()
        kf, h = self.kf, self._h
        K = np.asarray(kf.K_ref, dtype=np.float64)
        h.set_keyframe(0, kf.norm_coord, kf.grad, kf.inv_depth, kf.weights, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        h.set_state(0, self.px, self.qx, self.vx)
        out = h.update_points(0, delete_out_point)
        keep = out["kept"]
        if len(keep) != len(kf.inv_depth):
            for name in ("norm_coord", "grad", "weights", "inv_depth"):
                setattr(kf, name, np.ascontiguousarray(np.asarray(getattr(kf, name))[keep]))
            if len(kf.residuals) == 0 or len(kf.residuals) != len(keep):
                kf.residuals = np.zeros(0)
        self.tracks = out["tracks"]                                  # kf->tracks (Tracker.cpp:365)
        self.squared_norm_flow = out["mean_sq_flow"]                 # :372
        return out["coord"]

    def needNewKeyframe(self, weight_factor: float = 0.03) -> bool:
        """Tracker::needNewKeyframe (Tracker.cpp:650-654)."""
        rows, cols = self.kf.rows, self.kf.cols
        image_weight = (cols + rows) * weight_factor
        return bool(image_weight * float(np.sqrt(np.float32(getattr(self, "squared_norm_flow", 0.0)))) / (cols + rows) > 1)

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None
