/* =============================================================================
 * eds_hip.h — C ABI of libeds_hip.so: the MI355X (gfx950) event-to-model
 * photometric tracker of EDS.
 *
 * Drop-in boundary for the hot path of uzh-rpg/slam-eds:
 *   eds::tracking::Tracker::optimize            reference src/tracking/Tracker.cpp:104-241
 *   eds::tracking::PhotometricError::operator() reference src/tracking/PhotometricError.hpp:124-182
 * The reference exposes this path as a plain C++ class (Tracker.hpp:36-114, no
 * FFI).  This header is what a binding of that class would call; the C++ shim
 * `slam-eds_amd/csrc/Tracker.hpp` keeps the reference's member signatures on
 * top of it.
 *
 * Conventions
 *  - plain pointers and sizes only; all host buffers are caller-owned, fp64,
 *    row-major, and only need to live for the duration of the call (the library
 *    copies to HBM) — unlike the reference functor, which keeps raw pointers
 *    into KeyFrame vectors (PhotometricError.hpp:79-83).
 *  - every function returns EDS_OK (0) or a negative eds_status; on failure
 *    in/out pose arguments are left at their input values (Tracker.cpp:217-240).
 *  - a handle owns one HIP stream and is not re-entrant; distinct handles may be
 *    used concurrently.  A handle holds `batch` independent alignments (slots).
 *  - quaternions are stored x,y,z,w (Eigen coeffs(), Tracker.cpp:192);
 *    velocity is [linear(3), angular(3)] (PhotometricError.hpp:114-122);
 *    SE(3) tangents are [upsilon(3), omega(3)] (reference src/sophus/se3.hpp:406-428).
 * ============================================================================= */
#ifndef EDS_HIP_H_
#define EDS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 3): eds_trk_info.flags (was reserved), eds_kf_select.sobel_ksize, eds_event_times, eds_trk_launch_info; new entry points
 * eds_trk_last_launch, eds_trk_prepare_frames, eds_trk_residuals_and_loss, eds_event_times_aos, eds_trk_build_event_frames_aos_timed,
 * eds_pyr_*_batch / _slot, eds_trk_bench_live.  Nothing was removed or re-ordered; a caller built against 1 that zero-initialises
 * its structs keeps working except for eds_kf_select (its `reserved` word became sobel_ksize — same size; 0 there now reads as "3", but
 * call eds_kf_select_default first, as the header always said).
 * 4 (round 4): tuning knobs are per HANDLE — read from the environment once, at eds_trk_create, and changed with eds_trk_set_knob; no
 * entry point reads the environment afterwards.  New entry points: eds_trk_set_knob, eds_trk_get_strips_info, eds_gather_results*
 * (RCCL gather of the result table for a C / C++ caller).  Nothing was removed or re-ordered.
 * 5 (round 5): new entry points eds_trk_bench_kernel_cold, eds_trk_hbm_probe (measurement), eds_trk_set_event_frames (many host frames,
 * narrowed on a thread pool), eds_trk_kernel_instances (the compiled instantiation lists); new knob EDS_LM6_GROUPS (candidate groups
 * of the team kernel).  Nothing was removed or re-ordered.
 * 6 (round 6): new entry points eds_trk_optimize_batch_wait (launch + wait + collect in one call), eds_trk_bench_batch (the batched step
 * timed inside the library); new knob values EDS_REF12_KERNEL=half|full, new knob EDS_POLL_RESULTS; eds_trk_create FAILS
 * (EDS_ERR_INVALID) on an environment variable of a knob's name whose value the knob does not take (it was skipped before); the time-out
 * of a team launch that does not assemble is 5 ms (was 50).  Nothing was removed or re-ordered. */
#define EDS_HIP_ABI_VERSION 6
#define EDS_MAX_LEVELS 8

typedef enum eds_status {
    EDS_OK = 0,
    EDS_ERR_INVALID = -1,       /* bad argument / size (reference: assert only, PhotometricError.hpp:70-73) */
    EDS_ERR_HIP = -2,           /* HIP runtime failure; see eds_last_error() */
    EDS_ERR_NOT_USABLE = -3,    /* solve ended with a non-usable solution (Tracker.cpp:236-239 returns false) */
    EDS_ERR_STATE = -4,         /* keyframe / event frame not set */
    EDS_ERR_NO_DEVICE = -5      /* no gfx950 device visible: there is NO CPU fallback */
} eds_status;

/* how the brightness-increment frame is sampled */
typedef enum eds_sampling {
    EDS_SAMPLE_BICUBIC = 0,     /* ceres::BiCubicInterpolator, what the reference does (PhotometricError.hpp:110-111,172) */
    EDS_SAMPLE_BILINEAR = 1     /* 4-tap variant named by the north-star spec */
} eds_sampling;

/* which problem `optimize` solves */
typedef enum eds_solver {
    EDS_SOLVER_GN6 = 0,         /* pose-only SE(3) Gauss-Newton, velocity fixed: T <- exp(xi) T */
    EDS_SOLVER_LM6 = 1,         /* same with DSO-style damping + accept/reject (template: reference CoarseTracker.cpp:545-664) */
    EDS_SOLVER_REF12 = 2        /* the reference problem: 12 local parameters (t, quaternion, unit velocity),
                                   Ceres trust-region LM semantics (Tracker.cpp:108-143,192-202) */
} eds_solver;

/* where the iteration loop runs */
typedef enum eds_exec {
    EDS_EXEC_HOST = 0,          /* residual/Jacobian kernel + reduction kernel per iteration, small solve on the host */
    EDS_EXEC_DEVICE = 1         /* whole iteration loop on the GPU (one persistent workgroup per alignment) */
} eds_exec;

typedef enum eds_loss {         /* reference tracking/Config.hpp:36 — applied per residual block */
    EDS_LOSS_NONE = 0, EDS_LOSS_HUBER = 1, EDS_LOSS_CAUCHY = 2
} eds_loss;

typedef enum eds_loss_param_method {   /* reference Tracker.hpp:34 */
    EDS_LP_CONSTANT = 0, EDS_LP_MAD = 1, EDS_LP_STD = 2
} eds_loss_param_method;

/* Mirrors eds::tracking::Config / SolverOptions (reference tracking/Config.hpp:40-58). */
typedef struct eds_trk_cfg {
    int32_t device;                 /* HIP device ordinal */
    int32_t sampling;               /* eds_sampling */
    int32_t solver;                 /* eds_solver */
    int32_t exec;                   /* eds_exec */
    int32_t num_blocks;             /* options.num_threads: number of residual blocks, each with its own model norm
                                       (Tracker.cpp:178-195; PhotometricError.hpp:132-152) */
    int32_t loss_type;              /* eds_loss (REF12: per block, Tracker.cpp:146-161) */
    double  loss_param;             /* config.loss_params[0] */
    double  huber_tau;              /* per-point Huber threshold for GN6/LM6 (extension, cf. CoarseTracker.cpp:445); 0 = off */
    double  lambda0;                /* LM6 initial damping (CoarseTracker.cpp:561 uses 0.01) */
    int32_t num_levels;             /* entries used in max_num_iterations */
    int32_t max_num_iterations[EDS_MAX_LEVELS];   /* options.max_num_iterations[id] (Tracker.cpp:139) */
    double  function_tolerance;     /* YAML (Tracker.cpp:140) */
    double  gradient_tolerance;     /* 1e-8 (Tracker.cpp:142) */
    double  parameter_tolerance;    /* 1e-6 (Tracker.cpp:143) */
    int32_t nc;                     /* 1: PhotometricErrorNC residual (PhotometricErrorNC.hpp:124-192): r = w (m/||m|| - E/||E||_block).
                                     * REF12 and eval(ncols = 12) only; build_event_frame then stores the frame un-normalised,
                                     * as that functor requires (EventFrame.cpp:278-281) */
    int32_t reserved[7];
} eds_trk_cfg;

/* Mirrors eds::tracking::TrackerInfo (reference tracking/Config.hpp:60-68) + diagnostics. */
typedef struct eds_trk_info {
    double   meas_time_us;          /* wall time of the solve (Tracker.cpp:209) */
    uint32_t num_points;            /* number of residuals (Tracker.cpp:210) */
    int32_t  num_iterations;        /* successful + unsuccessful steps (Tracker.cpp:211) */
    double   time_seconds;          /* solver-reported time (Tracker.cpp:212) */
    uint8_t  success;               /* IsSolutionUsable (Tracker.cpp:213) */
    uint8_t  flags;                 /* EDS_INFO_* bits below (diagnostics; 0 in the normal case) */
    uint8_t  pad_[2];
    int32_t  termination;           /* 0 convergence, 1 no convergence (max iterations), 2 failure */
    int32_t  num_successful_steps;
    int32_t  num_unsuccessful_steps;
    double   initial_cost;
    double   final_cost;
    double   device_time_us;        /* GPU time of the solve: between two stream events around the launch; for launches of up to 64
                                     * alignments, from the kernels' own 100 MHz time stamps (member 0 of the first team in .. last result
                                     * out: excludes dispatch latency — the two sources differ by a few microseconds;
                                     * eds_trk_last_launch().timing_source says which one a call used) */
} eds_trk_info;

#define EDS_INFO_TEAM_TIMEOUT 1      /* this solve was launched on several CUs per alignment, a team did not assemble within 5 ms (something
                                     * else held the GPU), and the range was solved again with one CU per alignment: the result is valid, the
                                     * call took >= 5 ms, and the handle forms no teams for its next launches (it re-arms by itself) */
#define EDS_INFO_TEAMS_PAUSED 2     /* solved with one CU per alignment because an earlier time-out's cool-down is still running */

typedef struct eds_trk eds_trk;     /* opaque */

/* ---- library ---------------------------------------------------------------------------- */
int         eds_abi_version(void);
int         eds_device_count(void);
const char* eds_last_error(void);               /* thread-local message of the last failure */
void        eds_trk_cfg_default(eds_trk_cfg* cfg);
int         eds_trk_cfg_size(void);             /* sizeof(eds_trk_cfg)  — lets a binding verify its struct layout */
int         eds_trk_info_size(void);            /* sizeof(eds_trk_info) */

/* ---- lifetime --------------------------------------------------------------------------- */
/* Replaces Tracker::Tracker(config) (Tracker.cpp:40-47).  Allocates HBM for `batch`
 * alignments of up to `max_points` points on H x W frames; every slot starts at
 * p = 0, q = identity, v = normalize(0.001 * 1_6) like the reference constructor. */
int  eds_trk_create(const eds_trk_cfg* cfg, int batch, int max_points, int H, int W, eds_trk** out);
void eds_trk_destroy(eds_trk* h);
int  eds_trk_set_config(eds_trk* h, const eds_trk_cfg* cfg);    /* Tracker::config is a public member (Tracker.hpp:40) */
int  eds_trk_get_config(const eds_trk* h, eds_trk_cfg* cfg);

/* ---- inputs ----------------------------------------------------------------------------- */
/* Replaces the pointers PhotometricError::Create receives (Tracker.cpp:189-191):
 * norm_xy, grad_xy are N x 2 AoS (cv::Point2d layout), idp and w have N entries,
 * intrinsics come from kf->K_ref (Tracker.cpp:165-166).  Converted to SoA in HBM. */
int eds_trk_set_keyframe(eds_trk* h, int slot, int N, const double* norm_xy, const double* grad_xy,
                         const double* idp, const double* w, double fx, double fy, double cx, double cy);
/* The reference re-reads the inverse depths on every optimize (Tracker.cpp:167).  Like eds_trk_set_event_frame this returns
 * once the caller's array has been read (it may be reused at once); the device-side part is ordered before whatever the handle
 * does next and is not waited for. */
int eds_trk_set_idepth(eds_trk* h, int slot, int N, const double* idp);
/* The same straight from DepthPoints' own storage: the reference keeps [mu, sigma^2, a, b] per point (std::vector<Eigen::Vector4d>,
 * DepthPoints.hpp:38,53) and getIDepth copies every first element into a fresh vector on each optimize (DepthPoints.cpp:230-237); here
 * the inverse depth of point i is idp[i * stride] (stride in doubles: 4 for that container), no intermediate copy. */
int eds_trk_set_idepth_strided(eds_trk* h, int slot, int N, const double* idp, int stride);
/* Replaces `const std::vector<double>* event_frame` (Tracker.hpp:80): H*W row-major.  Returns as soon as the frame has been
 * narrowed to fp32 into the handle's staging buffer (the caller's buffer is free again); tiling on the device is ordered before the
 * next solve and not waited for. */
int eds_trk_set_event_frame(eds_trk* h, int slot, const double* frame);
int eds_trk_set_event_frame_f32(eds_trk* h, int slot, const float* frame);
/* ABI 5.  MANY host frames in one call: frames[i] (H x W row-major, caller-owned, needed only for the call) goes to slot first + i.
 * The fp64 -> fp32 narrowing runs on a few host threads (knob EDS_UPLOAD_THREADS, default 4) into a ring of pinned staging slots while
 * the store kernels of earlier frames read theirs over PCIe: a batch is bound by PCIe, not by one host thread.  The slots end up
 * bit-identical to `count` calls of eds_trk_set_event_frame (reference: the frames Tracker::optimize is handed, Tracker.hpp:80-81,
 * EventFrame.hpp:59).  Asynchronous on the handle's stream like the one-frame call. */
int eds_trk_set_event_frames(eds_trk* h, int first, int count, const double* const* frames);
int eds_trk_set_event_frames_f32(eds_trk* h, int first, int count, const float* const* frames);
/* Slot `slot` samples slot `src_slot`'s frame storage from now on (no copy) — several alignments against ONE event frame (pose
 * hypotheses, several keyframes).  Alignments that share a frame and are launched together re-use each other's lines in the L2 when
 * their slots are congruent modulo 8 (workgroup b of a launch runs on XCD b % 8).  A later frame written INTO `slot` (set / build)
 * ends the sharing; a frame written into `src_slot` is seen by both.  src_slot == slot: back to the slot's own storage. */
int eds_trk_share_event_frame(eds_trk* h, int slot, int src_slot);
/* ---- event-frame construction on the device ---------------------------------------- */
/* Forward undistortion LUT of the event camera, H x W floats each (EventFrame::fwd_mapx / fwd_mapy,
 * reference src/tracking/EventFrame.cpp:72-79,316-317); NULL, NULL = identity.  Once per handle. */
int eds_trk_set_undistort_map(eds_trk* h, const float* mapx, const float* mapy);
/* Replaces EventFrame::create for out_scale == 1 (EventFrame.cpp:302-389) and
 * drawValuesPoints(undist_coord, pol, H, W, "bilinear", 0.5, true) (src/utils/Utils.cpp:50-122): builds the
 * normalised brightness-increment frame of pyramid `level` (0 = plain; i >= 1 = dilate + erode with a (2i+1)^2 box,
 * EventFrame.cpp:350-357) from `n_events` events given as sensor coordinates x[i], y[i] and polarity[i] (0/1), in
 * time order, straight into slot `slot`'s frame storage.  blur_sigma <= 0 skips the 3x3 Gaussian (reference: 0.5);
 * use_exp_weights selects eds::utils::expWeight (reference: true).  norm_out (optional) receives the Frobenius norm
 * the frame was divided by (EventFrame::norm[level]). */
int eds_trk_build_event_frame(eds_trk* h, int slot, int n_events, const uint16_t* x, const uint16_t* y,
                              const uint8_t* polarity, int level, double blur_sigma, int use_exp_weights, double* norm_out);
/* The same from the reference's own container, an array of structs (EventFrame::create takes `const std::vector<base::samples::Event>&`,
 * EventFrame.hpp:78-85, and reads it->x, it->y, it->polarity, EventFrame.cpp:314-318): `events` points at n_events records of `stride`
 * bytes; x and y are uint16 fields at byte offsets off_x, off_y (2-byte aligned), the polarity is the byte at off_polarity (non-zero =
 * positive).  The records are copied as they are and picked apart on the device: no repacking loop on the caller's side. */
int eds_trk_build_event_frames_aos(eds_trk* h, int first_slot, int num_levels, int n_events, const void* events, int stride, int off_x,
                                   int off_y, int off_polarity, int sensor_H, int sensor_W, double blur_sigma, int use_exp_weights,
                                   double* norms);
/* The time bookkeeping EventFrame::create does on the same container before it draws the frame (EventFrame.cpp:313-335): first_time =
 * events[0].ts, last_time = events[n-1].ts, time = events[n/2].ts (the "median" is the middle ELEMENT of the time-ordered slice), delta_time =
 * last - first; all in the unit of the records' ts field (base::Time: int64 microseconds at byte offset off_ts).  The reference object is
 * stateful here: last_time is assigned only in the `else if` branch of its loop, which a single event never reaches, and clear() does not
 * reset it — a one-event slice keeps the PREVIOUS slice's last_time, and the order check and delta_time use that.  So `last_time` is
 * IN/OUT: pass the previous slice's value (0 for a fresh EventFrame; a zero-initialised struct), get this slice's back (`last_valid` = 0
 * says it was carried over).  A slice whose first time stamp is later than that last is the reference's `throw
 * std::runtime_error("[EVENT_FRAME] FATAL ERROR Event time[0] > event time [N-1]")`: EDS_ERR_INVALID, and nothing is built. */
typedef struct eds_event_times {
    int64_t first_time, last_time, time, delta_time;
    int32_t last_valid, reserved;
} eds_event_times;
int eds_event_times_aos(int n_events, const void* events, int stride, int off_ts, eds_event_times* out);
/* eds_trk_build_event_frames_aos preceded by that bookkeeping, in the reference's order: the time check first (on failure no frame is
 * touched), then the frames.  off_ts: byte offset of the int64 time stamp inside a record (8-byte aligned). */
int eds_trk_build_event_frames_aos_timed(eds_trk* h, int first_slot, int num_levels, int n_events, const void* events, int stride, int off_x,
                                         int off_y, int off_polarity, int off_ts, int sensor_H, int sensor_W, double blur_sigma,
                                         int use_exp_weights, double* norms, eds_event_times* times);
/* The batched tracker's counterpart (BASELINE.json configs[4]: one event frame per alignment): `count` independent event slices into
 * slots first_slot .. first_slot + count - 1 in one pass over the device per 32 slices — slice b's events are elements
 * offsets[b] .. offsets[b + 1] - 1 of x / y / polarity (offsets: count + 1 non-decreasing ints), each slice in time order, all at the
 * handle's H x W.  level, blur_sigma, use_exp_weights as above; norms (optional, count doubles) receives each frame's Frobenius norm.
 * Every frame equals what eds_trk_build_event_frame builds from the same slice. */
int eds_trk_build_event_frame_batch(eds_trk* h, int first_slot, int count, const int* offsets, const uint16_t* x, const uint16_t* y,
                                    const uint8_t* polarity, int level, double blur_sigma, int use_exp_weights, double* norms);
/* EventFrame::create as a whole (EventFrame.cpp:302-389): ALL `num_levels` frames of one event slice from a single vote, into slots
 * first_slot .. first_slot + num_levels - 1 (level i in slot first_slot + i; EventFrame::event_frame[i], norms[i] = EventFrame::norm[i]).
 * sensor_H x sensor_W is the size the events and the undistortion LUT live in; when it differs from the handle's H x W (out_scale != 1)
 * the brightness image is resized the way the reference's call does it (EventFrame.cpp:342-346: `cv::resize(img, img, out_size,
 * cv::INTER_CUBIC)` passes INTER_CUBIC in the `fx` position, so OpenCV interpolates with its default INTER_LINEAR, and with its 2x2
 * block average when both scales are exactly 2).  sensor_H, sensor_W <= 0: the handle's size. */
int eds_trk_build_event_frames(eds_trk* h, int first_slot, int num_levels, int n_events, const uint16_t* x, const uint16_t* y,
                               const uint8_t* polarity, int sensor_H, int sensor_W, double blur_sigma, int use_exp_weights,
                               double* norms);
/* The forward LUT at the sensor's size when that is not the handle's (out_scale != 1). */
int eds_trk_set_undistort_map_sized(eds_trk* h, const float* mapx, const float* mapy, int sensor_H, int sensor_W);
/* Reads slot `slot`'s frame back as H*W row-major doubles (whatever set_event_frame* / build_event_frame stored). */
int eds_trk_get_event_frame(eds_trk* h, int slot, double* frame);

/* Tracker::reset / set / optimize overloads seed px,qx,vx (Tracker.cpp:49-102). */
int eds_trk_set_state(eds_trk* h, int slot, const double p[3], const double q_xyzw[4], const double v[6]);
int eds_trk_get_state(eds_trk* h, int slot, double p[3], double q_xyzw[4], double v[6]);
/* Bulk forms for batches: p is count x 3, q count x 4, v count x 6 (row-major); any may be NULL. */
int eds_trk_set_states(eds_trk* h, int first, int count, const double* p, const double* q_xyzw, const double* v);
int eds_trk_get_states(eds_trk* h, int first, int count, double* p, double* q_xyzw, double* v);
/* Bulk result table of a batch: count x 16 doubles per slot = p[3] q[4] v[6] final_cost iterations success
 * (the row format gathered across GPUs by the multi-GPU driver). */
int eds_trk_get_results(eds_trk* h, int first, int count, double* table16);

/* ---- evaluation (one residual/Jacobian pass + reduction) ------------------------------------ */
/* Evaluates slot `slot` at (p,q,v).  ncols = 6: SE(3) left-perturbation Jacobian
 * [d/d upsilon, d/d omega]; ncols = 12: Ceres local coordinates [t, quaternion-local, velocity-local].
 * r: N raw residuals (what Tracker.cpp:223-230 stores); J: N x ncols row-major;
 * JtJ: ncols x ncols; Jtr: ncols; cost: 1/2 sum r^2 (no loss).  Any output may be NULL. */
int eds_trk_eval(eds_trk* h, int slot, const double p[3], const double q_xyzw[4], const double v[6], int ncols,
                 double* r, double* J, double* JtJ, double* Jtr, double* cost);

/* ---- solve ------------------------------------------------------------------------------ */
/* Replaces Tracker::optimize(id, event_frame, T_kf_ef, method) (Tracker.cpp:104-241) for one slot.
 * p,q,v are in/out; `level` indexes max_num_iterations.  Residuals at the solution are kept for
 * eds_trk_get_residuals / eds_trk_loss_param. */
int eds_trk_optimize(eds_trk* h, int slot, int level, double p[3], double q_xyzw[4], double v[6], eds_trk_info* info);
/* Batched form: solves slots [first, first+count) from their stored states, asynchronously on the
 * handle's stream; eds_trk_sync waits.  Results via eds_trk_get_state / eds_trk_get_info. */
int eds_trk_optimize_batch(eds_trk* h, int level, int first, int count);
int eds_trk_sync(eds_trk* h);
/* ABI 6: eds_trk_optimize_batch + eds_trk_sync in ONE call (what Tracker::optimize does for its single alignment, Tracker.cpp:201-203,
 * for a range of slots): launch, wait, collect.  A caller with several threads in a managed runtime (the Python binding under its
 * interpreter lock) then has no gap of its own between launch and wait. */
int eds_trk_optimize_batch_wait(eds_trk* h, int level, int first, int count);
int eds_trk_get_info(eds_trk* h, int slot, eds_trk_info* info);
/* Per-iteration trace of the last 6-DoF solve of a slot: increments (iters x 6), cost at the
 * candidate (iters), accepted flags (iters).  Returns the number of iterations (<= max_iters) or <0. */
int eds_trk_get_trace(eds_trk* h, int slot, int max_iters, double* increments, double* costs, int32_t* accepted);

/* ---- outputs ---------------------------------------------------------------------------- */
/* kf->residuals after the solve (Tracker.cpp:223-230), N entries. */
int eds_trk_get_residuals(eds_trk* h, int slot, double* r);
/* Tracker::getLossParams (Tracker.cpp:281-317) on the stored residuals; `tau` in/out
 * (CONSTANT leaves it).  Like the reference, MAD partially reorders the stored residuals.  For ONE slot the N residuals
 * are read back (the caller needs them for kf->residuals anyway) and the O(N) selection runs where the reference runs
 * it, on the host: 20 us against 80 us for a single-alignment sort on the GPU.  Batches: eds_trk_loss_param_batch. */
int eds_trk_loss_param(eds_trk* h, int slot, int method, double* tau);

/* Tracker.cpp:223-233 in one call and one read-back: r (N doubles) receives kf->residuals as the reference leaves them after
 * `kf->residuals[i] = ...` AND `config.loss_params = getLossParams(method)` — the MAD selection partially reorders them in place —
 * and *tau the loss scale (unchanged for EDS_LP_CONSTANT). */
int eds_trk_residuals_and_loss(eds_trk* h, int slot, int method, double* r, double* tau);

/* Batched form: tau[count] for slots [first, first+count).  When the residuals of the last solve are still resident
 * in HBM (device-mode solves) the median / MAD selection runs on the GPU — one workgroup per alignment, a most-significant-digit
 * radix SELECT on order-preserving 64-bit keys (median and MAD are order statistics: nothing is sorted) — and only 8 bytes per
 * alignment come back. */
int eds_trk_loss_param_batch(eds_trk* h, int first, int count, int method, double* tau);

/* ---- post-solve point maintenance ---------------------------------------------------- */
/* Replaces Tracker::getCoord(delete_out_point) (reference Tracker.cpp:319-376) for slot `slot` at its current pose:
 * re-projects every point with the RAW inverse depth (:343-351), optionally erases the points that left the frame
 * (xp < 0 || xp > cols || yp < 0 || yp > rows, :354) from every index-aligned per-point plane on the device —
 * order-preserving, like repeated KeyFrame::erasePoint (KeyFrame.cpp:1060-1106) — and returns for the n_kept
 * survivors their new pixel coordinates coord_xy (n x 2), tracks_xy = new - old keyframe pixel (:364-366), the
 * original indices kept_index (so the caller can erase the same entries from the KeyFrame members this library
 * does not hold: patches, bundle_patches, ...), and the mean squared flow (:372) that Tracker::needNewKeyframe
 * (:650-654) tests.  Any output pointer may be NULL; the arrays must hold N entries. */
int eds_trk_update_points(eds_trk* h, int slot, int delete_out_points, double* coord_xy, double* tracks_xy,
                          int32_t* kept_index, int* n_kept, double* mean_sq_flow);

/* The same for slots first .. first + count - 1 in one launch per 64 alignments (one workgroup each).  Outputs of alignment b start at
 * point index b * stride of coord_xy / tracks_xy / kept_index (stride >= the largest point count) and at n_kept[b], mean_sq_flow[b]; with
 * the three point arrays NULL only the culling, the counts and the keyframe criterion are produced (nothing but 16 bytes per
 * alignment comes back). */
int eds_trk_update_points_batch(eds_trk* h, int first, int count, int delete_out_points, int stride, double* coord_xy, double* tracks_xy,
                                int32_t* kept_index, int* n_kept, double* mean_sq_flow);

/* ---- keyframe point set-up on the device -------------------------------------------- */
enum eds_kf_method { EDS_KF_MAX = 0, EDS_KF_MEDIAN = 1 };   /* eds::tracking::CANDIDATE_POINT_METHOD */
enum eds_img_type { EDS_IMG_U8 = 0, EDS_IMG_F32 = 1, EDS_IMG_F64 = 2 };
typedef struct eds_kf_select {
    int32_t method;                 /* EDS_KF_MAX: num_points / n_cells strongest gradients per cell; EDS_KF_MEDIAN: every
                                     * pixel above its cell's median (KeyFrame::candidatePoints, KeyFrame.cpp:740-823) */
    int32_t cell;                   /* cell edge in pixels; the reference uses cv::Size(20, 20) (KeyFrame.cpp:408); <= 32 */
    int32_t num_points;             /* MAX only: the reference passes rows*cols*percent_points/100 (KeyFrame.cpp:406-409) */
    int32_t sobel_ksize;            /* aperture of cv::Sobel: 3 (KeyFrame::create, KeyFrame.cpp:384-385; 0 means 3) or 7 (the KeyFrame constructor,
                                     * KeyFrame.cpp:239-240: kernels [1 6 15 20 15 6 1] x [-1 -4 -5 0 5 4 1], un-normalised — gradients 256 times larger) */
    double  min_depth, max_depth;   /* without a depth map every point starts at idp = 1/((max-min)/2) (KeyFrame.cpp:1186-1192) */
    double  weight_threshold;       /* KeyFrame::cleanPoints(0.7) (KeyFrame.cpp:451,1566-1587) */
} eds_kf_select;
void eds_kf_select_default(eds_kf_select* sel);
/* Replaces the tracker-facing part of KeyFrame::create (reference src/tracking/KeyFrame.cpp:333-463): grayscale image
 * (H x W of the handle, row-major, already undistorted) -> [0,1] -> log(img + 0.2) -> Sobel 3x3 -> gradient magnitude ->
 * per-cell point selection -> norm_coord, grad -> nearest depth-map point (n_depth points depth_xy (n x 2, pixels) with
 * inverse depth depth_idp; n_depth = 0: constant initial depth) -> weights -> cleanPoints.  The surviving points go
 * straight into slot `slot` as by eds_trk_set_keyframe (same order as the reference pushes them); *n_points receives
 * their number (also when it exceeds the handle's max_points, which is an error). */
int eds_trk_build_keyframe(eds_trk* h, int slot, int img_type, const void* img, const eds_kf_select* sel, int n_depth,
                           const double* depth_xy, const double* depth_idp, double fx, double fy, double cx, double cy,
                           int* n_points);
/* The same from the image as the camera driver hands it over: img_H x img_W pixels with `channels` interleaved channels (1 grey,
 * 3 RGB; EDS_IMG_U8 / EDS_IMG_F32 — OpenCV's cvtColor takes no CV_64F colour image).  A size other than the handle's H x W is the
 * reference's out_scale != 1: the image is resized first (KeyFrame.cpp:352-357; the call `cv::resize(img, img, out_size,
 * cv::INTER_CUBIC)` passes INTER_CUBIC as `fx`, so OpenCV interpolates with INTER_LINEAR, and with the 2x2 block mean when both scales
 * are exactly 2), then converted to grey (cv::cvtColor COLOR_RGB2GRAY, :361), each with OpenCV's arithmetic for the element type.
 * fx, fy, cx, cy are the intrinsics of the RESIZED image (kf->K_ref). */
int eds_trk_build_keyframe_image(eds_trk* h, int slot, int img_type, const void* img, int img_H, int img_W, int channels,
                                 const eds_kf_select* sel, int n_depth, const double* depth_xy, const double* depth_idp,
                                 double fx, double fy, double cx, double cy, int* n_points);
/* The index-aligned vectors KeyFrame keeps (coord, norm_coord, grad, inv_depth, weights; KeyFrame.hpp:80-96) of the
 * keyframe eds_trk_build_keyframe built last; any pointer may be NULL. */
int eds_trk_get_keyframe_points(eds_trk* h, int slot, double* coord_xy, double* norm_xy, double* grad_xy, double* idp,
                                double* weights);

/* ---- coarse-to-fine tracking on an image pyramid (BASELINE.json configs[3]) ------------------------------- */
/* An extension patterned on the DSO-derived coarse tracker the reference carries: levels by 2x2 box averaging (reference
 * src/tracking/HessianBlocks.cpp:173-176), level intrinsics fx_l = fx_{l-1} / 2, cx_l = (cx_0 + 0.5) / 2^l - 0.5
 * (src/tracking/CoarseTracker.cpp:103-111), coarsest level first with the pose carried down (CoarseTracker.cpp:545-664).  One
 * handle holds `levels` frame sizes H >> l x W >> l and one point set per level (max_points[l] entries at most); the event frame
 * is handed over (or built from events) once, at level 0, and the coarser frames are made on the device. */
typedef struct eds_pyr eds_pyr;
int  eds_pyr_create(const eds_trk_cfg* cfg, int levels, const int* max_points, int H, int W, eds_pyr** out);
void eds_pyr_destroy(eds_pyr* p);
int  eds_pyr_set_config(eds_pyr* p, const eds_trk_cfg* cfg);
/* K[4] = fx, fy, cx, cy of `level` from the level-0 intrinsics (CoarseTracker.cpp:103-111) */
int  eds_pyr_level_intrinsics(int level, double fx0, double fy0, double cx0, double cy0, double K[4]);
/* the points tracked at `level` (arrays as eds_trk_set_keyframe; normalised coordinates do not depend on the level);
 * intrinsics are those of LEVEL 0 */
int  eds_pyr_set_keyframe(eds_pyr* p, int level, int N, const double* norm_xy, const double* grad_xy, const double* idp,
                          const double* w, double fx0, double fy0, double cx0, double cy0);
/* level-0 frame (H x W row-major doubles, as eds_trk_set_event_frame) or events (as eds_trk_build_event_frame, level 0 of
 * EventFrame::create); levels 1 .. L-1 follow on the device */
int  eds_pyr_set_event_frame(eds_pyr* p, const double* frame);
int  eds_pyr_build_event_frame(eds_pyr* p, int n_events, const uint16_t* x, const uint16_t* y, const uint8_t* polarity,
                               double blur_sigma, int use_exp_weights, double* norm_out);
int  eds_pyr_level_size(const eds_pyr* p, int level, int* H, int* W);
int  eds_pyr_get_level_frame(eds_pyr* p, int level, double* frame);     /* (H >> level) x (W >> level) doubles */
/* One call: levels L-1 .. 0, `max_num_iterations[level]` iterations each, (p, q, v) in/out and carried from level to level;
 * infos (optional) receives one eds_trk_info per level.  Returns the status of the finest level. */
int  eds_pyr_optimize(eds_pyr* p, double pose_p[3], double q_xyzw[4], double v[6], eds_trk_info* infos);
int  eds_pyr_get_residuals(eds_pyr* p, int level, double* r);
/* `batch` independent pyramids in one object (level l of all of them = one handle of `batch` slots = one launch per level):
 * the calls above address pyramid 0; these take the pyramid (slot).  eds_pyr_optimize_batch: P, Q, V are count x 3 / 4 / 6, in and
 * out; infos (optional) levels x count, level-major. */
int  eds_pyr_create_batch(const eds_trk_cfg* cfg, int batch, int levels, const int* max_points, int H, int W, eds_pyr** out);
int  eds_pyr_set_keyframe_slot(eds_pyr* p, int slot, int level, int N, const double* norm_xy, const double* grad_xy, const double* idp,
                               const double* w, double fx0, double fy0, double cx0, double cy0);
int  eds_pyr_set_event_frame_slot(eds_pyr* p, int slot, const double* frame);
int  eds_pyr_optimize_batch(eds_pyr* p, int first, int count, double* P, double* Q, double* V, eds_trk_info* infos);

/* ---- measurement ------------------------------------------------------------------------ */
/* HIP events on the handle's own stream (torch.cuda.Event cannot see it). */
int eds_trk_timer_start(eds_trk* h);
int eds_trk_timer_stop(eds_trk* h, float* elapsed_ms);      /* synchronises the stream */
/* Launches the residual/Jacobian kernel (and optionally the reduction kernel) over slots
 * [first, first+count) `reps` times back-to-back at the stored states and reports the mean
 * duration per launch in ms, measured with HIP events on the handle's stream. */
int eds_trk_bench_eval(eds_trk* h, int first, int count, int ncols, int with_reduction, int reps, float* mean_ms);
/* ABI 5.  ONE kernel of the streaming path timed COLD: before every repetition 1 GiB is READ through the caches (the Infinity
 * Cache holds 256 MB), then the kernel runs between its own pair of HIP events; mean over `reps`.  which: 0 the residual/Jacobian
 * kernel, 1 the reduction kernel over the planes of a residual/Jacobian pass made beforehand. */
int eds_trk_bench_kernel_cold(eds_trk* h, int first, int count, int ncols, int which, int reps, float* mean_ms);
/* ABI 5.  What the box's HBM streams through the library's OWN plain kernel (16 bytes per lane, grid-stride) over `bytes` of scratch:
 * a read-only pass and a copy (read + write counted), GB/s, HIP events on the handle's stream (the measured peak beside
 * the nominal 8 TB/s). */
int eds_trk_hbm_probe(eds_trk* h, size_t bytes, int reps, float* read_GBps, float* copy_GBps);
/* Latency of the live sequence measured INSIDE the library's language (no interpreter between the calls): `reps` times
 *   [eds_trk_set_idepth(idp)] -> [eds_trk_set_event_frame(frame)] -> eds_trk_optimize(level, p0, q0, v0) -> [eds_trk_residuals_and_loss(method)]
 * on `slot` (Tracker.cpp:167 -> EventFrame -> :104-241 -> :223-233), each from the same start state; the bracketed calls are skipped when
 * their pointer is NULL (`method` < 0 skips the last).  out_us[6] = medians over the repetitions of { whole sequence, set_idepth,
 * set_event_frame, optimize, residuals_and_loss, device time of the solve }, all in microseconds (std::chrono::steady_clock around
 * each call).  The state of the slot after the call is that of the last repetition. */
int eds_trk_bench_live(eds_trk* h, int slot, int level, const double* idp, const double* frame, const double p0[3], const double q0[4],
                       const double v0[6], int method, int reps, double out_us[6]);
/* ABI 6.  The BATCHED step timed inside the library, as eds_trk_bench_live times the single call: reps x { eds_trk_set_states(first, count,
 * p, q, v); eds_trk_optimize_batch_wait(level, first, count) } with std::chrono around each — what a C / C++ caller pays per step of
 * `count` alignments, without a binding's own work between the calls.  out_us = { median step, median set_states, median solve call,
 * median kernel (eds_trk_info.device_time_us), slowest step }. */
int eds_trk_bench_batch(eds_trk* h, int level, int first, int count, const double* p, const double* q_xyzw, const double* v, int reps,
                        double out_us[5]);
/* What the last on-device solve (eds_trk_optimize / _optimize_batch with exec = device) actually launched — so that a benchmark
 * prices the kernel that ran instead of mirroring the library's selection rule. */
/* The persistent kernels (and the streaming residual/Jacobian kernel on batches) gather from STRIP COPIES of the event frames
 * (csrc/eds_layout.hpp): 8-column-wide strips stored 2 x `phases` times so that every bicubic 4x4 patch is ONE 128-byte L2 line.
 * The frame writers keep writing the 4x4 tiles.  The copies cost more than ONE solve gains from them (10 MB written per 640x480 frame
 * with 4 phases: 2.9 us of the GPU, against 0.26 us gained per 2 000-point solve), so the library makes them for frames that are
 * solved AGAIN: the first solve on a new frame samples the tiles, the second one that finds the same frame converts it; a few new
 * frames among many converted ones are converted at once (csrc/eds_strips.hip; EDS_STRIPS_POLICY=eager|never overrides).  This
 * call makes the copies NOW for slots [first, first + count) — for frames that WILL be solved many times (batches that are
 * re-solved, parameter sweeps, benchmarks over resident inputs), or (force != 0: convert even what is current) to measure the
 * conversion: elapsed_ms (optional) = HIP events around the conversion launches.  No-op for the row-major layout. */
int eds_trk_prepare_frames(eds_trk* h, int first, int count, int force, float* elapsed_ms);

/* Tuning knobs (A/B runs and tests; the defaults are the measured best).  They belong to the HANDLE: eds_trk_create reads the process
 * environment once (variables of the same names), eds_trk_set_knob changes one knob of one handle afterwards — two handles with
 * different knobs coexist, and no solve touches getenv().  value NULL or "" restores the default.  Names and values (csrc/eds_launch_rule.hpp
 * holds the rule they steer): EDS_REF12_EXEC=device|host, EDS_FUSED_THREADS=64..1024, EDS_FUSED_PPT=0|1|2|4, EDS_LM6_SPEC=0|1,
 * EDS_LM6_KERNEL=resident|paired|wide, EDS_FUSED_LAYOUT=tiles|strips, EDS_FUSED_GATHER=quad|lane, EDS_LM6_TEAM / EDS_REF12_TEAM=1|2|4|8|16,
 * EDS_TEAM_WIDE=0|1, EDS_REF12_KERNEL=wide|paired|half|full (ABI 6: half = the paired shape with 736 cache slots per alignment, full = one alignment per CU with a slot for every point; one residual block, <= 2 000 points), EDS_STRIPS_PHASES=1|2|4, EDS_STRIPS_POLICY=reuse|eager|never,
 * EDS_STRIPS_BUDGET_PCT=1..95 (share of the FREE device memory the strip copies may take when they are first allocated; default 50),
 * EDS_REDUCE_PPL=4|8 (points a lane of the 6-column reduction folds: measured equal), EDS_NO_SPIN, EDS_POLL_RESULTS=0|1 (ABI 6: 0 waits for small solves through the stream instead of their workgroups' completion words in pinned memory), EDS_UPLOAD=bands, EDS_FUSED_REPORT,
 * EDS_TEAM_TEST_DROP_MEMBER (test hook); round 5: EDS_LM6_GROUPS=1|2|4|8 (candidate groups of the team kernel), EDS_UPLOAD_THREADS=1..64,
 * EDS_REF12_GROUPS=1|2|4 (the same for the REF12 team kernel), EDS_UPLOAD_DMA=0|1, EDS_UPLOAD_STREAMS=1|2 (eds_trk_set_event_frames), EDS_FORCE_FUSED6 / EDS_FORCE_FUSED12 (below).
 * EDS_FRAME_LAYOUT=rowmajor decides the allocation and is honoured at create only (EDS_ERR_STATE here).  Unknown name, or a value the
 * knob does not take: EDS_ERR_INVALID (nothing changes).  While a batch is in flight (eds_trk_optimize_batch without wait): EDS_ERR_STATE.
 * An EDS_STRIPS_PHASES / EDS_STRIPS_BUDGET_PCT change frees the strip copies: the next solve that wants them allocates them anew. */
int eds_trk_set_knob(eds_trk* h, const char* name, const char* value);
/* ABI 5.  The instantiations of the persistent kernels the library was compiled with (the launchers dispatch over exactly these lists).
 * family 0: eds_fused6_kernel<S, P, T, Q, K, G>, 1: eds_fused12_kernel<S, T, CAP, NC, K, Q>, 2: its candidate-group instantiations as
 * {S, T, NC, K, Q, G} (CAP = 512 for all of them; reached with EDS_FORCE_FUSED12 + EDS_REF12_GROUPS).  Returns how many the family has (-1:
 * no such family); with 0 <= index < count and args6 != NULL, writes the six template arguments of instantiation `index`.  The knobs
 * EDS_FORCE_FUSED6 / EDS_FORCE_FUSED12 = "a,b,c,d,e,f" launch one of them wherever it can solve the range (test hooks: tests/
 * test_instances_gpu.py checks every one against the CPU oracle). */
int eds_trk_kernel_instances(int family, int index, int32_t* args6);
/* The strip copies of the frames (csrc/eds_layout.hpp): bytes allocated (0: none yet), their row phases, and whether they were refused
 * because they did not fit the budget (the solves then sample the tiles; setting an EDS_STRIPS_* knob re-arms the allocation). */
int eds_trk_get_strips_info(eds_trk* h, int64_t* bytes, int32_t* row_phases, int32_t* unavailable);

typedef struct eds_trk_launch_info {
    char    kernel[96];             /* e.g. "eds_fused6_kernel<0, 4, 512, 3, 1>": name and template arguments as rocprofv3 prints them */
    int32_t workgroups;             /* of the last launch of the call (a range may go out in several team launches) */
    int32_t cus_per_alignment;      /* > 1: a team launch */
    int32_t first, count;           /* the range solved */
    int32_t layout;                 /* frame layout the kernel sampled: 0 row-major, 1 4x4 tiles, 2 strips (csrc/eds_layout.hpp) */
    int32_t timing_source;          /* device_time_us of eds_trk_info: 0 = two stream events around the launch(es), 1 = the workgroups' own
                                     * 100 MHz stamps (launches of <= 64 alignments: member 0 of each team stamps when IT starts, so the value
                                     * excludes dispatch latency and is not comparable with source 0 to better than a few microseconds) */
    /* digest of the workgroups' begin / end stamps (one CU per alignment only; zeros otherwise): */
    double  span_us;                /* first workgroup in .. last workgroup out */
    double  mean_workgroup_us;
    double  covered;                /* sum of workgroup durations / (256 CUs x span): what the launch tail and dispatch gaps leave */
    double  tail_idle_us;           /* mean idle time of a CU between its last workgroup's end and the end of the launch */
} eds_trk_launch_info;
int eds_trk_last_launch(eds_trk* h, eds_trk_launch_info* out);

#ifdef __cplusplus
}
#endif
#endif /* EDS_HIP_H_ */
