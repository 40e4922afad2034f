// Persistent on-device solve (EDS_EXEC_DEVICE): one workgroup per alignment runs the whole
// iteration loop — residual+Jacobian pass, wavefront reduction, small dense solve, pose update,
// accept test — without returning to the host between iterations.
//   eds_fused.hip    pose-only solvers (GN6 / LM6), point constants in registers: the lowest latency
//   eds_stream6.hip  the same with constants re-read per pass, in two shapes: two alignments per CU (large batches) and
//                    one 512-thread workgroup (keyframes with more than 2 048 points)
//   eds_fused12.hip  the reference problem (REF12: 12 local parameters, Ceres-LM semantics), same two shapes
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct eds_trk;

struct EdsFusedIn {            // start state of a slot
    double p[3], q[4], v[6], pad[3];
};
struct EdsFusedOut {           // compact result of a pose-only solve (trace stays in HBM until asked for)
    double p[3], q[4];
    double initial_cost, final_cost;
    int32_t iterations, ntrace, failed, naccepted;
    double pad[3];
    unsigned long long t_begin, t_end;   // s_memrealtime (100 MHz) when the alignment's (first) workgroup started / finished: the device time
                                         // of small launches without the two event packets around the kernel (each costs microseconds)
};
struct EdsFused12Out {         // compact result of a REF12 solve
    double p[3], q[4], v[6];
    double initial_cost, final_cost;
    int32_t termination, num_successful, num_unsuccessful, failed;
    double pad;
    unsigned long long t_begin, t_end;   // as EdsFusedOut
};

// ---- teams: several workgroups (CUs) per alignment, partial sums exchanged as tagged 8-byte granules (eds_fused.hip) -----------
#define EDS_TEAM_GRANULES 64                      // LM6, per member and parity: 56 used (28 doubles as two halves), padded to one 512-byte block
#define EDS_TEAM_TIMEOUT_TICKS 500000ull          // 5 ms of s_memrealtime (round 6: 50 before — a tracker cannot take a 50 ms stall on a 0.1 ms solve; tests/test_contention_gpu.py)
#define EDS_TEAM_COOLDOWN 16                      // solves without teams after a time-out (doubling up to EDS_TEAM_COOLDOWN_MAX while they recur)
#define EDS_TEAM_COOLDOWN_MAX 1024
#define EDS_TEAM_REARM_CLEAN 64                   // clean team launches after which a new time-out counts as a first one again
#define EDS_TEAM6_MAX 16                          // LM6: up to 16 CUs per alignment (16 384 points)
#define EDS_TEAM_SLOTS 128                        // up to 2 048 points: teams of two up to this many alignments per launch
#define EDS_TEAM_MEMBERS 4096                     // workgroups (alignments x team size) one team LAUNCH holds: the mailboxes' capacity; larger ranges go out in several launches
#define EDS_TEAM_MAIL_BYTES ((size_t)EDS_TEAM_MEMBERS * 2 * EDS_TEAM_GRANULES * 8)
#define EDS_TEAM12_VALUES 157                     // REF12, per residual block: ||r||^2, J^T J (144), J^T r (12)
#define EDS_TEAM12_GRANULES 2560                  // per member and parity: 2 x 157 x 8 blocks = 2 512, padded
#define EDS_TEAM12_SLOTS 64                       // a REF12 team launch holds at most this many alignments ...
#define EDS_TEAM12_MEMBERS 512                    // ... and at most this many workgroups (alignments x team size, up to 16 CUs each)
#define EDS_TEAM12_MAIL_BYTES ((size_t)EDS_TEAM12_MEMBERS * 2 * EDS_TEAM12_GRANULES * 8)

struct EdsFusedBuffers {
    EdsFusedIn* d_in = nullptr;             // device-side addresses of the pinned h_in / h_out / h_out12 below
    EdsFusedOut* d_out = nullptr;
    EdsFused12Out* d_out12 = nullptr;
    void* d_sv = nullptr;          // edss::Solver6 per slot (full state incl. trace)
    EdsFusedIn* h_in = nullptr;    // pinned
    EdsFusedOut* h_out = nullptr;  // pinned
    EdsFused12Out* h_out12 = nullptr;
    unsigned long long* d_mail = nullptr;   // team launches: tagged 8-byte granules, [team slot][parity][member][64]
    unsigned long long* d_mail12 = nullptr; // the same for REF12 (allocated at the first REF12 team launch)
    int* d_ticket = nullptr;                // team launches: workgroup arrival counter (team = ticket / K, member = ticket % K)
    unsigned epoch = 0;                     // launch sequence number inside the granule tags
    unsigned ticket_base = 0;               // tickets handed out by earlier team launches (the device counter is never reset)
    // Time-out policy of the team launches: a time-out (eds_fused_collect / eds_fused12_collect) re-runs the range with one CU per
    // alignment and PAUSES teams for `team_cooldown` further solves of this handle — 16 after the first time-out, doubled by every
    // time-out that follows a re-arm without EDS_TEAM_REARM_CLEAN clean team launches in between (up to 1 024), back to 16 after that
    // many.  One stall caused by something else on the GPU therefore costs the latency regime a few calls, not the handle's lifetime.
    int team_cooldown = 0;                  // solves left without teams
    int team_backoff = 0;                   // the cool-down the next time-out will impose (0: the default)
    int team_clean = 0;                     // clean team launches since the last re-arm
    bool pending_retry = false;             // the launch in flight is the one-CU re-run of a timed-out team launch
    bool pending_paused = false;            // the launch in flight would have used teams but for the cool-down
    int pending_team = 1, pending_level = 0;
    bool pending_ticks = false;    // device time from the kernels' own time stamps (no event records around the launch)
    unsigned *h_done = nullptr, *d_done = nullptr;   // pinned, device-mapped [B][EDS_DONE_WORDS]: one completion word per workgroup of a small solve
    unsigned done_seq = 0;         // the tag of the launch in flight (never 0)
    int pending_vteam = 0;         // > 0: the launch in flight reports through the done words, this many workgroups per alignment
    bool pending_host_r = false;   // the launch in flight mirrors its residuals into the handle's h_rmap (eds_mirror_residuals)
    char last_kernel[96] = {0};    // what the last solve launched (eds_trk_last_launch)
    int last_workgroups = 0, last_team = 1, last_first = 0, last_count = 0, last_layout = 1, last_kind = 0;
    bool last_ticks = false;
    int B = 0;
    int pending_first = 0, pending_count = 0, pending_kind = 0;   // range launched but not yet collected (kind 6 | 12)
    double launch_wall_us = 0.0;
};

int  eds_fused_alloc(EdsFusedBuffers* fb, int B);
// time-out policy (eds_fused.hip): may this solve form teams?  (counts the cool-down down); a team launch timed out / ended clean
unsigned eds_next_done_tag(EdsFusedBuffers* fb);     // the completion-word tag of the next small launch (eds_fused.hip)
bool eds_team_allowed(EdsFusedBuffers* fb);
void eds_team_timed_out(eds_trk* h);
void eds_team_clean(EdsFusedBuffers* fb);
void eds_fused_free(EdsFusedBuffers* fb);
int  eds_fused_solve(eds_trk* h, int level, int first, int count);   // asynchronous on h->st
int  eds_fused_collect(eds_trk* h);                                  // after the stream is idle
int  eds_fused_fetch_trace(eds_trk* h, int slot);                    // D2H of one slot's trace
struct eds_trk_launch_info;
int  eds_fused_last_launch(eds_trk* h, eds_trk_launch_info* out);

// eds_strips.hip: makes the strip copies of the frames the slots [first, first + count) sample current (allocates them at the first call;
// one conversion launch per run of stale slots, on h->st).  false: no memory for them — the caller uses the tiles.
bool eds_strips_prepare(eds_trk* h, int first, int count);
bool eds_strips_for_solve(eds_trk* h, int first, int count);      // the solve kernels' question: gather from strips this time? (converts what the policy says)
bool eds_strips_current(const eds_trk* h, int first, int count);  // every sampled slot of the range has an up-to-date strip copy (nothing is converted)
void eds_strips_free(eds_trk* h);

bool eds_fused12_supported(const eds_trk* h, int first, int count);
int  eds_fused12_solve(eds_trk* h, int level, int first, int count);
int  eds_fused12_collect(eds_trk* h);
struct EdsArrays;
// eds_stream6.hip (wide = 1: one 512-thread workgroup per CU; 0: two 256-thread workgroups per CU)
// launch arguments of one eds_fused6_kernel launch, for the instantiations that live in the second translation unit
// (eds_fused.hip, EDS_FUSED_BILINEAR_TU)
struct EdsFused6Launch {
    const EdsArrays* A; const EdsFusedIn* in; EdsFusedOut* out; void* sv;
    int first, count, threads, iters, damped; double lambda0, tau; int nb;
    unsigned long long* mail; int* ticket; unsigned ticket_base, epoch; int drop; hipStream_t st;
};
void eds_fused6_launch_bilinear(const EdsFused6Launch& L, int ppt, int team, int groups = 1);
void eds_stream6_launch(const EdsArrays& A, int sampling, int wide, const EdsFusedIn* d_in, EdsFusedOut* d_out, void* d_sv, int first,
                        int count, int iters, int damped, double lambda0, double huber_tau, int nb, hipStream_t st);


// ---- event-frame construction on device (eds_frame.hip) ---------------------------------------------------
struct EdsFrameBuffers {
    float *d_mapx = nullptr, *d_mapy = nullptr;     // forward undistortion LUT (sensor size map_H x map_W), optional
    int map_H = 0, map_W = 0;
    double *d_img = nullptr, *d_tmp = nullptr, *d_norm = nullptr;
    size_t img_elems = 0;                           // capacity of d_img / d_tmp (max of sensor and frame size)
    double* d_planes = nullptr;                     // the levels of one event frame, [plane_levels][H][W]
    int plane_levels = 0;
    uint8_t* h_events = nullptr;                   // pinned, device-mapped staging [x | y | polarity]: k_vote reads the events in place
    uint16_t *d_ex = nullptr, *d_ey = nullptr;     // ... as the device sees it
    uint8_t* d_pol = nullptr;
    int cap_events = 0;
    // d_norm holds TWO sets of sum-of-squares accumulators: a call accumulates into set (calls & 1) and its last launch clears the
    // other one for the next call; the totals come back through mapped pinned memory.  d_img is cleared by k_levels once the blur
    // has moved the image on (img_clean says whether that happened).  A frame is then 4 launches and one wait: no memset, no copy.
    uint8_t *h_aos = nullptr, *d_aos = nullptr;    // mapped pinned staging of array-of-struct events (copied as they are, read strided by k_vote_aos)
    size_t cap_aos = 0;
    double *h_norm_out = nullptr, *d_norm_out = nullptr;
    // batched builder (eds_frame_build_batch): accumulation / blur / level images of up to batch_cap slices, their accumulators, and a
    // mapped block [totals | event offsets]
    double *b_img = nullptr, *b_tmp = nullptr, *b_planes = nullptr, *b_norm = nullptr;
    char *h_bmeta = nullptr, *d_bmeta = nullptr;
    int batch_cap = 0, meta_cap = 0;
    unsigned calls = 0;
    bool img_clean = false;
};
void eds_frame_free(EdsFrameBuffers* fb);
int  eds_frame_build_batch(eds_trk* h, int first_slot, int count, const int* offsets, const uint16_t* ex, const uint16_t* ey, const uint8_t* pol,
                           int level, double blur_sigma, int use_exp_weights, double* norms_out);
// few alignments per launch: one more (tiny) launch writes the kept residuals into pinned host memory as well, so that reading
// them back (Tracker.cpp:223-230) costs no copy call and no second wait; false: not mirrored (fetch as usual)
bool eds_mirror_residuals(eds_trk* h, int first, int count);
void eds_frame_store_rowmajor(eds_trk* h, int slot, const float* d_src, int row_b, int row_e);     // row-major fp32 H x W in HBM -> the slot's tiled frame
void eds_frame_store_whole(eds_trk* h, int slot, const float* d_src, hipStream_t st);          // a whole row-major fp32 frame (device-visible) -> the slot's frame, on stream st
void eds_frame_store_follow(eds_trk* h, int slot, unsigned seq, int rows_per);     // ONE launch that follows the host through h_fstage (h_fprog[0]: rows narrowed so far)
int  eds_frame_set_map(eds_trk* h, const float* mapx, const float* mapy, int mH, int mW);
// events as an array of structs (what the reference holds: std::vector<base::samples::Event>): stride and field offsets in bytes
struct EdsEventAos { const void* data; int stride, off_x, off_y, off_pol; };
int  eds_frame_build_levels(eds_trk* h, int first_slot, int level0, int nlevels, int n_events, const uint16_t* ex, const uint16_t* ey,
                            const uint8_t* pol, int sH, int sW, double blur_sigma, int use_exp_weights, double* norms_out,
                            const EdsEventAos* aos = nullptr);

// ---- loss scale and point maintenance on device (eds_points.hip) --------------------------------------------
struct EdsPointBuffers {
    // getCoord's pose in and its outputs (summary, coordinates, tracks, kept indices) live in ONE device-mapped pinned block: the
    // kernel reads / writes it over PCIe (36 B per point), the call is one launch and one wait — no copy calls (each costs
    // 5-10 us on the live path: 93 -> 30 us per 2 000-point alignment)
    char* h_block = nullptr;            // host address of the block
    int cap = 0;                        // alignments it has room for (the batched entry point grows it)
    double *h_summary = nullptr, *h_pose = nullptr, *h_coord = nullptr, *h_track = nullptr;
    int* h_kept = nullptr;
    double *d_coord = nullptr, *d_track = nullptr, *d_summary = nullptr, *d_pose = nullptr;    // the same, as the device sees them
    int* d_kept = nullptr;
    double *h_tau = nullptr, *d_tau = nullptr;    // loss scales of a batch: mapped pinned memory, host / device view
};
void eds_points_free(EdsPointBuffers* pb);
bool eds_points_supported(const eds_trk* h, int first, int count);
int  eds_points_loss_param(eds_trk* h, int first, int count, int method, double* tau_out);
int  eds_points_update(eds_trk* h, int slot, int delete_out, double* coord_xy, double* tracks_xy, int32_t* kept_index, int* n_kept,
                       double* mean_sq_flow);
int  eds_points_update_batch(eds_trk* h, int first, int count, int delete_out, int stride, double* coord_xy, double* tracks_xy,
                             int32_t* kept_index, int* n_kept, double* mean_sq_flow);

// ---- keyframe point set-up on device (eds_keyframe.hip) ------------------------------------------------------
struct eds_kf_select;
struct EdsKeyframeBuffers {
    void* d_raw = nullptr;                                                   // the H x W grey image the pipeline starts from (u8 / f32 / f64)
    void* d_src = nullptr; size_t src_bytes = 0;                             // the image as handed over when it is resized / colour
    double *d_log = nullptr, *d_gx = nullptr, *d_gy = nullptr, *d_mag = nullptr, *d_partial = nullptr;
    int *d_cand = nullptr, *d_cnt = nullptr, *d_off = nullptr, *d_summary = nullptr;
    double *d_coord = nullptr, *d_grad = nullptr, *d_idp = nullptr, *d_w = nullptr;   // candidates, then the cleaned points (in place)
    double *d_dxy = nullptr, *d_didp = nullptr;                              // depth map
    int cap_depth = 0, last_slot = -1, last_N = 0, last_candidates = 0;
    double K[4] = {0, 0, 0, 0};
};
void eds_keyframe_free(EdsKeyframeBuffers* kb);
int  eds_keyframe_build(eds_trk* h, int slot, int img_type, const void* img, int img_H, int img_W, int channels, const eds_kf_select* sel, int n_depth,
                        const double* depth_xy, const double* depth_idp, double fx, double fy, double cx, double cy, int* n_points);
int  eds_keyframe_get_points(eds_trk* h, int slot, double* coord_xy, double* norm_xy, double* grad_xy, double* idp, double* weights);

// defined in eds_capi.hip
int eds_internal_refresh_gram(eds_trk* h, int slot);
int eds_internal_fail(int code, const char* msg);
int eds_stream_idle(eds_trk* h);        // eds_capi.hip: waits for the handle's stream if the last solve was only seen complete through its completion words
int eds_internal_solve_host(eds_trk* h, int level, int first, int count);
