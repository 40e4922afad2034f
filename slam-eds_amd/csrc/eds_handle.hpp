// Private definition of the opaque handle (include/eds_hip.h: `typedef struct eds_trk eds_trk`).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/eds_hip.h"
#include "eds_fused.hpp"
#include "eds_kernels.hpp"
#include "eds_launch_rule.hpp"

static_assert(EDS_RULE_TEAM6_MAX == EDS_TEAM6_MAX && EDS_RULE_TEAM_SLOTS == EDS_TEAM_SLOTS && EDS_RULE_TEAM12_SLOTS == EDS_TEAM12_SLOTS &&
              EDS_RULE_TEAM12_MEMBERS == EDS_TEAM12_MEMBERS && EDS_RULE_TEAM_MEMBERS == EDS_TEAM_MEMBERS, "eds_launch_rule.hpp repeats these constants of eds_fused.hpp");

// Host-side state of one alignment slot: what the reference keeps in Tracker members
// px,qx,vx,info (Tracker.hpp:46-52) and in kf->residuals (KeyFrame.hpp:88).
struct Slot {
    int N = 0;
    bool has_kf = false, has_frame = false;
    int frame_slot = -1;            // >= 0: this alignment samples THAT slot's frame storage (eds_trk_share_event_frame); -1: its own
    unsigned frame_version = 0;     // bumped by everything that writes this slot's frame storage ...
    unsigned strips_version = 0;    // ... and the version its strip copy (eds_layout.hpp) was made from; 0 = never
    unsigned solved_version = 0;    // ... and the version the last on-device solve sampled: a frame that is solved AGAIN gets its strip copy (eds_strips_for_solve)
    double p[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, v[6];
    double K[4] = {0, 0, 0, 0};
    eds_trk_info info;
    std::vector<double> residuals;
    int ntrace = 0;
    std::vector<double> tr_xi, tr_cost;
    std::vector<int32_t> tr_acc;
    bool gram_host_stale = false;   // the Gram matrices in HBM are newer than the host copy h_G (set_idepth does not fetch them: only
                                    // the host-driven solvers and eval read h_G, and they fetch it then — fill_pose)
    bool res_on_device = false;     // residuals of the last solve still only in HBM
    bool res_in_hostmap = false;    // ... and mirrored in the handle's pinned h_rmap by the kernel that produced them (small launches)
    bool trace_on_device = false;   // trace of the last solve still only in HBM
};


struct eds_trk {
    eds_trk_cfg cfg;
    EdsKnobs knobs;                     // tuning knobs (eds_launch_rule.hpp): the environment as it was at eds_trk_create, then eds_trk_set_knob
    bool strips_unavailable = false;    // the strip copies did not fit the memory budget: remembered, not retried on every solve
    size_t strips_bytes = 0;            // what they take (eds_trk_strips_bytes)
    int B = 0, Nmax = 0, Np = 0, H = 0, W = 0, max_seg = 0, dev = 0;
    int Hp = 0, Wp = 0, tiled = 1;      // frame allocation (eds_device.hpp FrameView)
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // device
    double *dpose = nullptr, *dG = nullptr, *dpart = nullptr, *dncstat = nullptr;
    float* dkf = nullptr;               // ONE allocation holding the nine per-point planes below, [9][B][Np] (eds_layout.hpp)
    float *dx = nullptr, *dy = nullptr, *drho = nullptr, *dgx = nullptr, *dgy = nullptr, *dw = nullptr;
    float *df0x = nullptr, *df0y = nullptr;
    int* dcell0 = nullptr;
    float *dmhat = nullptr, *dframe = nullptr, *dr = nullptr, *dJ = nullptr;
    float* dstrips = nullptr;           // strip layout of the frames, allocated by the first solve that uses it (eds_strips.hip)
    int strip_phases = 1;               // its row phases (eds_layout.hpp), fixed when it is allocated
    EdsFusedBuffers fused;
    EdsFrameBuffers frame_build;
    EdsPointBuffers point_ops;
    EdsKeyframeBuffers kf_build;
    // pinned host staging
    double *h_pose = nullptr, *h_part = nullptr, *h_G = nullptr;
    float *h_f32 = nullptr, *h_r = nullptr;
    float *h_fstage = nullptr, *d_fstage = nullptr;   // pinned, device-mapped H x W fp32: set_event_frame narrows into it; nobody else writes it
    hipEvent_t ev_stage = nullptr;      // recorded behind the last copy out of h_fstage
    unsigned *h_fprog = nullptr, *d_fprog = nullptr;  // pinned, device-mapped: [0] (upload number << 20 | rows of h_fstage narrowed so far), [1] time-out
                                                      // mark of the follower kernel (eds_frame_store_follow)
    unsigned upload_seq = 0;
    float *h_rmap = nullptr, *d_rmap = nullptr;       // pinned, device-mapped [min(B, EDS_RHOST_SLOTS)][Np]: residuals of small launches (eds_mirror_residuals)
    float* h_idp = nullptr;             // pinned [Np]: set_idepth narrows into it (private, like h_fstage)
    float* d_idp = nullptr;             // ... as the device sees it (the gram kernel reads the new depths in place)
    std::vector<double> scratch;        // host scratch of eds_trk_loss_param (no allocation per call on the live path)
    hipEvent_t ev_idp = nullptr;
    bool idp_busy = false;
    bool gram_pending = false;          // h_G's refresh is still in flight on the stream (set_idepth does not wait for it)
    bool stage_busy = false;            // ev_stage has to be waited for before h_fstage is written again
    bool stream_dirty = false;          // the last small solve was seen complete through its workgroups' done words in pinned memory
                                        // (eds_capi.hip: wait_stream) — the stream itself has not been waited for.  Records and residual mirror
                                        // are covered by the words; anything else a host reader wants of that launch: eds_stream_idle() first
    size_t h_f32_elems = 0;
    float *h_bstage = nullptr, *d_bstage = nullptr;   // pinned, device-mapped ring of staging slots of eds_trk_set_event_frames (allocated at its first call)
    hipStream_t st_up = nullptr;        // second stream of the batch upload (alternate frames), created at its first call
    hipEvent_t ev_up = nullptr;
    float* d_bdev = nullptr;            // two row-major H x W scratch frames in HBM (the copy-engine variant of the batch upload: EDS_UPLOAD_DMA)
    int bstage_slots = 0;
    bool bstage_busy = false;           // the last batch's store kernels may still read the ring
    std::vector<hipEvent_t> ev_bstage;  // one per staging slot: recorded behind the kernel that read it
    void* d_probe = nullptr;            // scratch of the measurement helpers (eds_trk_hbm_probe, eds_trk_bench_kernel_cold): allocated at their first call
    size_t probe_bytes = 0;
    std::vector<Slot> slots;

    EdsArrays arrays() const {
        EdsArrays A;
        A.x = dx; A.y = dy; A.rho = drho; A.gx = dgx; A.gy = dgy; A.w = dw;
        A.f0x = df0x; A.f0y = df0y; A.cell0 = dcell0;
        A.kf = dkf; A.kf_plane = (size_t)B * Np;
        A.mhat = dmhat; A.frame = dframe; A.rmap = nullptr; A.done = nullptr; A.done_tag = 0; A.strips = nullptr /* set by the launch sites that made sure the copies are current */; A.strip_phases = strip_phases; A.pose = dpose; A.G = dG; A.r = dr; A.J = dJ; A.part = dpart; A.ncstat = dncstat;
        A.B = B; A.Np = Np; A.H = H; A.W = W; A.max_seg = max_seg;
        A.Hp = Hp; A.Wp = Wp; A.tiled = tiled;
        return A;
    }
};

