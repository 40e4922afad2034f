// C ABI of libeds_hip.so (include/eds_hip.h), first of three translation units (eds_capi_internal.hpp): handle management,
// configuration and knobs, states, sync / info, and the rows around the path.  Inputs: eds_capi_inputs.hip; passes and solves:
// eds_capi_solve.hip; the persistent on-device loop: eds_fused.hip.
//
// Replaces, for the hot path only, reference src/tracking/Tracker.cpp:40-102 (state handling),
// :104-241 (optimize) and :281-317 (getLossParams).  There is deliberately NO CPU fallback:
// without a HIP device every entry point fails with EDS_ERR_NO_DEVICE.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "eds_capi_internal.hpp"

namespace {
thread_local std::string g_last_error;
}

namespace edscapi {

int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

int effective_blocks(const eds_trk* h) {
    int nb = h->cfg.num_blocks;
    if (nb < 1) nb = 1;
    if (nb > EDS_MAX_BLOCKS) nb = EDS_MAX_BLOCKS;
    return nb;
}
int level_iters(const eds_trk* h, int level) {
    if (level < 0) level = 0;
    if (level >= EDS_MAX_LEVELS) level = EDS_MAX_LEVELS - 1;
    return h->cfg.max_num_iterations[level];
}

int check_slot(const eds_trk* h, int slot) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (slot < 0 || slot >= h->B) return fail(EDS_ERR_INVALID, "slot out of range");
    return EDS_OK;
}

// constants of a slot's pose block that do not depend on (p,q,v)
void fill_static(const eds_trk* h, int slot) {
    const Slot& s = h->slots[slot];
    double* pb = h->h_pose + (size_t)slot * EDS_POSE_STRIDE;
    const int nb = effective_blocks(h);
    for (int i = 0; i < 4; ++i) pb[EDS_PB_K + i] = s.K[i];
    pb[EDS_PB_HUBER] = h->cfg.huber_tau > 0 ? h->cfg.huber_tau : 0.0;
    pb[EDS_PB_NB] = nb;
    pb[EDS_PB_NE] = s.N / nb;
    pb[EDS_PB_N] = s.N;
    pb[EDS_PB_NCMODE] = h->cfg.nc ? 1.0 : 0.0;
    pb[EDS_PB_FRAME] = (double)(s.frame_slot >= 0 ? s.frame_slot : slot);
}

void fill_pose(eds_trk* h, int slot, const double* p, const double* q, const double* v) {
    if (h->gram_pending) { hipStreamSynchronize(h->st); h->gram_pending = false; }      // h_G as an earlier refresh left it
    if (h->slots[slot].gram_host_stale) {           // set_idepth refreshed the Gram matrices in HBM only: fetch this slot's now
        const size_t off = (size_t)slot * EDS_MAX_BLOCKS * 36;
        hipMemcpyAsync(h->h_G + off, h->dG + off, (size_t)EDS_MAX_BLOCKS * 36 * 8, hipMemcpyDeviceToHost, h->st);
        hipStreamSynchronize(h->st);
        h->slots[slot].gram_host_stale = false;
    }
    fill_static(h, slot);
    edsm::fill_pose_block(p, q, v, h->h_G + (size_t)slot * EDS_MAX_BLOCKS * 36, effective_blocks(h),
                          h->h_pose + (size_t)slot * EDS_POSE_STRIDE);
}

int upload_pose(eds_trk* h, int first, int count) {
    EDS_HIP_TRY(hipMemcpyAsync(h->dpose + (size_t)first * EDS_POSE_STRIDE, h->h_pose + (size_t)first * EDS_POSE_STRIDE,
                               sizeof(double) * EDS_POSE_STRIDE * count, hipMemcpyHostToDevice, h->st));
    return EDS_OK;
}

int max_points(const eds_trk* h, int first, int count) {
    int m = 0;
    for (int s = first; s < first + count; ++s) m = std::max(m, h->slots[s].N);
    return m;
}

int check_range(const eds_trk* h, int first, int count) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (first < 0 || count < 1 || first + count > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds");
    return EDS_OK;
}

}  // namespace edscapi
using namespace edscapi;

namespace {

void free_all(eds_trk* h) {
    if (!h) return;
    hipSetDevice(h->dev);
    void* dptrs[] = {h->dkf, h->dpose, h->dG, h->dpart, h->dncstat,
                     h->dmhat, h->dframe, h->dr, h->dJ, h->d_probe, h->d_bdev};
    for (void* p : dptrs) if (p) hipFree(p);
    eds_fused_free(&h->fused);
    eds_strips_free(h);
    eds_frame_free(&h->frame_build);
    eds_points_free(&h->point_ops);
    eds_keyframe_free(&h->kf_build);
    void* hptrs[] = {h->h_pose, h->h_part, h->h_G, h->h_f32, h->h_r, h->h_fstage, h->h_rmap, h->h_idp, h->h_fprog, h->h_bstage};
    for (hipEvent_t e : h->ev_bstage) hipEventDestroy(e);
    for (void* p : hptrs) if (p) hipHostFree(p);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->ev_stage) hipEventDestroy(h->ev_stage);
    if (h->ev_idp) hipEventDestroy(h->ev_idp);
    if (h->ev_up) hipEventDestroy(h->ev_up);
    if (h->st_up) hipStreamDestroy(h->st_up);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

}  // namespace

int eds_internal_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }
int eds_internal_solve_host(eds_trk* h, int level, int first, int count) { return solve_host(h, level, first, count); }

// (C++ linkage: called from the other translation units of the library — eds_fused.hpp)
int eds_stream_idle(eds_trk* h) {
    if (!h->stream_dirty) return EDS_OK;
    h->stream_dirty = false;
    EDS_HIP_TRY(hipStreamSynchronize(h->st));
    return EDS_OK;
}

extern "C" {

int eds_abi_version(void) { return EDS_HIP_ABI_VERSION; }

int eds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* eds_last_error(void) { return g_last_error.c_str(); }

int eds_trk_cfg_size(void) { return (int)sizeof(eds_trk_cfg); }
int eds_trk_info_size(void) { return (int)sizeof(eds_trk_info); }

void eds_trk_cfg_default(eds_trk_cfg* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0;
    cfg->sampling = EDS_SAMPLE_BICUBIC;
    cfg->solver = EDS_SOLVER_LM6;
    cfg->exec = EDS_EXEC_DEVICE;
    cfg->num_blocks = 1;
    cfg->loss_type = EDS_LOSS_NONE;
    cfg->loss_param = 1.0;
    cfg->huber_tau = 0.0;
    cfg->lambda0 = 0.01;
    cfg->num_levels = 1;
    for (int i = 0; i < EDS_MAX_LEVELS; ++i) cfg->max_num_iterations[i] = 10;
    cfg->function_tolerance = 1e-6;
    cfg->gradient_tolerance = 1e-8;
    cfg->parameter_tolerance = 1e-6;
}

int eds_trk_create(const eds_trk_cfg* cfg, int batch, int max_points_, int H, int W, eds_trk** out) {
    if (!cfg || !out) return fail(EDS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (batch < 1 || max_points_ < 1 || H < 4 || W < 4) return fail(EDS_ERR_INVALID, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(EDS_ERR_NO_DEVICE, "no HIP device visible; libeds_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(EDS_ERR_INVALID, "device ordinal out of range");
    int device_cus = EDS_RULE_CUS;
    {   // the code object holds gfx950 kernels only (include/eds_hip.h: EDS_ERR_NO_DEVICE = "no gfx950 device visible")
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(EDS_ERR_HIP, "hipGetDeviceProperties failed");
        if (!std::strstr(prop.gcnArchName, "gfx950"))
            return fail(EDS_ERR_NO_DEVICE, std::string("device ") + std::to_string(cfg->device) + " is " + prop.gcnArchName +
                                               ", not gfx950 (MI355X); libeds_hip has no other code path and no CPU fallback");
        if (prop.multiProcessorCount > 0) device_cus = prop.multiProcessorCount;      // (a partitioned or smaller gfx950 part: the candidate-group rule counts ITS CUs)
    }
    eds_trk* h = new (std::nothrow) eds_trk();
    if (!h) return fail(EDS_ERR_INVALID, "out of memory");
    h->cfg = *cfg;
    h->B = batch; h->Nmax = max_points_; h->H = H; h->W = W; h->dev = cfg->device;
    h->Hp = eds_frame_extent(H); h->Wp = eds_frame_extent(W);
    h->tiled = 1;
    if (const char* bad = eds_knobs_from_env(&h->knobs)) {      // the ONLY place the library reads tuning variables from the environment (eds_launch_rule.hpp)
        const char* v = getenv(bad);
        std::string msg = std::string("environment variable ") + bad + "=" + (v ? v : "") + " is not a value that knob takes (include/eds_hip.h: eds_trk_set_knob)";
        delete h;
        return fail(EDS_ERR_INVALID, msg);
    }
    h->knobs.cus = device_cus;
    h->tiled = h->knobs.frame_rowmajor ? 0 : 1;
    h->Np = ((max_points_ + EDS_POINT_ALIGN - 1) / EDS_POINT_ALIGN) * EDS_POINT_ALIGN;
    h->max_seg = h->Np / EDS_TPB + 2 * EDS_MAX_BLOCKS + 2;
    h->slots.resize(batch);
    for (Slot& s : h->slots) {                       // Tracker ctor, Tracker.cpp:43-46
        const double c = 0.001 / std::sqrt(6.0 * 0.001 * 0.001);
        for (int i = 0; i < 6; ++i) s.v[i] = c;
        std::memset(&s.info, 0, sizeof(s.info));
    }
    const size_t BN = (size_t)batch * h->Np;
#define EDS_ALLOC(ptr, bytes)                                                                  \
    do {                                                                                       \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                    \
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } \
    } while (0)
#define EDS_HALLOC(ptr, bytes)                                                                 \
    do {                                                                                       \
        hipError_t e_ = hipHostMalloc((void**)&(ptr), (bytes), hipHostMallocDefault);          \
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e_)); } \
    } while (0)
    hipError_t e = hipSetDevice(h->dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_idp, hipEventDisableTiming);
    if (e != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("stream/event: ") + hipGetErrorString(e)); }
    EDS_ALLOC(h->dkf, BN * 4 * EDS_KF_PLANES);
    h->dx = h->dkf + EDS_KF_X * BN; h->dy = h->dkf + EDS_KF_Y * BN; h->drho = h->dkf + EDS_KF_RHO * BN;
    h->dgx = h->dkf + EDS_KF_GX * BN; h->dgy = h->dkf + EDS_KF_GY * BN; h->dw = h->dkf + EDS_KF_W * BN;
    h->df0x = h->dkf + EDS_KF_F0X * BN; h->df0y = h->dkf + EDS_KF_F0Y * BN;
    h->dcell0 = reinterpret_cast<int*>(h->dkf + EDS_KF_CELL0 * BN);
    EDS_ALLOC(h->dmhat, BN * 4); EDS_ALLOC(h->dr, BN * 4); EDS_ALLOC(h->dJ, BN * 4 * 12);
    EDS_ALLOC(h->dframe, (size_t)batch * h->Hp * h->Wp * 4);
    EDS_ALLOC(h->dpose, (size_t)batch * EDS_POSE_STRIDE * 8);
    EDS_ALLOC(h->dG, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8);
    EDS_ALLOC(h->dpart, (size_t)batch * h->max_seg * EDS_RED_K * 8);
    EDS_ALLOC(h->dncstat, (size_t)batch * EDS_MAX_BLOCKS * 8 * 8);
    EDS_HALLOC(h->h_pose, (size_t)batch * EDS_POSE_STRIDE * 8);
    EDS_HALLOC(h->h_part, (size_t)batch * h->max_seg * EDS_RED_K * 8);
    EDS_HALLOC(h->h_G, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8);
    h->h_f32_elems = std::max((size_t)h->Hp * h->Wp, (size_t)h->Np * 12);
    EDS_HALLOC(h->h_f32, h->h_f32_elems * 4);
    EDS_HALLOC(h->h_r, BN * 4);
    EDS_HALLOC(h->h_idp, (size_t)h->Np * 4);
    if (hipHostGetDevicePointer((void**)&h->d_idp, h->h_idp, 0) != hipSuccess) { (void)hipGetLastError(); h->d_idp = nullptr; }
    {   // mirror of the residual plane for the first few slots (eds_mirror_residuals)
        const size_t nr = (size_t)std::min(batch, EDS_RHOST_SLOTS) * h->Np;
        hipError_t e_ = hipHostMalloc((void**)&h->h_rmap, nr * 4, hipHostMallocMapped);
        if (e_ == hipSuccess) e_ = hipHostGetDevicePointer((void**)&h->d_rmap, h->h_rmap, 0);
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipHostMalloc (residual mirror): ") + hipGetErrorString(e_)); }
        std::memset(h->h_rmap, 0, nr * 4);
    }
    {   // device-mapped: the tiling kernel reads the staging buffer in place
        hipError_t e_ = hipHostMalloc((void**)&h->h_fstage, (size_t)H * W * 4, hipHostMallocMapped);
        if (e_ == hipSuccess) e_ = hipHostGetDevicePointer((void**)&h->d_fstage, h->h_fstage, 0);
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipHostMalloc (frame staging): ") + hipGetErrorString(e_)); }
    }
    {   // progress / time-out words of the follower upload (eds_frame_store_follow); without them set_event_frame launches per band
        hipError_t e_ = hipHostMalloc((void**)&h->h_fprog, 64, hipHostMallocMapped);
        if (e_ == hipSuccess) e_ = hipHostGetDevicePointer((void**)&h->d_fprog, h->h_fprog, 0);
        if (e_ != hipSuccess) { (void)hipGetLastError(); if (h->h_fprog) hipHostFree(h->h_fprog); h->h_fprog = nullptr; h->d_fprog = nullptr; }
        else std::memset(h->h_fprog, 0, 64);
    }
    std::memset(h->h_pose, 0, (size_t)batch * EDS_POSE_STRIDE * 8);
    std::memset(h->h_G, 0, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8);
    hipMemsetAsync(h->dcell0, 0, BN * 4, h->st);
    float* f32s[] = {h->dx, h->dy, h->drho, h->dgx, h->dgy, h->dw, h->dmhat, h->dr, h->df0x, h->df0y};
    for (float* p : f32s) hipMemsetAsync(p, 0, BN * 4, h->st);
    hipMemsetAsync(h->dJ, 0, BN * 4 * 12, h->st);
    hipMemsetAsync(h->dframe, 0, (size_t)batch * h->Hp * h->Wp * 4, h->st);
    hipMemsetAsync(h->dpose, 0, (size_t)batch * EDS_POSE_STRIDE * 8, h->st);
    hipMemsetAsync(h->dG, 0, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8, h->st);
    hipMemsetAsync(h->dpart, 0, (size_t)batch * h->max_seg * EDS_RED_K * 8, h->st);
    hipMemsetAsync(h->dncstat, 0, (size_t)batch * EDS_MAX_BLOCKS * 8 * 8, h->st);
    int rc = eds_fused_alloc(&h->fused, batch);
    if (rc != 0) { free_all(h); return fail(EDS_ERR_HIP, "fused buffers: hipMalloc failed"); }
    e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("init: ") + hipGetErrorString(e)); }
    *out = h;
    return EDS_OK;
#undef EDS_ALLOC
#undef EDS_HALLOC
}

void eds_trk_destroy(eds_trk* h) { free_all(h); }

int eds_trk_set_config(eds_trk* h, const eds_trk_cfg* cfg) {
    if (!h || !cfg) return fail(EDS_ERR_INVALID, "null argument");
    if (cfg->device != h->cfg.device) return fail(EDS_ERR_INVALID, "device cannot change after create");
    const bool blocks_changed = cfg->num_blocks != h->cfg.num_blocks;
    h->cfg = *cfg;
    if (blocks_changed) {               // Gram matrices are per residual block
        for (int s = 0; s < h->B; ++s) {
            if (!h->slots[s].has_kf) continue;
            fill_static(h, s);
            int rc = upload_pose(h, s, 1);
            if (rc) return rc;
            eds_launch_gram(h->arrays(), s, effective_blocks(h), h->st);
        }
        EDS_HIP_TRY(hipMemcpyAsync(h->h_G, h->dG, (size_t)h->B * EDS_MAX_BLOCKS * 36 * 8, hipMemcpyDeviceToHost, h->st));
        EDS_HIP_TRY(hipStreamSynchronize(h->st));
    }
    return EDS_OK;
}

int eds_trk_get_config(const eds_trk* h, eds_trk_cfg* cfg) {
    if (!h || !cfg) return fail(EDS_ERR_INVALID, "null argument");
    *cfg = h->cfg;
    return EDS_OK;
}

int eds_trk_set_state(eds_trk* h, int slot, const double p[3], const double q[4], const double v[6]) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    Slot& s = h->slots[slot];
    if (p) std::memcpy(s.p, p, sizeof(s.p));
    if (q) std::memcpy(s.q, q, sizeof(s.q));
    if (v) std::memcpy(s.v, v, sizeof(s.v));
    return EDS_OK;
}

int eds_trk_get_state(eds_trk* h, int slot, double p[3], double q[4], double v[6]) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    const Slot& s = h->slots[slot];
    if (p) std::memcpy(p, s.p, sizeof(s.p));
    if (q) std::memcpy(q, s.q, sizeof(s.q));
    if (v) std::memcpy(v, s.v, sizeof(s.v));
    return EDS_OK;
}

int eds_trk_set_states(eds_trk* h, int first, int count, const double* p, const double* q, const double* v) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    for (int i = 0; i < count; ++i) {
        Slot& s = h->slots[first + i];
        if (p) std::memcpy(s.p, p + 3 * i, sizeof(s.p));
        if (q) std::memcpy(s.q, q + 4 * i, sizeof(s.q));
        if (v) std::memcpy(s.v, v + 6 * i, sizeof(s.v));
    }
    return EDS_OK;
}

int eds_trk_get_states(eds_trk* h, int first, int count, double* p, double* q, double* v) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    for (int i = 0; i < count; ++i) {
        const Slot& s = h->slots[first + i];
        if (p) std::memcpy(p + 3 * i, s.p, sizeof(s.p));
        if (q) std::memcpy(q + 4 * i, s.q, sizeof(s.q));
        if (v) std::memcpy(v + 6 * i, s.v, sizeof(s.v));
    }
    return EDS_OK;
}

int eds_trk_get_results(eds_trk* h, int first, int count, double* t) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    if (!t) return fail(EDS_ERR_INVALID, "null output");
    for (int i = 0; i < count; ++i) {
        const Slot& s = h->slots[first + i];
        double* o = t + 16 * i;
        std::memcpy(o, s.p, 24); std::memcpy(o + 3, s.q, 32); std::memcpy(o + 7, s.v, 48);
        o[13] = s.info.final_cost; o[14] = s.info.num_iterations; o[15] = s.info.success ? 1.0 : 0.0;
    }
    return EDS_OK;
}

// Waits for the handle's stream.  A launch of a few alignments is over in 0.1-0.3 ms, and a blocking wait adds the wake-up of the
// calling thread to every such call; the latency regime therefore polls the stream (hipStreamQuery) for up to EDS_SPIN_US before
// it blocks.  Batches block right away: nobody should burn a core for milliseconds.
// Round 6: every workgroup of a small solve ends by storing the launch's tag into its own word of pinned host memory, behind a
// system-scope fence over everything it wrote (result record, residual mirror) — so the host can see the solve finish WITHOUT the runtime:
// no end-of-kernel cache release, no completion signal, no hipStreamQuery per poll (B = 1: -6 .. -8 us per call).  The stream is then not
// known to be idle (the pose-only kernel copies its trace to HBM behind its word): h->stream_dirty says so, and eds_stream_idle() is
// what the one reader of that copy (eds_fused_fetch_trace: a null-stream copy, which does not wait for this non-blocking stream) calls first.  A word that never arrives (a workgroup that never ran) ends the poll after EDS_SPIN_US.
#define EDS_SPIN_US 500.0
static hipError_t wait_stream(eds_trk* h) {
    const EdsFusedBuffers& fb = h->fused;
    const bool spin = h->cfg.exec == EDS_EXEC_DEVICE && fb.pending_count > 0 && fb.pending_count <= 64 && !h->knobs.no_spin;
    if (spin && fb.pending_vteam > 0 && fb.h_done) {
        const auto t0 = std::chrono::steady_clock::now();
        const unsigned tag = fb.done_seq;
        const unsigned* w = fb.h_done + (size_t)fb.pending_first * EDS_DONE_WORDS;
        int s = 0, m = 0;                                   // next word to see: alignment s of the range, workgroup m
        for (unsigned it = 0;; ++it) {
            while (s < fb.pending_count && __atomic_load_n(w + (size_t)s * EDS_DONE_WORDS + m, __ATOMIC_ACQUIRE) == tag)
                if (++m >= fb.pending_vteam) { m = 0; ++s; }
            if (s >= fb.pending_count) { h->stream_dirty = true; return hipSuccess; }
            if ((it & 63u) == 63u && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > EDS_SPIN_US) break;
            __builtin_ia32_pause();
        }
        h->stream_dirty = false;
        return hipStreamSynchronize(h->st);
    }
    h->stream_dirty = false;
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(h->st);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > EDS_SPIN_US) break;
        }
    }
    return hipStreamSynchronize(h->st);
}

int eds_trk_sync(eds_trk* h) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    EDS_HIP_TRY(wait_stream(h));
    if (h->h_fprog && (h->h_fprog[1] & 0x80000000u)) {      // a frame upload gave up waiting for the host: that slot's frame is incomplete
        h->h_fprog[1] = 0;
        if (h->cfg.exec == EDS_EXEC_DEVICE) (void)eds_fused_collect(h);
        return fail(EDS_ERR_HIP, "a frame upload timed out waiting for the host (set the event frame again)");
    }
    if (h->cfg.exec == EDS_EXEC_DEVICE) return eds_fused_collect(h);
    return EDS_OK;
}

int eds_trk_get_info(eds_trk* h, int slot, eds_trk_info* info) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!info) return fail(EDS_ERR_INVALID, "null info");
    *info = h->slots[slot].info;
    return EDS_OK;
}

int eds_trk_get_trace(eds_trk* h, int slot, int max_iters, double* increments, double* costs, int32_t* accepted) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (h->slots[slot].trace_on_device && (rc = eds_fused_fetch_trace(h, slot))) return rc;
    const Slot& s = h->slots[slot];
    const int n = std::min(max_iters, s.ntrace);
    for (int i = 0; i < n; ++i) {
        if (increments) std::memcpy(increments + 6 * i, &s.tr_xi[6 * i], 6 * sizeof(double));
        if (costs) costs[i] = 0.5 * s.tr_cost[i];
        if (accepted) accepted[i] = s.tr_acc[i];
    }
    return n;
}

int eds_trk_timer_start(eds_trk* h) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    EDS_HIP_TRY(hipEventRecord(h->ev0, h->st));
    return EDS_OK;
}

int eds_trk_prepare_frames(eds_trk* h, int first, int count, int force, float* elapsed_ms) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (!h->tiled) return EDS_OK;      // (both samplers gather from the strips: the bilinear one reads the middle of the same patches)
    EDS_HIP_TRY(hipSetDevice(h->dev));
    if (force)
        for (int s = first; s < first + count; ++s) h->slots[h->slots[s].frame_slot >= 0 ? h->slots[s].frame_slot : s].strips_version = 0;
    if (elapsed_ms) EDS_HIP_TRY(hipEventRecord(h->ev0, h->st));
    if (!eds_strips_prepare(h, first, count)) return fail(EDS_ERR_HIP, "no memory for the strip copies of the frames");
    if (elapsed_ms) {
        EDS_HIP_TRY(hipEventRecord(h->ev1, h->st));
        EDS_HIP_TRY(hipEventSynchronize(h->ev1));
        EDS_HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    }
    return EDS_OK;
}

int eds_trk_set_knob(eds_trk* h, const char* name, const char* value) {
    if (!h || !name) return fail(EDS_ERR_INVALID, "null argument");
    if (std::strcmp(name, "EDS_FRAME_LAYOUT") == 0) return fail(EDS_ERR_STATE, "EDS_FRAME_LAYOUT decides the allocation: environment at eds_trk_create only");
    //  a solve in flight reads the knobs between its launch and its collection: no changes under it
    if (h->fused.pending_count > 0) return fail(EDS_ERR_STATE, "a batch is in flight: call eds_trk_sync before changing a knob");
    const int rc = eds_knobs_set(&h->knobs, name, value);
    if (rc == -1) return fail(EDS_ERR_INVALID, std::string("unknown knob ") + name);
    if (rc != 0) return fail(EDS_ERR_INVALID, std::string("knob ") + name + " does not take the value '" + (value ? value : "") + "'");
    if (std::strncmp(name, "EDS_STRIPS_", 11) == 0) {
        // a new budget / phase count / policy: the copies are allocated again (with the new row phases) by the next solve that wants them
        h->strips_unavailable = false;
        if (h->dstrips && (std::strcmp(name, "EDS_STRIPS_PHASES") == 0 || std::strcmp(name, "EDS_STRIPS_BUDGET_PCT") == 0)) {
            EDS_HIP_TRY(hipSetDevice(h->dev));
            EDS_HIP_TRY(hipStreamSynchronize(h->st));
            eds_strips_free(h);
            for (Slot& sl : h->slots) sl.strips_version = 0;
        }
    }
    return EDS_OK;
}

// The instantiations of the persistent kernels this library was compiled with (csrc/eds_launch_rule.hpp's X-macro lists: the launchers
// dispatch over the same lists).  family 0: eds_fused6_kernel<S, P, T, Q, K, G> (both translation units, candidate groups included),
// 1: eds_fused12_kernel<S, T, CAP, NC, K, Q>.  Returns the number of instantiations of the family (or -1); when 0 <= index < count and
// args6 is not NULL the six template arguments of instantiation `index` are written to it.
int eds_trk_kernel_instances(int family, int index, int32_t* args6) {
    static const int32_t f6[][6] = {
#define EDS_I5_(s, p, t, q, k) {s, p, t, q, k, 1},
#define EDS_I6_(s, p, t, q, k, g) {s, p, t, q, k, g},
        EDS_FUSED6_MAIN_INSTANCES(EDS_I5_) EDS_FUSED6_BILINEAR_INSTANCES(EDS_I5_) EDS_FUSED6_GROUP_INSTANCES(EDS_I6_) EDS_FUSED6_BILINEAR_GROUP_INSTANCES(EDS_I6_)
#undef EDS_I5_
#undef EDS_I6_
    };
    static const int32_t f12[][6] = {
#define EDS_I12_(s, t, c, n, k, q) {s, t, c, n ? 1 : 0, k, q},
        EDS_FUSED12_INSTANCES(EDS_I12_)
#undef EDS_I12_
    };
    static const int32_t f12g[][6] = {              // family 2: the candidate-group instantiations of eds_fused12_kernel as {S, T, NC, K, Q, G} (CAP = 512 for all of them)
#define EDS_I12G_(s, t, c, n, k, q, g) {s, t, n ? 1 : 0, k, q, g},
        EDS_FUSED12_GROUP_INSTANCES(EDS_I12G_)
#undef EDS_I12G_
    };
    const int n = family == 0 ? (int)(sizeof(f6) / sizeof(f6[0])) : (family == 1 ? (int)(sizeof(f12) / sizeof(f12[0])) : (family == 2 ? (int)(sizeof(f12g) / sizeof(f12g[0])) : -1));
    if (n > 0 && args6 && index >= 0 && index < n) std::memcpy(args6, family == 0 ? f6[index] : (family == 1 ? f12[index] : f12g[index]), 6 * sizeof(int32_t));
    return n;
}

int eds_trk_get_strips_info(eds_trk* h, int64_t* bytes, int32_t* row_phases, int32_t* unavailable) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (bytes) *bytes = (int64_t)h->strips_bytes;
    if (row_phases) *row_phases = h->dstrips ? h->strip_phases : 0;
    if (unavailable) *unavailable = h->strips_unavailable ? 1 : 0;
    return EDS_OK;
}

int eds_trk_last_launch(eds_trk* h, eds_trk_launch_info* out) {
    if (!h || !out) return fail(EDS_ERR_INVALID, "null argument");
    return eds_fused_last_launch(h, out);
}

int eds_trk_timer_stop(eds_trk* h, float* elapsed_ms) {
    if (!h || !elapsed_ms) return fail(EDS_ERR_INVALID, "null argument");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    EDS_HIP_TRY(hipEventRecord(h->ev1, h->st));
    EDS_HIP_TRY(hipEventSynchronize(h->ev1));
    EDS_HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return EDS_OK;
}

int eds_trk_update_points(eds_trk* h, int slot, int delete_out_points, double* coord_xy, double* tracks_xy, int32_t* kept_index,
                          int* n_kept, double* mean_sq_flow) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    Slot& s = h->slots[slot];
    if (!s.has_kf) return fail(EDS_ERR_STATE, "keyframe not set");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    int n = 0;
    if ((rc = eds_points_update(h, slot, delete_out_points != 0, coord_xy, tracks_xy, kept_index, &n, mean_sq_flow))) return rc;
    if (n_kept) *n_kept = n;
    if (n != s.N) {                      // points were erased: every index-aligned plane was compacted on the device
        s.N = n;
        s.residuals.clear();
        s.res_on_device = false;
        if (n > 0 && (rc = refresh_gram(h, slot))) return rc;
        if (n == 0) s.has_kf = false;
    }
    return EDS_OK;
}

int eds_trk_update_points_batch(eds_trk* h, int first, int count, int delete_out_points, int stride, double* coord_xy, double* tracks_xy,
                                int32_t* kept_index, int* n_kept, double* mean_sq_flow) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (count < 1 || first < 0 || first + count > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds");
    int maxN = 0;
    for (int s = first; s < first + count; ++s) {
        if (!h->slots[s].has_kf) return fail(EDS_ERR_STATE, "keyframe not set");
        maxN = std::max(maxN, h->slots[s].N);
    }
    if ((coord_xy || tracks_xy || kept_index) && stride < maxN) return fail(EDS_ERR_INVALID, "stride smaller than the largest point count");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    std::vector<int> n(count);
    int rc = eds_points_update_batch(h, first, count, delete_out_points != 0, stride, coord_xy, tracks_xy, kept_index, n.data(), mean_sq_flow);
    if (rc) return rc;
    bool any = false;
    for (int b = 0; b < count; ++b) {
        Slot& s = h->slots[first + b];
        if (n_kept) n_kept[b] = n[b];
        if (n[b] != s.N) {                  // points were erased: every index-aligned plane was compacted on the device
            s.N = n[b];
            s.residuals.clear();
            s.res_on_device = false;
            if (n[b] > 0) { if ((rc = refresh_gram(h, first + b, false))) return rc; any = true; }
            else s.has_kf = false;
        }
    }
    if (any) { EDS_HIP_TRY(hipStreamSynchronize(h->st)); h->gram_pending = false; }
    return EDS_OK;
}

/* ---- keyframe point set-up on the device ------------------------------------------ */
void eds_kf_select_default(eds_kf_select* sel) {
    if (!sel) return;
    std::memset(sel, 0, sizeof(*sel));
    sel->method = EDS_KF_MEDIAN;            // KeyFrame::create falls back to MEDIAN without a point target (KeyFrame.cpp:410-411)
    sel->cell = 20;                          // cv::Size(20, 20)  (KeyFrame.cpp:408)
    sel->num_points = 0;
    sel->sobel_ksize = 3;                    // KeyFrame::create (KeyFrame.cpp:384-385); 7: the constructor's (KeyFrame.cpp:239-240)
    sel->min_depth = 1.0; sel->max_depth = 3.0;
    sel->weight_threshold = 0.7;             // cleanPoints(0.7)  (KeyFrame.cpp:451)
}

int eds_trk_build_keyframe(eds_trk* h, int slot, int img_type, const void* img, const eds_kf_select* sel, int n_depth,
                           const double* depth_xy, const double* depth_idp, double fx, double fy, double cx, double cy, int* n_points) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!img || !sel) return fail(EDS_ERR_INVALID, "null image or selection parameters");
    if (n_depth < 0) return fail(EDS_ERR_INVALID, "negative depth-map size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_keyframe_build(h, slot, img_type, img, h->H, h->W, 1, sel, n_depth, depth_xy, depth_idp, fx, fy, cx, cy, n_points);
}

int eds_trk_build_keyframe_image(eds_trk* h, int slot, int img_type, const void* img, int img_H, int img_W, int channels,
                                 const eds_kf_select* sel, int n_depth, const double* depth_xy, const double* depth_idp, double fx, double fy,
                                 double cx, double cy, int* n_points) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!img || !sel) return fail(EDS_ERR_INVALID, "null image or selection parameters");
    if (n_depth < 0) return fail(EDS_ERR_INVALID, "negative depth-map size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_keyframe_build(h, slot, img_type, img, img_H, img_W, channels, sel, n_depth, depth_xy, depth_idp, fx, fy, cx, cy, n_points);
}

int eds_trk_get_keyframe_points(eds_trk* h, int slot, double* coord_xy, double* norm_xy, double* grad_xy, double* idp, double* weights) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_keyframe_get_points(h, slot, coord_xy, norm_xy, grad_xy, idp, weights);
}

}  // extern "C"

int eds_internal_refresh_gram(eds_trk* h, int slot) { return refresh_gram(h, slot); }
