// Coarse-to-fine tracking on an image pyramid (BASELINE.json configs[3]; an extension patterned on the DSO-derived coarse tracker
// the reference carries: pyramid by 2x2 box averaging, reference src/tracking/HessianBlocks.cpp:173-176; per-level intrinsics
// fx_l = fx_{l-1} / 2, cx_l = (cx_0 + 0.5) / 2^l - 0.5, src/tracking/CoarseTracker.cpp:103-111; coarsest level first, the pose
// carried from level to level, CoarseTracker.cpp:545-664).
//
// An eds_pyr owns one eds_trk handle per level (each with its own frame size and point set — the alignment itself is the same
// persistent kernel at every level).  The event frame goes up once, at level 0; the coarser frames are built on the device by
// k_pyr_down straight into the levels' tiled fp32 frame storage (margins replicated like every other frame writer), so a level
// costs 1/4 of the previous one in HBM and nothing over PCIe.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/eds_hip.h"
#include "eds_handle.hpp"

namespace {

// dst(r, c) = 0.25f * (src(2r, 2c) + src(2r, 2c+1) + src(2r+1, 2c) + src(2r+1, 2c+1)), in that order, in fp32 — for every element
// of the destination ALLOCATION: margin elements take the value of the nearest image pixel (Grid2D's clamp, eds_layout.hpp).
__global__ void k_pyr_down(const float* __restrict__ src, int sHp, int sWp, int s_tiled, float* __restrict__ dst, int dH, int dW, int dHp, int dWp,
                           int d_tiled, int first_slot) {
    const int cc = blockIdx.x * blockDim.x + threadIdx.x, rr = blockIdx.y * blockDim.y + threadIdx.y;     // allocation coordinates
    if (rr >= dHp || cc >= dWp) return;
    src += (size_t)(first_slot + blockIdx.z) * sHp * sWp;          // blockIdx.z: the pyramid (slot) of a batched eds_pyr
    dst += (size_t)(first_slot + blockIdx.z) * dHp * dWp;
    const int r = min(max(rr - EDS_FRAME_MARGIN, 0), dH - 1), c = min(max(cc - EDS_FRAME_MARGIN, 0), dW - 1);
    const float a = src[eds_frame_index(2 * r, 2 * c, sWp, s_tiled)], b = src[eds_frame_index(2 * r, 2 * c + 1, sWp, s_tiled)];
    const float cpix = src[eds_frame_index(2 * r + 1, 2 * c, sWp, s_tiled)], d = src[eds_frame_index(2 * r + 1, 2 * c + 1, sWp, s_tiled)];
    dst[eds_frame_index(rr - EDS_FRAME_MARGIN, cc - EDS_FRAME_MARGIN, dWp, d_tiled)] = 0.25f * (((a + b) + cpix) + d);
}

}  // namespace

struct eds_pyr {
    int levels = 0, batch = 1;
    eds_trk* lv[EDS_MAX_LEVELS] = {nullptr};
    double K0[4] = {0, 0, 0, 0};
    bool has_frame = false;
    std::vector<char> slot_has_frame;   // batched pyramids: per slot
    hipEvent_t ev_levels = nullptr;     // recorded behind the last down-sampling launch
};

extern "C" {

int eds_pyr_create(const eds_trk_cfg* cfg, int levels, const int* max_points, int H, int W, eds_pyr** out) {
    return eds_pyr_create_batch(cfg, 1, levels, max_points, H, W, out);
}

// `batch` independent pyramids in one object: level l is ONE handle of `batch` slots, so that a level of all pyramids is one launch
int eds_pyr_create_batch(const eds_trk_cfg* cfg, int batch, int levels, const int* max_points, int H, int W, eds_pyr** out) {
    if (!cfg || !out || !max_points) return eds_internal_fail(EDS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (batch < 1) return eds_internal_fail(EDS_ERR_INVALID, "batch out of range");
    if (levels < 1 || levels > EDS_MAX_LEVELS) return eds_internal_fail(EDS_ERR_INVALID, "levels out of range");
    if ((H >> (levels - 1)) < 4 || (W >> (levels - 1)) < 4) return eds_internal_fail(EDS_ERR_INVALID, "coarsest level smaller than 4x4");
    eds_pyr* p = new (std::nothrow) eds_pyr();
    if (!p) return eds_internal_fail(EDS_ERR_INVALID, "out of memory");
    p->levels = levels; p->batch = batch;
    p->slot_has_frame.assign(batch, 0);
    for (int l = 0; l < levels; ++l) {
        int rc = eds_trk_create(cfg, batch, max_points[l], H >> l, W >> l, &p->lv[l]);
        if (rc != EDS_OK) { for (int k = 0; k < l; ++k) eds_trk_destroy(p->lv[k]); delete p; return rc; }
    }
    if (hipEventCreateWithFlags(&p->ev_levels, hipEventDisableTiming) != hipSuccess) {
        for (int k = 0; k < levels; ++k) eds_trk_destroy(p->lv[k]);
        delete p;
        return eds_internal_fail(EDS_ERR_HIP, "hipEventCreate");
    }
    *out = p;
    return EDS_OK;
}

void eds_pyr_destroy(eds_pyr* p) {
    if (!p) return;
    for (int l = 0; l < p->levels; ++l) eds_trk_destroy(p->lv[l]);
    if (p->ev_levels) hipEventDestroy(p->ev_levels);
    delete p;
}

int eds_pyr_set_config(eds_pyr* p, const eds_trk_cfg* cfg) {
    if (!p || !cfg) return eds_internal_fail(EDS_ERR_INVALID, "null argument");
    for (int l = 0; l < p->levels; ++l) { int rc = eds_trk_set_config(p->lv[l], cfg); if (rc) return rc; }
    return EDS_OK;
}

int eds_pyr_level_intrinsics(int level, double fx0, double fy0, double cx0, double cy0, double K[4]) {
    if (level < 0 || level >= EDS_MAX_LEVELS || !K) return eds_internal_fail(EDS_ERR_INVALID, "bad level");
    double fx = fx0, fy = fy0;
    for (int l = 0; l < level; ++l) { fx *= 0.5; fy *= 0.5; }                       // CoarseTracker.cpp:107-108
    K[0] = fx; K[1] = fy;
    K[2] = level ? (cx0 + 0.5) / (double)(1 << level) - 0.5 : cx0;                  // :109-110
    K[3] = level ? (cy0 + 0.5) / (double)(1 << level) - 0.5 : cy0;
    return EDS_OK;
}

int eds_pyr_set_keyframe(eds_pyr* p, int level, int N, const double* norm_xy, const double* grad_xy, const double* idp, const double* w,
                         double fx0, double fy0, double cx0, double cy0) {
    return eds_pyr_set_keyframe_slot(p, 0, level, N, norm_xy, grad_xy, idp, w, fx0, fy0, cx0, cy0);
}
int eds_pyr_set_keyframe_slot(eds_pyr* p, int slot, int level, int N, const double* norm_xy, const double* grad_xy, const double* idp, const double* w,
                              double fx0, double fy0, double cx0, double cy0) {
    if (!p) return eds_internal_fail(EDS_ERR_INVALID, "null handle");
    if (level < 0 || level >= p->levels) return eds_internal_fail(EDS_ERR_INVALID, "level out of range");
    if (slot < 0 || slot >= p->batch) return eds_internal_fail(EDS_ERR_INVALID, "slot out of range");
    double K[4];
    eds_pyr_level_intrinsics(level, fx0, fy0, cx0, cy0, K);
    return eds_trk_set_keyframe(p->lv[level], slot, N, norm_xy, grad_xy, idp, w, K[0], K[1], K[2], K[3]);
}

// Levels 1 .. L-1 from level 0, all on level 0's stream (whatever wrote level 0 — set_event_frame's band launches, the event-frame
// builder — is on that stream too), one launch behind the other; the other levels' streams then wait for ONE event.  The host
// waits for nothing: a level's solve is ordered behind its frame on the level's own stream.
static int build_levels(eds_pyr* p, int first = 0, int count = 1) {
    hipStream_t st0 = p->lv[0]->st;
    hipError_t e = hipSetDevice(p->lv[0]->dev);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    for (int l = 1; l < p->levels; ++l) {
        eds_trk* s = p->lv[l - 1];
        eds_trk* d = p->lv[l];
        const dim3 b(32, 8), g((d->Wp + 31) / 32, (d->Hp + 7) / 8, count);
        hipLaunchKernelGGL(k_pyr_down, g, b, 0, st0, s->dframe, s->Hp, s->Wp, s->tiled, d->dframe, d->H, d->W, d->Hp, d->Wp, d->tiled, first);
        for (int k = first; k < first + count; ++k) { d->slots[k].has_frame = true; ++d->slots[k].frame_version; }
    }
    e = hipGetLastError();
    if (e == hipSuccess && p->levels > 1) {
        e = hipEventRecord(p->ev_levels, st0);
        for (int l = 1; l < p->levels && e == hipSuccess; ++l) e = hipStreamWaitEvent(p->lv[l]->st, p->ev_levels, 0);
    }
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    for (int k = first; k < first + count; ++k) p->slot_has_frame[k] = 1;
    p->has_frame = true;
    return EDS_OK;
}

int eds_pyr_set_event_frame(eds_pyr* p, const double* frame) { return eds_pyr_set_event_frame_slot(p, 0, frame); }
int eds_pyr_set_event_frame_slot(eds_pyr* p, int slot, const double* frame) {
    if (!p || !frame) return eds_internal_fail(EDS_ERR_INVALID, "null argument");
    if (slot < 0 || slot >= p->batch) return eds_internal_fail(EDS_ERR_INVALID, "slot out of range");
    int rc = eds_trk_set_event_frame(p->lv[0], slot, frame);
    if (rc) return rc;
    return build_levels(p, slot, 1);
}

int eds_pyr_build_event_frame(eds_pyr* p, int n_events, const uint16_t* x, const uint16_t* y, const uint8_t* polarity, double blur_sigma,
                              int use_exp_weights, double* norm_out) {
    if (!p) return eds_internal_fail(EDS_ERR_INVALID, "null handle");
    int rc = eds_trk_build_event_frame(p->lv[0], 0, n_events, x, y, polarity, 0, blur_sigma, use_exp_weights, norm_out);
    if (rc) return rc;
    return build_levels(p);
}

int eds_pyr_get_level_frame(eds_pyr* p, int level, double* frame) {
    if (!p) return eds_internal_fail(EDS_ERR_INVALID, "null handle");
    if (level < 0 || level >= p->levels) return eds_internal_fail(EDS_ERR_INVALID, "level out of range");
    return eds_trk_get_event_frame(p->lv[level], 0, frame);
}

int eds_pyr_level_size(const eds_pyr* p, int level, int* H, int* W) {
    if (!p || level < 0 || level >= p->levels) return eds_internal_fail(EDS_ERR_INVALID, "level out of range");
    if (H) *H = p->lv[level]->H;
    if (W) *W = p->lv[level]->W;
    return EDS_OK;
}

int eds_pyr_optimize(eds_pyr* p, double pp[3], double q[4], double v[6], eds_trk_info* infos) {
    if (!p || !pp || !q || !v) return eds_internal_fail(EDS_ERR_INVALID, "null argument");
    if (!p->has_frame) return eds_internal_fail(EDS_ERR_STATE, "event frame not set");
    double cp[3], cq[4], cv[6];
    std::memcpy(cp, pp, sizeof(cp)); std::memcpy(cq, q, sizeof(cq)); std::memcpy(cv, v, sizeof(cv));
    int last_rc = EDS_OK;
    for (int l = p->levels - 1; l >= 0; --l) {                      // coarsest first; `level` also indexes max_num_iterations
        eds_trk_info info;
        std::memset(&info, 0, sizeof(info));
        const int rc = eds_trk_optimize(p->lv[l], 0, l, cp, cq, cv, &info);     // on EDS_ERR_NOT_USABLE the state is left where it was:
        if (infos) infos[l] = info;                                            // the next finer level starts from the last good pose
        if (rc != EDS_OK && rc != EDS_ERR_NOT_USABLE) return rc;
        if (l == 0) last_rc = rc;
    }
    if (last_rc == EDS_OK) { std::memcpy(pp, cp, sizeof(cp)); std::memcpy(q, cq, sizeof(cq)); std::memcpy(v, cv, sizeof(cv)); }
    return last_rc;
}

// The same for pyramids [first, first + count) of a batched object: P, Q, V are count x 3 / 4 / 6 in and out; every level of all
// of them is one eds_trk_optimize_batch launch, the states carried on the host between the levels (104 B per pyramid).
// infos (optional): levels x count, level-major.  A pyramid whose level fails keeps its last good pose for the next level.
int eds_pyr_optimize_batch(eds_pyr* p, int first, int count, double* P, double* Q, double* V, eds_trk_info* infos) {
    if (!p || !P || !Q || !V) return eds_internal_fail(EDS_ERR_INVALID, "null argument");
    if (first < 0 || count < 1 || first + count > p->batch) return eds_internal_fail(EDS_ERR_INVALID, "range out of bounds");
    for (int k = first; k < first + count; ++k)
        if (!p->slot_has_frame[k]) return eds_internal_fail(EDS_ERR_STATE, "event frame not set");
    for (int l = p->levels - 1; l >= 0; --l) {
        eds_trk* h = p->lv[l];
        int rc = eds_trk_set_states(h, first, count, P, Q, V);
        if (rc) return rc;
        rc = eds_trk_optimize_batch(h, l, first, count);
        if (rc) return rc;
        if ((rc = eds_trk_sync(h))) return rc;
        if ((rc = eds_trk_get_states(h, first, count, P, Q, V))) return rc;     // a failed slot still holds the state it was given
        if (infos)
            for (int k = 0; k < count; ++k) eds_trk_get_info(h, first + k, &infos[(size_t)l * count + k]);
    }
    return EDS_OK;
}

int eds_pyr_get_residuals(eds_pyr* p, int level, double* r) {
    if (!p || level < 0 || level >= p->levels) return eds_internal_fail(EDS_ERR_INVALID, "level out of range");
    return eds_trk_get_residuals(p->lv[level], 0, r);
}

}  // extern "C"
