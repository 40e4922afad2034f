// C ABI of libeds_hip.so, second translation unit: what goes INTO a slot — keyframe points (eds_trk_set_keyframe), inverse depths
// (Tracker.cpp:167 re-reads them on every optimize), event frames as host buffers (the reference hands optimize a std::vector<double>)
// or built on the device from events (EventFrame.cpp:302-389), shared frames.  Nothing here waits for the GPU on the live path.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "eds_capi_internal.hpp"

using namespace edscapi;

// Row-major H x W host frame (double or float) -> the slot's frame in HBM.  The host only narrows to fp32 (a loop the compiler
// vectorises; no index arithmetic) into device-mapped pinned staging, in EDS_UPLOAD_BANDS bands of rows; behind every band a
// launch of k_store_rowmajor reads it over PCIe and writes tiles, padding and the replicated margin (= Grid2D's clamp), while
// the host narrows the next band.  Nothing is waited for: whatever uses the frame is ordered behind the launches on the handle's
// stream, and the staging buffer is private to this function (its event is waited for before the next frame overwrites it).
// (Round 1 built the tiled, margin-padded image element by element on one host thread: 300 us for 640x480, more than the solve;
// chunked hipMemcpyAsync into HBM + one tiling launch: 80 us, 30 of them after the host had finished.)
// slots that are about to receive a frame of their own stop sampling somebody else's
static int unshare_frames(eds_trk* h, int first, int count) {
    for (int s = first; s < first + count; ++s) {
        if (h->slots[s].frame_slot < 0) continue;
        h->slots[s].frame_slot = -1;
        fill_static(h, s);
        int rc = upload_pose(h, s, 1);
        if (rc) return rc;
    }
    return EDS_OK;
}
#define EDS_UPLOAD_BANDS 4
// fp64 -> fp32 narrowing of a band of the frame (set_event_frame's host work: 2.46 MB in, 1.23 MB out for 640x480).  The library is
// built without -march, so the plain loop is SSE2 — cvtpd2ps, two doubles per instruction, 36 us per VGA frame; the hosts of the pool
// (Zen 4 / 5) have AVX-512, older ones AVX2: pick at run time (function multiversioning by hand, __builtin_cpu_supports).
#if defined(__x86_64__)
#include <immintrin.h>
// (EDS_NARROW_NT: non-temporal stores into the staging buffer — the destination lines are written whole and read next by the GPU over
// PCIe, never by this core: without the read-for-ownership of every line the band is ~25 % less memory traffic.  Weakly ordered: an
// sfence closes every band before its progress is published.)
#ifndef EDS_NARROW_NT
#define EDS_NARROW_NT 1
#endif
__attribute__((target("avx512f"))) static void narrow_avx512(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = 0;
    if (EDS_NARROW_NT && (reinterpret_cast<uintptr_t>(dst) & 31) == 0) {
        for (; i + 16 <= n; i += 16) {
            _mm256_stream_ps(dst + i, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)));
            _mm256_stream_ps(dst + i + 8, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 8)));
        }
        _mm_sfence();
    } else
    for (; i + 16 <= n; i += 16) {
        _mm256_storeu_ps(dst + i, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)));
        _mm256_storeu_ps(dst + i + 8, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 8)));
    }
    for (; i < n; ++i) dst[i] = (float)src[i];
}
__attribute__((target("avx2"))) static void narrow_avx2(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = 0;
    if (EDS_NARROW_NT && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        for (; i + 8 <= n; i += 8) {
            _mm_stream_ps(dst + i, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i)));
            _mm_stream_ps(dst + i + 4, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4)));
        }
        _mm_sfence();
    } else
    for (; i + 8 <= n; i += 8) {
        _mm_storeu_ps(dst + i, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i)));
        _mm_storeu_ps(dst + i + 4, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4)));
    }
    for (; i < n; ++i) dst[i] = (float)src[i];
}
#endif
static void narrow_band(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
#if defined(__x86_64__)
    static const int level = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
    if (level == 2) return narrow_avx512(src, dst, n);
    if (level == 1) return narrow_avx2(src, dst, n);
#endif
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i];       // (round-to-nearest-even in every variant: bit-identical results)
}
static void narrow_band(const float* __restrict__ src, float* __restrict__ dst, size_t n) { std::memcpy(dst, src, n * sizeof(float)); }

template <class T>
static int upload_frame(eds_trk* h, int slot, const T* frame) {
    { int rc_ = unshare_frames(h, slot, 1); if (rc_) return rc_; }     // a frame of its own again
    float* stage = h->h_fstage;
    if (h->stage_busy) { EDS_HIP_TRY(hipEventSynchronize(h->ev_stage)); h->stage_busy = false; }   // the previous frame's reads (long done)
    const bool banded = h->knobs.upload_bands != 0;     // A/B knob: one launch per band (round 2)
    if (h->d_fprog && !banded && h->H < (1 << 20)) {
        // ONE launch (round 3): k_store_follow's workgroups wait for the rows they move; the host publishes its progress after every
        // band in a pinned word (release store behind the band's plain stores: x86 keeps them in order for the device's reads).
        // 16 bands: what is left after the host's last store is 1/16 of a frame over PCIe.  (4 launches cost the host 16 of its 44 us.)
        if (h->h_fprog[1] & 0x80000000u) { h->h_fprog[1] = 0; return fail(EDS_ERR_HIP, "the previous frame upload timed out waiting for the host"); }
        // bands of 32 k rows (a band boundary is then a multiple of 128 bytes into the staging buffer whatever W is), at most 16 of them
        const int rows_per = 32 * std::max(1, (h->H + 32 * 16 - 1) / (32 * 16)), nbands = (h->H + rows_per - 1) / rows_per;
        const unsigned seq = (++h->upload_seq) & 0xfffu;
        __atomic_store_n(&h->h_fprog[0], seq << 20, __ATOMIC_RELEASE);
        const auto t_host = std::chrono::steady_clock::now();
        eds_frame_store_follow(h, slot, seq, rows_per);
        for (int k = 0; k < nbands; ++k) {
            const int rb = rows_per * k, re = std::min(h->H, rows_per * (k + 1));
            const size_t b = (size_t)rb * h->W, e = (size_t)re * h->W;
            narrow_band(frame + b, stage + b, e - b);
            __atomic_store_n(&h->h_fprog[0], (seq << 20) | (unsigned)re, __ATOMIC_RELEASE);
        }
        // A workgroup of the follower gives up after 2 s without progress (a host thread that was descheduled mid-frame).  Only then can
        // the slot be half-written, and only if this loop took that long: in that case wait, and store the frame again from the (now
        // complete) staging buffer with plain launches — a slot is never left holding a partial frame, has_frame stays truthful, and a
        // pyramid built on it (k_pyr_down) sees whole levels.
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_host).count() > 1.0) {
            EDS_HIP_TRY(hipStreamSynchronize(h->st));
            if (h->h_fprog[1] & 0x80000000u) {
                h->h_fprog[1] = 0;
                for (int k = 0; k < nbands; ++k) eds_frame_store_rowmajor(h, slot, h->d_fstage, rows_per * k, std::min(h->H, rows_per * (k + 1)));
            }
        }
    } else
    for (int k = 0; k < EDS_UPLOAD_BANDS; ++k) {
        const int rb = h->H * k / EDS_UPLOAD_BANDS, re = h->H * (k + 1) / EDS_UPLOAD_BANDS;
        const size_t b = (size_t)rb * h->W, e = (size_t)re * h->W;
        narrow_band(frame + b, stage + b, e - b);
        eds_frame_store_rowmajor(h, slot, h->d_fstage, rb, re);
    }
    EDS_HIP_TRY(hipGetLastError());
    EDS_HIP_TRY(hipEventRecord(h->ev_stage, h->st));
    h->stage_busy = true;
    h->slots[slot].has_frame = true;
    ++h->slots[slot].frame_version;             // its strip copy (eds_strips.hip) is out of date
    return EDS_OK;
}

// MANY host frames in one call (round 5; VERDICT r4 Next #5): eds_trk_set_event_frame spends its time in ONE host thread narrowing
// fp64 to fp32 (34 us per VGA frame against ~25 us of PCIe), so a batch of 256 took 20 ms.  Here `threads` workers narrow the frames
// into a ring of pinned, device-mapped staging slots; this thread launches one store kernel per frame as it becomes ready (the kernel
// reads the staging slot over PCIe and writes the slot's tiles) and hands a staging slot back to the workers when the kernel that read
// it has finished (HIP events).  Only this thread talks to HIP.  Frames land bit-identical to the one-frame path (same narrowing, same
// store kernel).
// The thread's last error right behind a kernel launch.  hipErrorNotReady is not one: on ROCm < 7 the PREVIOUS iteration's event poll
// leaves it there (every API return becomes the last error), and a launch that succeeded does not clear it.
static inline hipError_t launch_error() {
    const hipError_t e = hipGetLastError();
    return e == hipErrorNotReady ? hipSuccess : e;
}

template <class T>
static int upload_frames_batch(eds_trk* h, int first, int count, const T* const* frames) {
    { int rc_ = unshare_frames(h, first, count); if (rc_) return rc_; }
    const size_t fe = (size_t)h->H * h->W;
    const int S = std::min(count, 16);
    if (h->bstage_slots < S) {
        if (h->h_bstage) { EDS_HIP_TRY(hipStreamSynchronize(h->st)); hipHostFree(h->h_bstage); h->h_bstage = nullptr; h->bstage_slots = 0; }
        EDS_HIP_TRY(hipHostMalloc((void**)&h->h_bstage, (size_t)S * fe * sizeof(float), hipHostMallocMapped));
        EDS_HIP_TRY(hipHostGetDevicePointer((void**)&h->d_bstage, h->h_bstage, 0));
        h->bstage_slots = S;
    }
    const bool dma = h->knobs.upload_dma != 0;
    const bool two_streams = !dma && h->knobs.upload_streams != 1 && count > 1;
    if (two_streams) {                  // the second stream starts behind everything the handle's stream holds so far ...
        if (!h->st_up) { EDS_HIP_TRY(hipStreamCreateWithFlags(&h->st_up, hipStreamNonBlocking)); EDS_HIP_TRY(hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming)); }
        EDS_HIP_TRY(hipEventRecord(h->ev_up, h->st));
        EDS_HIP_TRY(hipStreamWaitEvent(h->st_up, h->ev_up, 0));
    }
    if (dma && !h->d_bdev) EDS_HIP_TRY(hipMalloc((void**)&h->d_bdev, 2 * fe * sizeof(float)));
    while ((int)h->ev_bstage.size() < S) { hipEvent_t e; EDS_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ev_bstage.push_back(e); }
    if (h->bstage_busy) { EDS_HIP_TRY(hipStreamSynchronize(h->st)); h->bstage_busy = false; }      // an earlier batch's last reads of the ring
    int nthr = h->knobs.upload_threads > 0 ? h->knobs.upload_threads : 4;      // measured (tools/bench_upload_batch.py): two threads already keep PCIe busy; more only contend
    nthr = std::max(1, std::min(nthr, std::min(count, (int)std::max(1u, std::thread::hardware_concurrency()))));
    std::atomic<int> next{0}, released{S};
    std::vector<std::atomic<unsigned char>> ready(count);
    for (auto& r : ready) r.store(0, std::memory_order_relaxed);
    std::atomic<bool> abort{false};
    float* const stage = h->h_bstage;
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= count) return;
            for (int spin = 0; i >= released.load(std::memory_order_acquire); ++spin) {
                if (abort.load(std::memory_order_relaxed)) return;
                if (spin > 256) std::this_thread::yield();
            }
            narrow_band(frames[i], stage + (size_t)(i % S) * fe, fe);
            ready[i].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    try {                                           // (no exception may cross the C ABI: threads the system refuses are simply not used)
        pool.reserve(nthr);
        for (int t = 0; t < nthr; ++t) pool.emplace_back(work);
    } catch (...) { }
    const bool inline_work = pool.empty();          // not even one: this thread narrows every frame itself, just in front of its kernel
    int completed = 0, launched = 0;
    hipError_t err = hipSuccess;
    for (int i = 0; i < count && err == hipSuccess; ++i) {
        if (inline_work) {
            if (i >= S) { err = hipEventSynchronize(h->ev_bstage[i % S]); if (err != hipSuccess) break; }
            narrow_band(frames[i], stage + (size_t)(i % S) * fe, fe);
            ready[i].store(1, std::memory_order_release);
        }
        for (int spin = 0; !ready[i].load(std::memory_order_acquire); ++spin) if (spin > 256) std::this_thread::yield();
        if (dma) {          // copy engine: pinned staging slot -> row-major scratch in HBM, then the store kernel reads HBM
            err = hipMemcpyAsync(h->d_bdev + (size_t)(i % 2) * fe, stage + (size_t)(i % S) * fe, fe * sizeof(float), hipMemcpyHostToDevice, h->st);
            if (err == hipSuccess) err = hipEventRecord(h->ev_bstage[i % S], h->st);      // the staging slot is free once the copy is through
            eds_frame_store_whole(h, first + i, h->d_bdev + (size_t)(i % 2) * fe, h->st);
            if (err == hipSuccess) err = launch_error();                                  // the launch's own error, asked for BEFORE any event is polled
        } else {            // the store kernel reads the staging slot over PCIe; consecutive frames alternate between two streams, so that one
            hipStream_t sx = two_streams && (i & 1) ? h->st_up : h->st;                  // kernel's tail overlaps the next one's ramp-up
            eds_frame_store_whole(h, first + i, h->d_bstage + (size_t)(i % S) * fe, sx);
            err = launch_error();                   // the launch's own error: checked here, not behind the polls below (hipEventQuery's hipErrorNotReady
            if (err == hipSuccess) err = hipEventRecord(h->ev_bstage[i % S], sx);        // becomes the thread's last error on ROCm < 7: ADVICE r5)
        }
        launched = i + 1;
        // staging slots whose kernel has finished go back to the workers; when none is free and frames are still to be staged: wait for the oldest
        while (completed <= i && hipEventQuery(h->ev_bstage[completed % S]) == hipSuccess) ++completed;
        if (completed + S <= i + 1 && i + 1 < count && err == hipSuccess) { err = hipEventSynchronize(h->ev_bstage[completed % S]); ++completed; }
        released.store(completed + S, std::memory_order_release);
    }
    if (err != hipSuccess) abort.store(true);
    for (auto& t : pool) t.join();
    if (two_streams && err == hipSuccess) {      // ... and the handle's stream goes on behind the second one's last kernel
        err = hipEventRecord(h->ev_up, h->st_up);
        if (err == hipSuccess) err = hipStreamWaitEvent(h->st, h->ev_up, 0);
    }
    if (err != hipSuccess) {
        // Both streams are drained (kernels on st_up may still read the ring or write slots), and every slot whose store kernel was
        // launched may hold a partial frame: it no longer counts as holding one, and its strip copy is out of date.
        hipStreamSynchronize(h->st);
        if (h->st_up) hipStreamSynchronize(h->st_up);
        (void)hipGetLastError();
        h->bstage_busy = false;
        for (int s = first; s < first + launched; ++s) { h->slots[s].has_frame = false; ++h->slots[s].frame_version; }
        return fail(EDS_ERR_HIP, hipGetErrorString(err));
    }
    h->bstage_busy = true;
    for (int s = first; s < first + count; ++s) { h->slots[s].has_frame = true; ++h->slots[s].frame_version; }
    return EDS_OK;
}

static int upload_points(eds_trk* h, int slot, int N, const double* norm_xy, const double* grad_xy, const double* idp,
                         const double* w) {
    const size_t off = (size_t)slot * h->Np;
    const Slot& s = h->slots[slot];
    float* f = h->h_f32;
    const int Np = h->Np;
    int* cell = reinterpret_cast<int*>(f + (size_t)8 * Np);
    for (int i = 0; i < Np; ++i) {
        const bool in = i < N;
        f[0 * Np + i] = in ? (float)norm_xy[2 * i] : 0.f;
        f[1 * Np + i] = in ? (float)norm_xy[2 * i + 1] : 0.f;
        f[2 * Np + i] = in ? (float)idp[i] : 1.f;
        if (grad_xy) { f[3 * Np + i] = in ? (float)grad_xy[2 * i] : 0.f; f[4 * Np + i] = in ? (float)grad_xy[2 * i + 1] : 0.f; }
        if (w) f[5 * Np + i] = in ? (float)w[i] : 0.f;
        // the point's own keyframe pixel u0 = fx x + cx, v0 = fy y + cy in fp64, split into an integer
        // cell and an fp32 fraction: the kernels only ever add a small displacement to it (eds_device.hpp)
        double u0 = in ? s.K[0] * norm_xy[2 * i] + s.K[2] : 0.0, v0 = in ? s.K[1] * norm_xy[2 * i + 1] + s.K[3] : 0.0;
        double cu = std::floor(u0), cv = std::floor(v0);
        if (!(cu > -32000.0)) cu = -32000.0; if (cu > 32000.0) cu = 32000.0;     // far-off points keep the excess in the fraction
        if (!(cv > -32000.0)) cv = -32000.0; if (cv > 32000.0) cv = 32000.0;
        f[6 * Np + i] = (float)(u0 - cu);
        f[7 * Np + i] = (float)(v0 - cv);
        cell[i] = (int)(((unsigned)(int)cv << 16) | ((unsigned)(int)cu & 0xffffu));
    }
    float* dst[8] = {h->dx, h->dy, h->drho, h->dgx, h->dgy, h->dw, h->df0x, h->df0y};
    for (int k = 0; k < 8; ++k) {
        if ((k == 3 || k == 4) && !grad_xy) continue;
        if (k == 5 && !w) continue;
        EDS_HIP_TRY(hipMemcpyAsync(dst[k] + off, f + (size_t)k * Np, (size_t)Np * 4, hipMemcpyHostToDevice, h->st));
    }
    EDS_HIP_TRY(hipMemcpyAsync(h->dcell0 + off, cell, (size_t)Np * 4, hipMemcpyHostToDevice, h->st));
    return EDS_OK;
}

namespace edscapi {
int refresh_gram(eds_trk* h, int slot, bool wait) {
    fill_static(h, slot);
    int rc = upload_pose(h, slot, 1);
    if (rc) return rc;
    eds_launch_gram(h->arrays(), slot, effective_blocks(h), h->st);
    EDS_HIP_TRY(hipGetLastError());
    const size_t off = (size_t)slot * EDS_MAX_BLOCKS * 36;
    EDS_HIP_TRY(hipMemcpyAsync(h->h_G + off, h->dG + off, (size_t)EDS_MAX_BLOCKS * 36 * 8, hipMemcpyDeviceToHost, h->st));
    h->slots[slot].gram_host_stale = false;
    if (wait) { EDS_HIP_TRY(hipStreamSynchronize(h->st)); h->gram_pending = false; }
    else h->gram_pending = true;        // the device solvers read dG on the stream; host readers of h_G wait in fill_pose
    return EDS_OK;
}
}  // namespace edscapi

extern "C" {

int eds_trk_set_keyframe(eds_trk* h, int slot, int N, const double* norm_xy, const double* grad_xy, const double* idp,
                         const double* w, double fx, double fy, double cx, double cy) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (N < 1 || N > h->Nmax) return fail(EDS_ERR_INVALID, "N out of range for this handle");
    if (!norm_xy || !grad_xy || !idp || !w) return fail(EDS_ERR_INVALID, "null keyframe array");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    Slot& s = h->slots[slot];
    s.N = N; s.K[0] = fx; s.K[1] = fy; s.K[2] = cx; s.K[3] = cy;
    if ((rc = upload_points(h, slot, N, norm_xy, grad_xy, idp, w))) return rc;
    if ((rc = refresh_gram(h, slot))) return rc;
    s.has_kf = true;
    // residuals and trace of an earlier solve belong to the previous keyframe: drop the host copy AND the "still in HBM" marks,
    // so that get_residuals / loss_param before the next optimize report EDS_ERR_STATE instead of another keyframe's plane
    s.residuals.clear();
    s.res_on_device = false; s.trace_on_device = false; s.ntrace = 0;
    return EDS_OK;
}

int eds_trk_set_idepth(eds_trk* h, int slot, int N, const double* idp) { return eds_trk_set_idepth_strided(h, slot, N, idp, 1); }

int eds_trk_set_idepth_strided(eds_trk* h, int slot, int N, const double* idp, int stride) {
    if (stride < 1) return fail(EDS_ERR_INVALID, "stride must be at least 1");
    int rc = check_slot(h, slot);
    if (rc) return rc;
    Slot& s = h->slots[slot];
    if (!s.has_kf) return fail(EDS_ERR_STATE, "keyframe not set");
    if (N != s.N || !idp) return fail(EDS_ERR_INVALID, "idp size mismatch");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    // only the inverse-depth plane changes (the geometry uses rho' = idp + 1e-5, the model the raw idp)
    // (Tracker.cpp:167 re-reads the depths on every optimize: this is on the live path, so nothing here waits for the GPU)
    if (h->idp_busy) { EDS_HIP_TRY(hipEventSynchronize(h->ev_idp)); h->idp_busy = false; }
    for (int i = 0; i < h->Np; ++i) h->h_idp[i] = i < N ? (float)idp[(size_t)i * stride] : 1.f;
    if (h->d_idp) {
        // ONE launch: the Gram kernel reads the new depths out of the mapped staging, stores them into the rho plane on its way and
        // leaves the Gram matrices in HBM, where the device solvers read them; the host copy is fetched only if a host-side solver or
        // eval asks for it (fill_pose).  (Round 2: copy + event + pose upload + launch + copy back = ~20 us of host time.)
        eds_launch_gram(h->arrays(), slot, effective_blocks(h), h->st, h->d_idp);
        EDS_HIP_TRY(hipGetLastError());
        EDS_HIP_TRY(hipEventRecord(h->ev_idp, h->st));
        h->idp_busy = true;
        s.gram_host_stale = true;
        return EDS_OK;
    }
    EDS_HIP_TRY(hipMemcpyAsync(h->drho + (size_t)slot * h->Np, h->h_idp, (size_t)h->Np * 4, hipMemcpyHostToDevice, h->st));
    EDS_HIP_TRY(hipEventRecord(h->ev_idp, h->st));
    h->idp_busy = true;
    return refresh_gram(h, slot, false);
}

int eds_trk_set_event_frame(eds_trk* h, int slot, const double* frame) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!frame) return fail(EDS_ERR_INVALID, "null frame");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return upload_frame(h, slot, frame);
}

int eds_trk_set_event_frame_f32(eds_trk* h, int slot, const float* frame) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!frame) return fail(EDS_ERR_INVALID, "null frame");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return upload_frame(h, slot, frame);
}

int eds_trk_set_event_frames(eds_trk* h, int first, int count, const double* const* frames) {
    if (!h || !frames) return fail(EDS_ERR_INVALID, "null argument");
    if (first < 0 || count < 1 || first + count > h->B) return fail(EDS_ERR_INVALID, "bad slot range");
    for (int i = 0; i < count; ++i) if (!frames[i]) return fail(EDS_ERR_INVALID, "null frame");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return upload_frames_batch(h, first, count, frames);
}

int eds_trk_set_event_frames_f32(eds_trk* h, int first, int count, const float* const* frames) {
    if (!h || !frames) return fail(EDS_ERR_INVALID, "null argument");
    if (first < 0 || count < 1 || first + count > h->B) return fail(EDS_ERR_INVALID, "bad slot range");
    for (int i = 0; i < count; ++i) if (!frames[i]) return fail(EDS_ERR_INVALID, "null frame");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return upload_frames_batch(h, first, count, frames);
}

int eds_trk_set_undistort_map(eds_trk* h, const float* mapx, const float* mapy) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if ((mapx == nullptr) != (mapy == nullptr)) return fail(EDS_ERR_INVALID, "mapx and mapy must both be given or both be NULL");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_frame_set_map(h, mapx, mapy, h->H, h->W);
}

int eds_trk_set_undistort_map_sized(eds_trk* h, const float* mapx, const float* mapy, int sensor_H, int sensor_W) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if ((mapx == nullptr) != (mapy == nullptr)) return fail(EDS_ERR_INVALID, "mapx and mapy must both be given or both be NULL");
    if (mapx && (sensor_H < 1 || sensor_W < 1)) return fail(EDS_ERR_INVALID, "bad sensor size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_frame_set_map(h, mapx, mapy, sensor_H, sensor_W);
}

int eds_trk_build_event_frame(eds_trk* h, int slot, int n_events, const uint16_t* x, const uint16_t* y, const uint8_t* polarity,
                              int level, double blur_sigma, int use_exp_weights, double* norm_out) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (n_events < 0 || level < 0 || level > 16) return fail(EDS_ERR_INVALID, "bad event count or level");
    if (n_events > 0 && (!x || !y || !polarity)) return fail(EDS_ERR_INVALID, "null event array");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    if ((rc = unshare_frames(h, slot, 1))) return rc;
    return eds_frame_build_levels(h, slot, level, 1, n_events, x, y, polarity, h->H, h->W, blur_sigma, use_exp_weights, norm_out);
}

int eds_trk_build_event_frames(eds_trk* h, int first_slot, int num_levels, int n_events, const uint16_t* x, const uint16_t* y,
                               const uint8_t* polarity, int sensor_H, int sensor_W, double blur_sigma, int use_exp_weights, double* norms) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (num_levels < 1 || num_levels > EDS_MAX_LEVELS) return fail(EDS_ERR_INVALID, "num_levels out of range");
    if (first_slot < 0 || first_slot + num_levels > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds (one slot per level)");
    if (n_events < 0) return fail(EDS_ERR_INVALID, "bad event count");
    if (n_events > 0 && (!x || !y || !polarity)) return fail(EDS_ERR_INVALID, "null event array");
    if (sensor_H <= 0 || sensor_W <= 0) { sensor_H = h->H; sensor_W = h->W; }
    if (sensor_H < 2 || sensor_W < 2) return fail(EDS_ERR_INVALID, "bad sensor size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    { int rc_ = unshare_frames(h, first_slot, num_levels); if (rc_) return rc_; }
    return eds_frame_build_levels(h, first_slot, 0, num_levels, n_events, x, y, polarity, sensor_H, sensor_W, blur_sigma, use_exp_weights, norms);
}

int eds_trk_build_event_frames_aos(eds_trk* h, int first_slot, int num_levels, int n_events, const void* events, int stride, int off_x,
                                   int off_y, int off_polarity, int sensor_H, int sensor_W, double blur_sigma, int use_exp_weights, double* norms) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (num_levels < 1 || num_levels > EDS_MAX_LEVELS) return fail(EDS_ERR_INVALID, "num_levels out of range");
    if (first_slot < 0 || first_slot + num_levels > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds (one slot per level)");
    if (n_events < 0 || (n_events > 0 && !events)) return fail(EDS_ERR_INVALID, "bad event array");
    if (stride < 5 || off_x < 0 || off_y < 0 || off_polarity < 0 || off_x + 2 > stride || off_y + 2 > stride || off_polarity + 1 > stride ||
        (off_x & 1) || (off_y & 1) || (stride & 1))
        return fail(EDS_ERR_INVALID, "bad event layout (x, y: 2-byte aligned uint16 fields inside an even stride)");
    if (sensor_H <= 0 || sensor_W <= 0) { sensor_H = h->H; sensor_W = h->W; }
    if (sensor_H < 2 || sensor_W < 2) return fail(EDS_ERR_INVALID, "bad sensor size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    { int rc_ = unshare_frames(h, first_slot, num_levels); if (rc_) return rc_; }
    const EdsEventAos aos = {events, stride, off_x, off_y, off_polarity};
    return eds_frame_build_levels(h, first_slot, 0, num_levels, n_events, nullptr, nullptr, nullptr, sensor_H, sensor_W, blur_sigma, use_exp_weights,
                                  norms, &aos);
}

// EventFrame::create's time bookkeeping (EventFrame.cpp:313-335), host only.  The reference object is STATEFUL: clear() (called at the head
// of create) does not touch first_time / last_time, and last_time is only assigned in the `else if ((it + 1) == events.end())` branch — so a
// slice of ONE event keeps the PREVIOUS slice's last_time, and both the order check and delta_time use that.  `out->last_time` is therefore
// in/out: on entry the previous slice's last_time (0 on a fresh object), on return this slice's — unchanged for a single event
// (last_valid = 0) and for an empty slice.
int eds_event_times_aos(int n_events, const void* events, int stride, int off_ts, eds_event_times* out) {
    if (!out) return fail(EDS_ERR_INVALID, "null output");
    const int64_t prev_last = out->last_time;
    std::memset(out, 0, sizeof(*out));
    out->last_time = prev_last;
    if (n_events < 0 || (n_events > 0 && !events)) return fail(EDS_ERR_INVALID, "bad event array");
    if (stride < 8 || off_ts < 0 || off_ts + 8 > stride) return fail(EDS_ERR_INVALID, "bad event layout (ts: int64 field inside the stride)");
    if (n_events == 0) { out->delta_time = out->last_time; return EDS_OK; }   // (the loop does not run: first_time stays as well — reported as 0 here, the caller holds the state)
    auto ts = [&](int i) { int64_t t; std::memcpy(&t, static_cast<const char*>(events) + (size_t)i * stride + off_ts, 8); return t; };
    out->first_time = ts(0);
    if (n_events > 1) { out->last_time = ts(n_events - 1); out->last_valid = 1; }     // `else if ((it + 1) == events.end())`: never for a single event
    if (out->first_time > out->last_time)
        return fail(EDS_ERR_INVALID, "[EVENT_FRAME] Event time[0] > event time [N-1] (EventFrame.cpp:325-329)");
    out->time = ts(n_events / 2);
    out->delta_time = out->last_time - out->first_time;
    return EDS_OK;
}

int eds_trk_build_event_frames_aos_timed(eds_trk* h, int first_slot, int num_levels, int n_events, const void* events, int stride, int off_x,
                                         int off_y, int off_polarity, int off_ts, int sensor_H, int sensor_W, double blur_sigma,
                                         int use_exp_weights, double* norms, eds_event_times* times) {
    eds_event_times local;
    std::memset(&local, 0, sizeof(local));           // (last_time is in/out: a caller without a struct of its own has no history)
    int rc = eds_event_times_aos(n_events, events, stride, off_ts, times ? times : &local);
    if (rc) return rc;
    return eds_trk_build_event_frames_aos(h, first_slot, num_levels, n_events, events, stride, off_x, off_y, off_polarity, sensor_H, sensor_W,
                                          blur_sigma, use_exp_weights, norms);
}

int eds_trk_build_event_frame_batch(eds_trk* h, int first_slot, int count, const int* offsets, const uint16_t* x, const uint16_t* y,
                                    const uint8_t* polarity, int level, double blur_sigma, int use_exp_weights, double* norms) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (count < 1 || first_slot < 0 || first_slot + count > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds");
    if (!offsets || offsets[0] < 0) return fail(EDS_ERR_INVALID, "bad offsets");
    if (level < 0 || level > 16) return fail(EDS_ERR_INVALID, "bad level");
    if (offsets[count] > offsets[0] && (!x || !y || !polarity)) return fail(EDS_ERR_INVALID, "null event array");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    { int rc_ = unshare_frames(h, first_slot, count); if (rc_) return rc_; }
    return eds_frame_build_batch(h, first_slot, count, offsets, x, y, polarity, level, blur_sigma, use_exp_weights, norms);
}

int eds_trk_share_event_frame(eds_trk* h, int slot, int src_slot) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if ((rc = check_slot(h, src_slot))) return rc;
    Slot& s = h->slots[slot];
    const Slot& src = h->slots[src_slot];
    if (src.frame_slot >= 0 && src_slot != slot) return fail(EDS_ERR_INVALID, "the source slot itself shares another slot's frame");
    if (!src.has_frame && src_slot != slot) return fail(EDS_ERR_STATE, "the source slot has no event frame yet");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    s.frame_slot = src_slot == slot ? -1 : src_slot;
    if (src_slot != slot) s.has_frame = true;
    fill_static(h, slot);
    return upload_pose(h, slot, 1);                             // ordered before the next solve on the handle's stream
}

int eds_trk_get_event_frame(eds_trk* h, int slot, double* frame) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!frame) return fail(EDS_ERR_INVALID, "null output");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    const size_t n = (size_t)h->Hp * h->Wp;
    EDS_HIP_TRY(hipStreamSynchronize(h->st));                    // set_event_frame does not wait for its own upload
    const int fs = h->slots[slot].frame_slot >= 0 ? h->slots[slot].frame_slot : slot;      // a sharing slot: the frame it samples
    EDS_HIP_TRY(hipMemcpy(h->h_f32, h->dframe + (size_t)fs * n, n * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < h->H; ++r)
        for (int c = 0; c < h->W; ++c) frame[(size_t)r * h->W + c] = h->h_f32[eds_frame_index(r, c, h->Wp, h->tiled)];
    return EDS_OK;
}

}  // extern "C"
