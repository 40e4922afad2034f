// Strip copies of the event frames (eds_layout.hpp): the layout the persistent pose-only kernel gathers from since round 3.
//
// The frame writers (set_event_frame, the event-frame builders, the pyramid) keep writing 4x4 tiles — every other kernel samples
// those — and bump the slot's frame_version.  A solve asks eds_strips_for_solve whether to gather from the copies: they are made for
// frames that are solved AGAIN (the rule and its arithmetic: below), by eds_strips_prepare — one launch per run of stale slots,
// every thread moving one aligned 16-byte tile row into its place in a strip (per 640x480 frame and row phase: 2.5 MB written,
// 1.0 us of the GPU with one phase, 2.9-3.2 us with four).  eds_trk_prepare_frames makes them on request.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_layout.hpp"

// grid: (pieces of one frame's strips / 256, slots of the run); copy index = 2 * row phase + column copy
__global__ __launch_bounds__(256) void k_tiles_to_strips(const float* __restrict__ tiles, float* __restrict__ strips, int first, int Hp, int Wp, int phases) {
    const int slot = first + blockIdx.y;
    const int NS = eds_strips_count(Wp), TW = Wp >> 2;
    const int per_copy = NS * Hp * 2;                       // 16-byte pieces per copy
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * phases * per_copy) return;
    const int cidx = i / per_copy, k = i - cidx * per_copy;
    const int copy = cidx & 1, p = cidx >> 1;
    const int half = k & 1, pos = (k >> 1) % Hp, strip = (k >> 1) / Hp;
    int row = pos + p;                                      // copy p holds allocation row r at position r - p
    if (row > Hp - 1) row = Hp - 1;                         // (its last p positions are never sampled)
    int col0 = 8 * strip + 4 * copy + 4 * half;             // allocation column of the piece's first pixel: a multiple of 4 = one tile row
    if (col0 > Wp - 4) col0 = Wp - 4;                       // past the allocation (last strip of the shifted copy): never sampled, any finite filler
    const float4 v = *reinterpret_cast<const float4*>(tiles + (size_t)slot * Hp * Wp + ((size_t)(row >> 2) * TW + (col0 >> 2)) * 16 + ((row & 3) << 2));
    *reinterpret_cast<float4*>(strips + (size_t)slot * 2 * phases * eds_strips_copy_elems(Hp, Wp) + (size_t)i * 4) = v;
}

bool eds_strips_prepare(eds_trk* h, int first, int count) {
    if (!h->tiled) return false;
    if (h->strips_unavailable) return false;       // they did not fit the budget when they were asked for: remembered (eds_trk_set_knob re-arms)
    if (!h->dstrips) {
        // row phases (eds_layout.hpp): 4 — one 128-byte line per patch — for handles that hold batches, whose solves are bound by the
        // fabric's line fills; 1 for the handles of the latency regime (a lone alignment is not bandwidth-bound, and its frame changes
        // with every call: 2.5 MB of copies to write instead of 10).  EDS_STRIPS_PHASES=1|2|4 overrides.
        // The copies are 8x the tiled frames with 4 phases (41 GB for 4 096 VGA slots, 124 GB at 1280x720): they may take at most
        // EDS_STRIPS_BUDGET_PCT (50) percent of the memory that is free NOW — fewer phases beyond that, none if even one does not fit —
        // so that a second handle, the team mailboxes or the caller's own buffers are not starved by a grab that happened to succeed.
        int phases = h->B >= 32 ? 4 : 1;
        if (h->knobs.strips_phases) phases = h->knobs.strips_phases;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        const size_t two_copies = (size_t)2 * eds_strips_copy_elems(h->Hp, h->Wp) * sizeof(float);
        phases = eds_strips_phases_for_budget(phases, (unsigned long long)h->B, two_copies, free_b, h->knobs.strips_budget_pct);
        for (; phases >= 1; phases >>= 1) {
            const size_t bytes = (size_t)h->B * phases * two_copies;
            if (hipMalloc((void**)&h->dstrips, bytes) == hipSuccess) { h->strips_bytes = bytes; break; }
            (void)hipGetLastError(); h->dstrips = nullptr;
        }
        if (!h->dstrips) { h->strips_unavailable = true; return false; }
        h->strip_phases = phases;
    }
    // the slots whose storage is sampled: a slot's own, or the one it shares (eds_trk_share_event_frame)
    std::vector<char> stale(h->B, 0);
    bool any = false;
    for (int s = first; s < first + count; ++s) {
        const int fs = h->slots[s].frame_slot >= 0 ? h->slots[s].frame_slot : s;
        Slot& src = h->slots[fs];
        if (src.strips_version != src.frame_version || src.strips_version == 0) { stale[fs] = 1; any = true; }
    }
    if (!any) return true;
    const int pieces = 2 * h->strip_phases * eds_strips_count(h->Wp) * h->Hp * 2;
    for (int s = 0; s < h->B;) {
        if (!stale[s]) { ++s; continue; }
        int e = s;
        while (e < h->B && stale[e] && e - s < 65535) ++e;
        hipLaunchKernelGGL(k_tiles_to_strips, dim3((pieces + 255) / 256, e - s), dim3(256), 0, h->st, h->dframe, h->dstrips, s, h->Hp, h->Wp, h->strip_phases);
        for (int k = s; k < e; ++k) {
            if (h->slots[k].frame_version == 0) h->slots[k].frame_version = 1;       // (frames written before versions were kept)
            h->slots[k].strips_version = h->slots[k].frame_version;
        }
        s = e;
    }
    return hipGetLastError() == hipSuccess;
}

bool eds_strips_current(const eds_trk* h, int first, int count) {
    if (!h->tiled || !h->dstrips) return false;
    for (int s = first; s < first + count; ++s) {
        const Slot& src = h->slots[h->slots[s].frame_slot >= 0 ? h->slots[s].frame_slot : s];
        if (src.strips_version == 0 || src.strips_version != src.frame_version) return false;
    }
    return true;
}

// Strips or tiles for this solve?  The copies cost what they weigh: 2 x phases x 1.3 MB written per 640x480 frame — 2.9 us of the
// whole GPU per frame with 4 row phases (bench.py frame_layout_prep), against 0.26 us that a 2 000-point solve gains from them
// (0.40 instead of 0.66 us per alignment at 4 096 per launch).  A frame that is solved ONCE — a live tracker: one event frame, one
// optimize — is therefore sampled from the tiles it was written in; a frame that is solved AGAIN (the second solve that finds the same
// frame version: batches that are re-solved, parameter sweeps, bench.py's steps over resident inputs) gets its copy then, and
// eds_trk_prepare_frames makes it up front.  A few new frames among many kept ones are converted at once (one launch of tiles for
// the whole range would cost more than their copies).  EDS_STRIPS_POLICY = reuse (default) | eager (convert at the first solve:
// round 3's first rule) | never.
bool eds_strips_for_solve(eds_trk* h, int first, int count) {
    const int policy = h->knobs.strips_policy;         // 0 reuse | 1 eager | 2 never (per handle: eds_launch_rule.hpp)
    if (!h->tiled) return false;
    int stale = 0, fresh = 0;
    for (int s = first; s < first + count; ++s) {
        Slot& src = h->slots[h->slots[s].frame_slot >= 0 ? h->slots[s].frame_slot : s];
        if (src.frame_version == 0) src.frame_version = 1;       // (frames written before versions were kept)
        const bool cur = h->dstrips && src.strips_version == src.frame_version;
        if (!cur) { ++stale; if (src.solved_version != src.frame_version) ++fresh; }
    }
    for (int s = first; s < first + count; ++s) {
        Slot& src = h->slots[h->slots[s].frame_slot >= 0 ? h->slots[s].frame_slot : s];
        src.solved_version = src.frame_version;
    }
    const int what = eds_strips_decide(policy, stale, fresh, count);           // (eds_layout.hpp; tested on the CPU: tests/test_host_logic.py)
    if (what == 0) return false;          // first solve on (most of) these frames: the tiles
    if (what == 1) return h->dstrips != nullptr;
    return eds_strips_prepare(h, first, count);
}

void eds_strips_free(eds_trk* h) {
    if (h->dstrips) hipFree(h->dstrips);
    h->dstrips = nullptr;
    h->strips_bytes = 0;
}
