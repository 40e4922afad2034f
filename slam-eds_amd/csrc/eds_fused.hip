// Persistent per-alignment solver kernel for gfx950 (EDS_EXEC_DEVICE).
//
// At N = 2 000 points one Gauss-Newton iteration moves ~0.2 MB and takes a few microseconds —
// two orders of magnitude below a launch + PCIe round trip — so the iteration loop itself has
// to live on the GPU.  One workgroup owns one alignment:
//
//   every lane   keeps its points' keyframe constants in REGISTERS for the whole solve (PPT points
//                per lane), projects them in fp32 through the small-displacement form of
//                eds_device.hpp, samples the frame (bicubic 4x4 / bilinear 2x2) from an LDS-resident
//                patch cache (HBM only when a point changes cell), forms r and the 1x6 SE(3) row in
//                registers and folds them straight into 28 running sums (J is never written)
//   wavefront    reduce-scatter butterfly (eds_device.hpp), LDS across the wavefronts
//   lane 0       runs the SAME edss::Solver6 state machine the host mode runs (eds_solver.hpp):
//                damped 6x6 Cholesky in fp64, exp(xi) T, accept / reject
//
// and loops until the solver reports done; the last pass stores the residuals at the accepted
// pose (what reference Tracker.cpp:223-230 writes to kf->residuals).  Independent alignments
// run on different CUs, B >> 256 fills the chip.  No MFMA-shaped work exists on this path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "eds_device.hpp"
#include "eds_fused.hpp"
#ifndef EDS_GATHER_STAGES
#define EDS_GATHER_STAGES 2      // groups the strip kernels consume their points in, each behind a counted wait (1: one wait for all rows)
#endif
#include "eds_handle.hpp"
#include "eds_launch_rule.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"
#include "eds_solver6_spec.hpp"

using namespace edsd;

#define EDS_FUSED_MAX_WAVES 16
#define EDS_CACHE_CAP 2048         // points whose frame patch stays resident in LDS across passes

// LDS-resident patch cache.  Every pass of an alignment samples the SAME frame at poses that differ
// by a fraction of a pixel, so the 4x4 (bicubic) / 2x2 (bilinear) neighbourhood of a point only
// changes when its integer cell does.  2 048 points x 16 taps x 4 B = 128 KB of the CU's 160 KB LDS
// hold every patch of the headline configuration; passes after the first then read the frame from
// LDS instead of re-gathering ~200 B of 64-B HBM sectors per point.  Tap-major layout
// ([tap][point]) keeps a wavefront's 64 lanes on 64 consecutive banks.  A point is always handled
// by the same lane, so the cache needs no synchronisation.
//
// PPT > 0: each lane owns points tid, tid + nthr, ... (PPT of them, N <= PPT * nthr) and keeps their
// constants in registers; PPT == 0: any N, constants re-read from HBM/L2 every pass.
// QUAD (bicubic, PPT > 0): the quad-cooperative gather of eds_device.hpp — lane j of a quad loads row j of each of the quad's
// four patches, the row splines run where the rows landed, a DPP transpose returns them to the point's own lane; the cache then
// holds [point][row] units of 16 bytes.  QUAD = 0 is the lane-per-point gather of round 1 (kept for bilinear and for A/B runs).
//
// TEAM = K > 1: K workgroups (K CUs) share ONE alignment — the latency regime, where a launch holds fewer alignments than the chip
// has CUs (the reference's own operating point is a single optimize per event slice, Tracker.cpp:104).  Member m takes the points
// [m, m + 1) * PPT * nthr; after the in-workgroup reduction every member publishes its 28 fp64 partial sums as 56 eight-byte
// {32 data bits, 32-bit tag} granules (one sc1 store each: a granule is written and read atomically, so data and "ready" flag
// cannot be seen apart and no fence is needed — MI355X_MICROARCH.md, granule hand-off), polls the other members' granules, and sums
// all K contributions in member order.  Every member therefore holds bit-identical totals and runs the (prepared-candidate) solver
// redundantly: no broadcast step, one exchange per pass (~1.5 us on an idle chip against a 4-9 us pass).  Teams are formed from an
// atomic ticket — members of a team hold consecutive tickets, so at most one team of a launch is ever incomplete and it is completed
// by the very next workgroups to start: no co-residency assumption, no deadlock whatever the dispatch order.  Polls are bounded
// (EDS_TEAM_TIMEOUT_TICKS of the 100 MHz clock); on a timeout the solve is reported failed-with-timeout and the host re-runs the
// range with TEAM = 1.
//
// GROUPS = G > 1 (round 5; TEAM > 1 only): SPECULATIVE CANDIDATE GROUPS.  One optimize per event slice (Tracker.cpp:104) leaves 250 of the
// 256 CUs idle, and the damped solver rejects more than half of its candidates on this problem (accept pattern 0000010111) — each
// rejection a full pass whose only product is one number.  The candidates that a run of rejections would walk through (lambda,
// 4 lambda, 16 lambda, ...) are all known once a linearisation exists (eds_solver6_spec.hpp prepares EDS_NSPEC of them side by side), so
// G teams of K CUs evaluate candidates k .. k + G - 1 AT THE SAME TIME: team g samples the frame at candidate k + g, all G x K
// workgroups exchange their 28 sums in one round of granules, and every workgroup replays Solver6::on_eval over the G results in
// order — reject, reject, ..., accept (everything behind the first accepted candidate is discarded).  Decisions, lambdas, trace
// records and iteration counts are those of the sequential solver bit for bit (each candidate's sums are added in the same member
// order as a team of K adds them); the pattern above takes 5 rounds instead of 11 passes.  The residuals at the accepted pose live in
// the registers of the group that evaluated it: that group writes them at the end.
#ifndef EDS_TEAM_P1_WAVES_PER_EU
#define EDS_TEAM_P1_WAVES_PER_EU 1
#endif
template <int SAMPLING, int PPT, int MAXT, int QUAD, int TEAM, int GROUPS = 1>
__global__ __launch_bounds__(MAXT, (TEAM > 1 && PPT == 1) ? EDS_TEAM_P1_WAVES_PER_EU : 1) void eds_fused6_kernel(EdsArrays A, const EdsFusedIn* __restrict__ in,
                                                          EdsFusedOut* __restrict__ out, edss::Solver6* __restrict__ sv_all,
                                                          int first, int iters, int damped, double lambda0,
                                                          double huber_tau, int nb, unsigned long long* __restrict__ mail,
                                                          int* __restrict__ ticket, unsigned ticket_base, unsigned epoch) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int s_ticket;
    static_assert(GROUPS == 1 || (TEAM > 1 && EDS_NSPEC % GROUPS == 0), "candidate groups: teams only, whole rounds of prepared candidates");
    constexpr int VTEAM = TEAM * GROUPS;              // workgroups per alignment
    __shared__ unsigned s_xchg[TEAM > 1 ? VTEAM : 1][EDS_TEAM_GRANULES];
    __shared__ int s_timeout;
    int team_slot = blockIdx.x, member = 0, group = 0;
    if (TEAM > 1) {
        if (tid == 0) { s_ticket = (int)((unsigned)atomicAdd(ticket, 1) - ticket_base); s_timeout = 0; }   // the counter is never reset: the host knows how many tickets earlier launches took
        __syncthreads();
        // a ticket outside this launch's stretch means host and device counters disagree (a launch that failed half-way): leave —
        // the host pre-marks every slot of a team launch as timed out, so the range is solved again and the counter reset
        if ((unsigned)s_ticket >= gridDim.x) return;
        team_slot = s_ticket / VTEAM; member = s_ticket % TEAM; group = (s_ticket % VTEAM) / TEAM;
    }
    const int slot = first + team_slot;
    if (tid == 0 && member == 0 && group == 0) out[slot].t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned pass_no = 0;                         // exchanges so far (TEAM > 1)
    int res_owner = 0;                            // GROUPS > 1: the group whose registers hold the residuals of the accepted pose
    __shared__ edss::Solver6 sv;
    __shared__ double s_pose[EDS_POSE_STRIDE];
    __shared__ float s_posef[16];      // fp32 copies for the point phase: [0..11] R - I and t of the pose to evaluate (serial solver), [12..13] fx, fy
    __shared__ float s_red[EDS_FUSED_MAX_WAVES][EDS_RED_K6];
    __shared__ float s_costp[2][EDS_FUSED_MAX_WAVES];   // per-wavefront cost of the pass in flight, double-buffered by pass parity (quick accept test)
    __shared__ edss::Sums6 s_sums;
    __shared__ int s_state;            // 0: iterate, 1: this pass is the final one, 2: done
    __shared__ int s_accept;           // the pass just consumed became the accepted pose (GROUPS > 1: 1 + the group whose candidate was
                                       // accepted, 0: none, -1: the first pass — every group evaluated the start pose)
    __shared__ edsp::SpecState sp;     // damped solver with register-resident points: prepared candidates (eds_solver6_spec.hpp)
    const bool spec_mode = (PPT > 0) && damped == 1 && iters > 0;   // (iters == 0: one residual pass, no solve) damped == 2: the serial solver lane of round 1 (A/B runs: EDS_LM6_SPEC=0)
    constexpr int NTAP = (SAMPLING == 0) ? 16 : 4;
    constexpr int NREG = PPT > 0 ? PPT : 1;
    constexpr bool CACHE = true;
    constexpr int NPATCH = (QUAD >= 3) ? 16 : NTAP;      // the strips' landing zone holds 4 rows x 4 pixels per point whatever the sampler reads of them
    // (the bilinear sampler on the strips reads two 16-byte units of OTHER lanes' rows per point: its landing slots are 1 KB + 16 bytes apart
    // — ZSH floats of slack behind every slot — so that the four lanes of a quad, which read the same units of four consecutive slots, do
    // not meet in one bank group: 16-way conflicts on every read before, profiles/r04_sq_counters.txt)
    constexpr int ZSH = (QUAD >= 3 && SAMPLING == 1) ? 4 : 0;
    constexpr int ZSLOT = 256 + ZSH;                     // floats from one landing slot (64 lanes x 16 bytes) to the next
    // (a team member of 512 points caches 512 patches: 32 KB instead of 128 — what lets two such workgroups share a CU)
    constexpr int CCAP = (TEAM > 1 && PPT == 1 && QUAD < 3) ? 512 : EDS_CACHE_CAP;
    __shared__ __attribute__((aligned(16))) float s_patch[CACHE ? NPATCH : 1][CACHE ? CCAP + 8 * ZSH : 1];
    static_assert(QUAD < 3 || (MAXT / 64) * NREG * 4 * ZSLOT <= NPATCH * (CCAP + 8 * ZSH), "landing zone exceeds its allocation");
    __shared__ int s_cell[CACHE ? CCAP : 1];

    const double* __restrict__ gpb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int Nall = (int)gpb[EDS_PB_N];
    const int ne = Nall / nb;
    // a team member sees its own slice of the points as "the" points: local index i <-> point poff + i
    const int poff = TEAM > 1 ? member * PPT * MAXT : 0;
    const int N = TEAM > 1 ? (Nall - poff < 0 ? 0 : (Nall - poff > PPT * MAXT ? PPT * MAXT : Nall - poff)) : Nall;
    const size_t base = (size_t)slot * A.Np + poff;
    static_assert(TEAM == 1 || PPT > 0, "teams keep their points in registers");
    const int fslot = __builtin_amdgcn_readfirstlane((int)gpb[EDS_PB_FRAME]);   // the slot whose frame storage is sampled (its own unless shared);
                                                                               // wave-uniform: the frame address then lives in SGPRs
    const FrameView frame = make_frame_view(A.frame, fslot, A.H, A.W, A.Hp, A.Wp, 1);    // persistent kernels: tiled frames only (eds_fused_solve)
    // start of that allocation (quad gather); fslot is wave-uniform, so this is scalar arithmetic and the address lives in SGPRs
    // (the product in a 32-bit scalar multiply: 64-bit it became VALU work whose result sat in VGPRs — two v_readfirstlane per load)
    const float* __restrict__ tiles = A.frame + (size_t)((unsigned)fslot * (unsigned)(A.Hp * A.Wp / 16)) * 16;
    static_assert(!QUAD || ((SAMPLING == 0 || QUAD >= 3) && PPT > 0 && PPT * MAXT <= CCAP), "quad gather: register-resident points, all cached; the bilinear sampler on strips only");

    if (tid == 0) {
        const EdsFusedIn& I = in[slot];
        for (int i = 0; i < 4; ++i) s_pose[EDS_PB_K + i] = gpb[EDS_PB_K + i];
        edsm::fill_pose_block(I.p, I.q, I.v, A.G + (size_t)slot * EDS_MAX_BLOCKS * 36, nb, s_pose);
        sv.init(damped != 0, iters, lambda0, I.p, I.q, PPT > 0 ? 1 : 0);   // PPT > 0: residuals of the accepted pose stay in registers
        s_state = sv.final_pass ? 1 : 0;
        for (int i = 0; i < 9; ++i) s_posef[i] = (float)s_pose[EDS_PB_D + i];
        for (int i = 0; i < 3; ++i) s_posef[9 + i] = (float)s_pose[EDS_PB_T + i];
        s_posef[12] = (float)s_pose[EDS_PB_K]; s_posef[13] = (float)s_pose[EDS_PB_K + 1];
        if (spec_mode) {                // candidate 0 of the first pass = the start pose
            edsp::Spec6& c0 = sp.spec[0];
            for (int i = 0; i < 3; ++i) { c0.p[i] = I.p[i]; c0.rt.t[i] = s_pose[EDS_PB_T + i]; }
            for (int i = 0; i < 4; ++i) c0.q[i] = I.q[i];
            for (int i = 0; i < 6; ++i) c0.xi[i] = 0.0;
            for (int i = 0; i < 9; ++i) c0.rt.D[i] = s_pose[EDS_PB_D + i];
            for (int i = 0; i < 9; ++i) c0.rt.f[i] = (float)c0.rt.D[i];
            for (int i = 0; i < 3; ++i) c0.rt.f[9 + i] = (float)c0.rt.t[i];
            c0.ok = 1;
            sp.k = 0; sp.mode = edsp::MODE_USE;
            for (int i = 0; i < EDS_RED_N6; ++i) sp.cur[i] = 0.0;
        }
    }
    for (int k = tid; k < EDS_FUSED_MAX_WAVES * EDS_RED_K6; k += nthr) (&s_red[0][0])[k] = 0.0f;   // rows of absent wavefronts stay 0
    if (tid < 2 * EDS_FUSED_MAX_WAVES) (&s_costp[0][0])[tid] = 0.0f;
    __syncthreads();

    // per-point constants: registers (PPT > 0) — loaded once, coalesced — and the normalised model
    // for the fixed velocity, mhat_i = a_i.v / n_block(i)
    PointKf kf[NREG];
    float kw[NREG], kmh[NREG];
    // the same constants as PAIRS of points for the packed point phase (quad gather, even PPT): element e of pair g = point 2 g + e
    constexpr bool PAIRS = QUAD != 0 && PPT > 0 && PPT % 2 == 0;
    constexpr int NPAIR = PAIRS ? NREG / 2 : 1;
    f2 k2x[NPAIR], k2y[NPAIR], k2rhop[NPAIR], k2f0x[NPAIR], k2f0y[NPAIR], k2w[NPAIR], k2mh[NPAIR];
    int kcell[NREG];
    {
        float vf[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) vf[k] = (float)s_pose[EDS_PB_V + k];
        const int reps = PPT > 0 ? PPT : (N + nthr - 1) / nthr;
        for (int j = 0; j < reps; ++j) {
            const int i = tid + j * nthr;
            const bool in_range = i < N;
            // (a lane without a point reads the SLOT's first point: `base` itself lies beyond the slot's planes for a team member whose
            // slice starts past the padded point count — 8 722 points on 16 members: slices from 9 216 on, planes end at 8 960)
            const size_t o = in_range ? base + i : (size_t)slot * A.Np;
            const float x = A.x[o], y = A.y[o], rho = A.rho[o];
            float a[6];
            model_row(x, y, rho, A.gx[o], A.gy[o], a);
            float m = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) m += a[k] * vf[k];
            const float mh = m * (float)s_pose[EDS_PB_BLK + EDS_PB_BLK_STRIDE * block_of(in_range ? poff + i : 0, ne, nb)];
            if (PPT > 0) {
#pragma unroll
                for (int jj = 0; jj < NREG; ++jj) {
                    if (jj == j) {          // j is a compile-time constant after unrolling
                        kf[jj].x = x; kf[jj].y = y; kf[jj].rhop = rho + 1e-5f;
                        kf[jj].f0x = A.f0x[o]; kf[jj].f0y = A.f0y[o]; kf[jj].cell0 = A.cell0[o];
                        kw[jj] = in_range ? A.w[o] : 0.0f;      // w = 0 silences out-of-range lanes
                        kmh[jj] = mh;
                        if (PAIRS) {
                            const int g = jj >> 1;
                            if (jj & 1) { k2x[g].y = x; k2y[g].y = y; k2rhop[g].y = rho + 1e-5f; k2f0x[g].y = kf[jj].f0x; k2f0y[g].y = kf[jj].f0y; k2w[g].y = kw[jj]; k2mh[g].y = mh; }
                            else { k2x[g].x = x; k2y[g].x = y; k2rhop[g].x = rho + 1e-5f; k2f0x[g].x = kf[jj].f0x; k2f0y[g].x = kf[jj].f0y; k2w[g].x = kw[jj]; k2mh[g].x = mh; }
                            kcell[jj] = kf[jj].cell0;
                        }
                    }
                }
            } else if (in_range) {
                A.mhat[o] = mh;
            }
            if (CACHE && i < CCAP) s_cell[i] = 0x7fffffff;       // no cell cached yet
        }
    }
    const float tau = (float)huber_tau;

#ifdef EDS_FUSED_STAMPS
    unsigned long long stamp_acc[3] = {0, 0, 0}, stamp_t = 0;
#define EDS_STAMP(k)                                                             \
    do {                                                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();            \
        if ((k) > 0) stamp_acc[(k) > 0 ? (k)-1 : 0] += now_ - stamp_t;           \
        stamp_t = now_;                                                          \
    } while (0)
#else
#define EDS_STAMP(k) do { } while (0)
#endif
    // second diagnostic set (-DEDS_FUSED_STAMPS=2): the reduction split into [wave butterfly + LDS write | barrier wait | cross-wave sum]
#if defined(EDS_FUSED_STAMPS) && EDS_FUSED_STAMPS == 2
    unsigned long long stamp2_acc[3] = {0, 0, 0}, stamp2_t = 0;
#define EDS_STAMP2(k)                                                            \
    do {                                                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();            \
        if ((k) > 1) stamp2_acc[(k)-1] += now_ - stamp2_t;                       \
        if ((k) == 1) stamp2_acc[0] += now_ - stamp_t;                           \
        stamp2_t = now_;                                                         \
    } while (0)
#else
#define EDS_STAMP2(k) do { } while (0)
#endif
#if defined(EDS_FUSED_STAMPS) && EDS_FUSED_STAMPS == 3
    __shared__ unsigned s_miss_total, s_pass_total;      // diagnostic build 3: patches gathered / passes run (cache hit rate)
    if (tid == 0) { s_miss_total = 0; s_pass_total = 0; }
#define EDS_COUNT_MISS(m) do { const unsigned long long bal_ = __ballot(m); if (lane == 0) atomicAdd(&s_miss_total, (unsigned)__popcll(bal_)); } while (0)
#else
#define EDS_COUNT_MISS(m) do { } while (0)
#endif
    // One CU per alignment, prepared candidates: what decides a pass is its COST alone, and LM rejects more than half of its
    // candidates — a rejected pass needs neither the other 27 sums nor the solver.  So the cost is reduced first (one value per
    // wavefront, one barrier), every wavefront takes the accept test itself, and after a rejection all of them walk on to the next
    // prepared candidate at once: no 28-value reduction, no second barrier, no serial section.  The control state each wavefront
    // needs for that lives in its registers (k_r: candidate under evaluation, iter_r, have_cur_r, cur_cost_r), reloaded from LDS after
    // every pass that did go through the solver lane.  The cost partials are double-buffered by pass parity (a wavefront can run at
    // most one barrier ahead of the slowest reader).
    constexpr bool QUICK = TEAM == 1 && PPT > 0;
    int k_r = 0, iter_r = 0, have_cur_r = 0;
    double cur_cost_r = 0.0;
    unsigned parity = 0;
    float rcand[NREG], racc[NREG];      // residuals of the pass in flight / of the accepted pose (PPT > 0)
#pragma unroll
    for (int j = 0; j < NREG; ++j) { rcand[j] = 0.0f; racc[j] = 0.0f; }
    for (;;) {
        EDS_STAMP(0);
        const int state = s_state;
        // the prepared candidate this pass evaluates (candidate groups: group g takes candidate sp.k + g; the first pass has one candidate, the start pose)
        const int kcur = (QUICK && spec_mode) ? k_r : (GROUPS > 1 ? sp.k + (sv.have_cur ? group : 0) : sp.k);
        PoseF ps;
        if (PAIRS) { }
        else if (spec_mode) load_pose_rt(sp.spec[kcur].rt.D, sp.spec[kcur].rt.t, s_pose, ps);
        else load_pose(s_pose, ps);
        float acc[EDS_RED_K6];
#pragma unroll
        for (int j = 0; j < EDS_RED_K6; ++j) acc[j] = 0.0f;
        // Consumes one point: sample from its (register-resident) taps, residual, 1x6 row, running sums.
        auto consume = [&](const PointGeom& pg, float (&tap)[NTAP], float w, float mhat, int i) -> float {
            const float r = point_row6<SAMPLING, NTAP>(ps, pg, tap, w, mhat, tau, acc);
            if (PPT == 0 && state == 1 && i < N) A.r[base + i] = r;      // streaming variant: residuals stored by a final pass
            return r;
        };
        if constexpr (QUAD >= 3) {
            // ---- round 3, second step: the gather on STRIPS, landing in LDS (eds_layout.hpp; tools/ubench_gather_lds.hip) -------------
            // Every patch row is ONE 16-byte read at a 4-byte-aligned address, issued as global_load_lds_dwordx4: lane L's 16 bytes go
            // straight to  zone[wave][4 j + q][L]  in LDS — no VGPRs while in flight, no barrel shift, and the landing zone IS the patch
            // cache: a point whose cell did not change simply does not issue its loads (its rows of the previous pass are still there).
            // The owner of a point computes the byte offset of its patch's first row once; it goes round the quad by DPP with the
            // "gather it" flag in bit 31, and lane j adds j rows (and clears the flag) with one add.
            typedef __attribute__((address_space(3))) void* lds_ptr;
            typedef const __attribute__((address_space(1))) void* glb_ptr;
            const int jr = lane & 3;
            const float* __restrict__ rtf = spec_mode ? sp.spec[kcur].rt.f : s_posef;
            EDS_LOAD_POSE_SCALARS(rtf, s_posef + 12);
            PairGeom pg[NPAIR];
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            float* __restrict__ zone = &s_patch[0][0] + wave_u * (NREG * 4 * ZSLOT);
            const unsigned copy_bytes = (unsigned)(eds_strips_copy_elems(A.Hp, A.Wp) * 4);
            const char* __restrict__ sbase = reinterpret_cast<const char*>(A.strips) + (size_t)(unsigned)fslot * ((size_t)(2 * A.strip_phases) * copy_bytes);
            const unsigned row_add = 0x80000000u + 32u * (unsigned)jr;
            int nld[NREG];                               // row-load instructions this wavefront really issued for point j (wave-uniform)
#pragma unroll
            for (int g = 0; g < NPAIR; ++g) {
                int r0[2], c0[2];
                project_pair(EDS_POSE_SCALARS, k2x[g], k2y[g], k2rhop[g], k2f0x[g], k2f0y[g], kcell[2 * g], kcell[2 * g + 1], pg[g], r0, c0);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * g + e, i = tid + j * nthr;
                    nld[j] = 0;
                    const int key = (r0[e] << 16) ^ (c0[e] & 0xffff);
                    const bool miss = s_cell[i] != key;
                    EDS_COUNT_MISS(miss && i < N);
                    if (miss) s_cell[i] = key;
                    const int ra = clampi(r0[e], -2, frame.H) + (EDS_FRAME_MARGIN - 1), ca = clampi(c0[e], -2, frame.W) + (EDS_FRAME_MARGIN - 1);
                    const int off = (int)(eds_strips_row_offset(ra, ca, A.Hp, copy_bytes, A.strip_phases) | (miss ? 0x80000000u : 0u));
                    const int o0 = quad_bcast_i<0>(off), o1 = quad_bcast_i<1>(off), o2 = quad_bcast_i<2>(off), o3 = quad_bcast_i<3>(off);
                    const int oq[4] = {o0, o1, o2, o3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool need = oq[q] < 0 && (SAMPLING == 0 || jr == 1 || jr == 2);      // (the bilinear sampler reads rows 1 and 2 of the patch only)
                        if (need)
                            __builtin_amdgcn_global_load_lds((glb_ptr)(sbase + ((unsigned)oq[q] + row_add)), (lds_ptr)(zone + (4 * j + q) * ZSLOT), 16, 0, 0);
                        if (EDS_GATHER_STAGES > 1) nld[j] += __ballot(need) != 0ull ? 1 : 0;
                    }
                    // The counted waits below rely on the row loads being ISSUED in point order (loads return in order: "at most n
                    // outstanding" then means the earlier points' rows have landed).  Today that order follows from the M0 dependency
                    // chain of global_load_lds; this barrier pins it in the source as well — no load of point j + 1 may be
                    // moved in front of a load of point j.  tests/test_gather_stages_gpu.py compares against an EDS_GATHER_STAGES=1 build.
                    asm volatile("" ::: "memory");
                }
            }
            // Loads return in order, so "at most n outstanding" with n = the loads issued for LATER points means this point's rows have
            // landed: the points are consumed in EDS_GATHER_STAGES groups, each behind its own counted wait, while the rows of the groups
            // behind it are still on their way.  (n rounded down to the immediates below: a stricter wait, never a laxer one.)
            auto wait_rows_but = [](int later) {
                if (later >= 12) __builtin_amdgcn_s_waitcnt(0x0F7C);
                else if (later >= 8) __builtin_amdgcn_s_waitcnt(0x0F78);
                else if (later >= 4) __builtin_amdgcn_s_waitcnt(0x0F74);
                else __builtin_amdgcn_s_waitcnt(0x0F70);
                asm volatile("" ::: "memory");
            };
            if (EDS_GATHER_STAGES <= 1) wait_rows_but(0);        // vmcnt(0): every row has landed (each lane reads back only what it fetched itself)
            // phase B: the rows of two patches per packed instruction straight out of LDS (ds_read2st64_b32 pairs {q, q + 1}), the
            // transposes, the column spline of {value, column derivative}, the row and its 28 products
            Acc6 A6;
            A6.clear();
            const float* __restrict__ mine = zone + 4 * lane;
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const int g = j >> 1;
                if (EDS_GATHER_STAGES > 1 && j % (NREG / (EDS_GATHER_STAGES < NREG ? EDS_GATHER_STAGES : NREG)) == 0) {
                    constexpr int per = NREG / (EDS_GATHER_STAGES < NREG ? EDS_GATHER_STAGES : NREG);
                    int later = 0;
#pragma unroll
                    for (int jj = 0; jj < NREG; ++jj)
                        if (jj >= j + per) later += nld[jj];
                    // (the group before this wait must really be computed before it: without the tie the scheduler sinks its arithmetic below
                    // the wait and reads every group's rows up front)
                    if (j > 0) asm volatile("" : "+v"(rcand[j - 1]) :: "memory");
                    wait_rows_but(later);
                }
                const float ax_j = (j & 1) ? pg[g].ax.y : pg[g].ax.x, ay_j = (j & 1) ? pg[g].ay.y : pg[g].ay.x;
                if constexpr (SAMPLING == 1) {
                    // bilinear: the lane's own point needs pixels 1-2 of rows 1-2 of its patch — the rows quad lanes 1 and 2 fetched
                    // (the wavefront's counted wait covers every lane's loads of this point): two 16-byte LDS reads, no row pass, no transposes
                    const float* __restrict__ r1 = zone + (4 * j + (lane & 3)) * ZSLOT + 4 * ((lane & ~3) + 1);
                    const float4 u1 = *reinterpret_cast<const float4*>(r1), u2 = *reinterpret_cast<const float4*>(r1 + 4);      // rows 1 and 2 as whole 16-byte units
                    const float p4[4] = {u1.y, u1.z, u2.y, u2.z};              // (row 2 sits one lane — 4 floats — further)
                    float E, Er, Ec;
                    bilinear_patch(p4, ay_j, ax_j, E, Er, Ec);
                    const float iz_b = (j & 1) ? pg[g].iz.y : pg[g].iz.x, un_b = (j & 1) ? pg[g].un.y : pg[g].un.x, vn_b = (j & 1) ? pg[g].vn.y : pg[g].vn.x;
                    const float w_b = (j & 1) ? k2w[g].y : k2w[g].x, mh_b = (j & 1) ? k2mh[g].y : k2mh[g].x;
                    rcand[j] = row6_accumulate<(QUAD == 4)>(ps_fx, ps_fy, iz_b, un_b, vn_b, E, Er, Ec, w_b, mh_b, tau, A6);
                    continue;
                }
                const f2 x01 = {quad_bcast_f<0>(ax_j), quad_bcast_f<1>(ax_j)}, x23 = {quad_bcast_f<2>(ax_j), quad_bcast_f<3>(ax_j)};
                f2 ta[4], tb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ta[k] = (f2){mine[(4 * j + 0) * ZSLOT + k], mine[(4 * j + 1) * ZSLOT + k]};
                    tb[k] = (f2){mine[(4 * j + 2) * ZSLOT + k], mine[(4 * j + 3) * ZSLOT + k]};
                }
                f2 f01, d01, f23, d23;
                hermite_pair(ta[0], ta[1], ta[2], ta[3], x01, 0.5f * x01, 3.0f * x01, f01, d01);
                hermite_pair(tb[0], tb[1], tb[2], tb[3], x23, 0.5f * x23, 3.0f * x23, f23, d23);
                float f[4] = {f01.x, f01.y, f23.x, f23.y}, d[4] = {d01.x, d01.y, d23.x, d23.y};
                quad_transpose(f, lane);
                quad_transpose(d, lane);
                f2 EEc, dE;
                const f2 y2 = (f2)(ay_j);
                hermite_pair((f2){f[0], d[0]}, (f2){f[1], d[1]}, (f2){f[2], d[2]}, (f2){f[3], d[3]}, y2, 0.5f * y2, 3.0f * y2, EEc, dE);
                const float iz_j = (j & 1) ? pg[g].iz.y : pg[g].iz.x, un_j = (j & 1) ? pg[g].un.y : pg[g].un.x, vn_j = (j & 1) ? pg[g].vn.y : pg[g].vn.x;
                const float w_j = (j & 1) ? k2w[g].y : k2w[g].x, mh_j = (j & 1) ? k2mh[g].y : k2mh[g].x;
                rcand[j] = row6_accumulate<(QUAD == 4)>(ps_fx, ps_fy, iz_j, un_j, vn_j, EEc.x, dE.x, EEc.y, w_j, mh_j, tau, A6);
            }
            A6.unpack(acc);
        } else if constexpr (QUAD != 0 && PPT > 0 && PPT % 2 == 0) {
            // ---- round 3: the quad-cooperative gather on pairs (eds_device.hpp, "instruction diet") --------------------------------
            // phase A: the lane's points go through the projection two at a time (packed fp32); every point probes the cache and
            // its packed origin goes round the quad: a MISSING patch has lane j put row j in flight from HBM, a CACHED one has lane j
            // read its row of the cache into the SAME registers (origin 0: shift amount 0, so phase B's barrel shift passes it through
            // unchanged — no select between "gathered" and "cached" any more)
            const int jr = lane & 3;
            float* __restrict__ cache = &s_patch[0][0] + 16 * (tid & ~3) + 4 * jr;
            const int swz = (tid >> 2) & 3;
            const int cq0 = 16 * (0 ^ swz), cq1 = 16 * (1 ^ swz), cq2 = 16 * (2 ^ swz), cq3 = 16 * (3 ^ swz);
            const int cq[4] = {cq0, cq1, cq2, cq3};
            const float* __restrict__ rtf = spec_mode ? sp.spec[kcur].rt.f : s_posef;
            EDS_LOAD_POSE_SCALARS(rtf, s_posef + 12);
            PairGeom pg[NPAIR];
            int org[NREG];
            float4 ra[NREG][4], rb[NREG][4];
#pragma unroll
            for (int g = 0; g < NPAIR; ++g) {
                int r0[2], c0[2];
                project_pair(EDS_POSE_SCALARS, k2x[g], k2y[g], k2rhop[g], k2f0x[g], k2f0y[g], kcell[2 * g], kcell[2 * g + 1], pg[g], r0, c0);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * g + e, i = tid + j * nthr;
                    const int key = (r0[e] << 16) ^ (c0[e] & 0xffff);
                    const bool miss = s_cell[i] != key;
                    if (miss) s_cell[i] = key;
                    org[j] = pack_origin(frame, r0[e], c0[e]) | (miss ? (int)0x80000000 : 0);
                    const int o0 = quad_bcast_i<0>(org[j]), o1 = quad_bcast_i<1>(org[j]), o2 = quad_bcast_i<2>(org[j]), o3 = quad_bcast_i<3>(org[j]);
                    const int oq[4] = {o0, o1, o2, o3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ra[j][q] = dont_care4(); rb[j][q] = dont_care4();
                        if (oq[q] < 0) load_patch_row(tiles, frame.TW, oq[q] & 0x7fffffff, jr, ra[j][q], rb[j][q]);
                    }
                }
            }
            // phase B (branch-free): barrel shift, the row back to the cache, the row splines of two patches per packed instruction,
            // the transposes, the column spline of {value, column derivative} as one pair, the row and its 28 products
            Acc6 A6;
            A6.clear();
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const int g = j >> 1;
                const float ax_j = (j & 1) ? pg[g].ax.y : pg[g].ax.x, ay_j = (j & 1) ? pg[g].ay.y : pg[g].ay.x;
                const int o0 = quad_bcast_i<0>(org[j]), o1 = quad_bcast_i<1>(org[j]), o2 = quad_bcast_i<2>(org[j]), o3 = quad_bcast_i<3>(org[j]);
                const int oq[4] = {o0, o1, o2, o3};
                const f2 x01 = {quad_bcast_f<0>(ax_j), quad_bcast_f<1>(ax_j)}, x23 = {quad_bcast_f<2>(ax_j), quad_bcast_f<3>(ax_j)};
                float t[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 c = *reinterpret_cast<const float4*>(cache + cq[q] + 16 * j * nthr);
                    shift_patch_row3(ra[j][q], rb[j][q], oq[q], t[q]);
                    const bool m = oq[q] < 0;                    // gathered this pass
                    t[q][0] = flag_select(m, t[q][0], c.x); t[q][1] = flag_select(m, t[q][1], c.y); t[q][2] = flag_select(m, t[q][2], c.z); t[q][3] = flag_select(m, t[q][3], c.w);
                    *reinterpret_cast<float4*>(cache + cq[q] + 16 * j * nthr) = make_float4(t[q][0], t[q][1], t[q][2], t[q][3]);
                }
                f2 f01, d01, f23, d23;
                hermite_pair((f2){t[0][0], t[1][0]}, (f2){t[0][1], t[1][1]}, (f2){t[0][2], t[1][2]}, (f2){t[0][3], t[1][3]}, x01, 0.5f * x01, 3.0f * x01, f01, d01);
                hermite_pair((f2){t[2][0], t[3][0]}, (f2){t[2][1], t[3][1]}, (f2){t[2][2], t[3][2]}, (f2){t[2][3], t[3][3]}, x23, 0.5f * x23, 3.0f * x23, f23, d23);
                float f[4] = {f01.x, f01.y, f23.x, f23.y}, d[4] = {d01.x, d01.y, d23.x, d23.y};
                quad_transpose(f, lane);
                quad_transpose(d, lane);
                f2 EEc, dE;
                const f2 y2 = (f2)(ay_j);
                hermite_pair((f2){f[0], d[0]}, (f2){f[1], d[1]}, (f2){f[2], d[2]}, (f2){f[3], d[3]}, y2, 0.5f * y2, 3.0f * y2, EEc, dE);
                const float iz_j = (j & 1) ? pg[g].iz.y : pg[g].iz.x, un_j = (j & 1) ? pg[g].un.y : pg[g].un.x, vn_j = (j & 1) ? pg[g].vn.y : pg[g].vn.x;
                const float w_j = (j & 1) ? k2w[g].y : k2w[g].x, mh_j = (j & 1) ? k2mh[g].y : k2mh[g].x;
                rcand[j] = row6_accumulate<(QUAD == 2)>(ps_fx, ps_fy, iz_j, un_j, vn_j, EEc.x, dE.x, EEc.y, w_j, mh_j, tau, A6);
            }
            A6.unpack(acc);
        } else if (QUAD) {
            // phase A: every lane projects its own points and probes the cache; the packed origins go round the quad and every
            // lane puts its ROW of each missing patch in flight
            const int jr = lane & 3;
            // this lane's four cache units per point slot: unit = 4 * point + row, XOR-swizzled by the quad index (patch_unit); the
            // quad index of slot j is (tid >> 2) + j * nthr / 4 and nthr / 4 is a multiple of 4, so the swizzle is per lane, and the
            // slot enters as a compile-time offset
            float* __restrict__ cache = &s_patch[0][0] + 16 * (tid & ~3) + 4 * jr;
            const int swz = (tid >> 2) & 3;
            const int cq0 = 16 * (0 ^ swz), cq1 = 16 * (1 ^ swz), cq2 = 16 * (2 ^ swz), cq3 = 16 * (3 ^ swz);
            const int cq[4] = {cq0, cq1, cq2, cq3};
            PointGeom pg[NREG];
            int org[NREG];
            float4 ra[NREG][4], rb[NREG][4];
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const int i = tid + j * nthr;
                project_point(ps, kf[j], pg[j]);
                const int key = (pg[j].r0 << 16) ^ (pg[j].c0 & 0xffff);
                const bool miss = s_cell[i] != key;
                if (miss) s_cell[i] = key;
                org[j] = pack_origin(frame, pg[j].r0, pg[j].c0) | (miss ? (int)0x80000000 : 0);
                const int o0 = quad_bcast_i<0>(org[j]), o1 = quad_bcast_i<1>(org[j]), o2 = quad_bcast_i<2>(org[j]), o3 = quad_bcast_i<3>(org[j]);
                const int oq[4] = {o0, o1, o2, o3};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ra[j][q] = dont_care4(); rb[j][q] = dont_care4();
                    // (tried: cached patches as origin 0 and the loads unconditional — 7 % fewer instructions, 1 % slower here,
                    // 3 % faster in eds_fused12_kernel, which does it that way)
                    if (oq[q] < 0) load_patch_row(tiles, frame.TW, oq[q] & 0x7fffffff, jr, ra[j][q], rb[j][q]);
                }
            }
            // phase B (branch-free): the cached row and the gathered row are both formed, a bit mask picks one; the row goes (back)
            // to the cache, its spline runs here, the transposes return the four row results to the lane that owns the point
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const int o0 = quad_bcast_i<0>(org[j]), o1 = quad_bcast_i<1>(org[j]), o2 = quad_bcast_i<2>(org[j]), o3 = quad_bcast_i<3>(org[j]);
                const int oq[4] = {o0, o1, o2, o3};
                const float x0 = quad_bcast_f<0>(pg[j].ax), x1 = quad_bcast_f<1>(pg[j].ax), x2 = quad_bcast_f<2>(pg[j].ax), x3 = quad_bcast_f<3>(pg[j].ax);
                const float xq[4] = {x0, x1, x2, x3};
                float4 c[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) c[q] = *reinterpret_cast<const float4*>(cache + cq[q] + 16 * j * nthr);
                float f[4], d[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t[4];
                    shift_patch_row(ra[j][q], rb[j][q], oq[q], t);
                    const int m = oq[q] >> 31;                   // all ones: gathered this pass
                    t[0] = flag_select(m != 0, t[0], c[q].x); t[1] = flag_select(m != 0, t[1], c[q].y); t[2] = flag_select(m != 0, t[2], c[q].z); t[3] = flag_select(m != 0, t[3], c[q].w);
                    *reinterpret_cast<float4*>(cache + cq[q] + 16 * j * nthr) = make_float4(t[0], t[1], t[2], t[3]);
                    hermite(t[0], t[1], t[2], t[3], xq[q], f[q], d[q]);
                }
                quad_transpose(f, lane);
                quad_transpose(d, lane);
                float E, Er, Ec, unused;
                hermite(f[0], f[1], f[2], f[3], pg[j].ay, E, Er);
                hermite(d[0], d[1], d[2], d[3], pg[j].ay, Ec, unused);
                rcand[j] = point_row6_sampled(ps, pg[j], E, Er, Ec, kw[j], kmh[j], tau, acc);
            }
        } else if (PPT > 0) {
            // phase A: project every point of this lane, probe the patch cache, and put ALL the
            // missing gathers in flight before anything waits on one (memory-level parallelism:
            // a miss costs a ~2 us HBM round trip, paid once per pass instead of once per point)
            PointGeom pg[NREG];
            float tap[NREG][NTAP];
            bool miss[NREG];
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const int i = tid + j * nthr;
                project_point(ps, kf[j], pg[j]);
                const bool cached = CACHE && i < CCAP;
                const int key = (pg[j].r0 << 16) ^ (pg[j].c0 & 0xffff);
                miss[j] = !(cached && s_cell[i] == key);
                if (miss[j]) {
                    if (SAMPLING == 0) load_patch16(frame, pg[j].r0, pg[j].c0, reinterpret_cast<float(&)[16]>(tap[j]));
                    else load_patch4(frame, pg[j].r0, pg[j].c0, reinterpret_cast<float(&)[4]>(tap[j]));
                    if (cached) s_cell[i] = key;
                } else {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) tap[j][t] = s_patch[t][i];
                }
            }
            // phase B: refill the cache lines that missed, then consume
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const int i = tid + j * nthr;
                if (CACHE && miss[j] && i < CCAP) {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) s_patch[t][i] = tap[j][t];
                }
                rcand[j] = consume(pg[j], tap[j], kw[j], kmh[j], i);
            }
        } else {
            for (int i = tid; i < N; i += nthr) {
                const size_t o = base + i;
                PointKf k;
                k.x = A.x[o]; k.y = A.y[o]; k.rhop = A.rho[o] + 1e-5f; k.f0x = A.f0x[o]; k.f0y = A.f0y[o]; k.cell0 = A.cell0[o];
                PointGeom pg;
                project_point(ps, k, pg);
                float tap[NTAP];
                const bool cached = CACHE && i < CCAP;
                const int key = (pg.r0 << 16) ^ (pg.c0 & 0xffff);
                if (cached && s_cell[i] == key) {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) tap[t] = s_patch[t][i];
                } else {
                    if (SAMPLING == 0) load_patch16(frame, pg.r0, pg.c0, reinterpret_cast<float(&)[16]>(tap));
                    else load_patch4(frame, pg.r0, pg.c0, reinterpret_cast<float(&)[4]>(tap));
                    if (cached) {
#pragma unroll
                        for (int t = 0; t < NTAP; ++t) s_patch[t][i] = tap[t];
                        s_cell[i] = key;
                    }
                }
                consume(pg, tap, A.w[o], A.mhat[o], i);
            }
        }
        EDS_STAMP(1);
#if defined(EDS_FUSED_STAMPS) && EDS_FUSED_STAMPS == 3
        if (tid == 0) ++s_pass_total;
#endif
        double quick_cost = 0.0;
        if (QUICK && spec_mode) {
            // the pass's cost: in-row DPP sums, the four row totals by v_readlane, one float per wavefront, fp64 across the wavefronts
            float c = acc[EDS_RED_N6 - 1];
            c += dpp_f<0xB1>(c);                 // quad_perm [1,0,3,2]
            c += dpp_f<0x4E>(c);                 // quad_perm [2,3,0,1]
            c += dpp_f<0x141>(c);                // row_half_mirror
            c += dpp_f<0x140>(c);                // row_mirror
            const float wc = ((__int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 16))) +
                              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 32))) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 48));
            if (lane == 0) s_costp[parity][wave] = wc;
            __syncthreads();
#pragma unroll
            for (int wv = 0; wv < EDS_FUSED_MAX_WAVES; ++wv) quick_cost += (double)s_costp[parity][wv];
            parity ^= 1u;
            const bool reject = have_cur_r && !(quick_cost < cur_cost_r);
            const bool last = iter_r + 1 >= sv.max_iters;
            if (reject && (last || k_r + 1 < EDS_NSPEC)) {
                // Solver6::on_eval for a rejected candidate, without the solver lane: trace, counters, lambda (kept by lane 0 of
                // wavefront 0 for the end of the solve and for the next proposal), then the candidate prepared for exactly this lambda
                if (tid == 0) {
                    const int nt = sv.ntrace;
                    if (nt < EDS_MAX_TRACE) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) sv.tr_xi[nt][i] = sp.spec[k_r].xi[i];
                        sv.tr_cost[nt] = quick_cost; sv.tr_acc[nt] = 0;
                        sv.ntrace = nt + 1;
                    }
                    sv.iter = iter_r + 1; sv.last_accepted = 0;
                    sv.lambda = edsp::next_lambda_after_reject(sv.lambda);
                    sp.k = last ? 0 : k_r + 1; sp.mode = last ? edsp::MODE_DONE : edsp::MODE_USE;
                    if (last) { sv.final_cost = cur_cost_r; sv.done = 1; s_state = 2; }
                }
                ++iter_r;
                if (last) break;
                ++k_r;
                if (!sp.spec[k_r].ok) {             // damped matrix not positive definite: Solver6 gives up here (iteration >= 1: not a failure)
                    if (tid == 0) { sv.failed = 0; sv.final_cost = cur_cost_r; sv.done = 1; }
                    break;
                }
                continue;
            }
        }
        wave_reduce_scatter<EDS_RED_K6>(acc, lane);
        if (lane < 32) s_red[wave][wave_red_index<EDS_RED_K6>(lane, 0)] = acc[0];
        EDS_STAMP2(1);
        __syncthreads();
        EDS_STAMP2(2);
        if (spec_mode) {
            // ---- damped solver with prepared candidates (eds_solver6_spec.hpp); decisions as edss::Solver6::on_eval takes them ----
            if (TEAM > 1) {
                // Exchange of the partial sums.  Wavefront 0 adds the workgroup's eight partials and publishes them: lane t < 28
                // writes its double as two tagged 8-byte granules (sc1 stores).  Wavefront w gathers the granules of members
                // w, w + 8, ... (own member included: every member then adds the same numbers in the same order) — the members sit
                // on different XCDs (consecutive workgroup ids), so a poll is a fabric round trip, and K of them in a row on one
                // wavefront were most of the exchange.
                const unsigned tag = (epoch << 8) | ((pass_no & 0x7f) + 1);
                unsigned long long* mb = mail + ((size_t)team_slot * 2 + (pass_no & 1)) * (VTEAM * EDS_TEAM_GRANULES);
                const int me = group * TEAM + member;
                if (wave == 0 && lane < EDS_RED_N6) {
                    double s = 0.0;
#pragma unroll
                    for (int wv = 0; wv < EDS_FUSED_MAX_WAVES; ++wv) s += (double)s_red[wv][lane];
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
                    __hip_atomic_store(mb + me * EDS_TEAM_GRANULES + 2 * lane, ((unsigned long long)tag << 32) | (bits & 0xffffffffull),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(mb + me * EDS_TEAM_GRANULES + 2 * lane + 1, ((unsigned long long)tag << 32) | (bits >> 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                if constexpr (GROUPS > 1) {
                    // G x K members: a wavefront asks for ALL its members' granules at once (one fabric round trip for the lot), then
                    // asks again only for the late ones
                    constexpr int NW = MAXT / 64, PER = (VTEAM + NW - 1) / NW;
                    if (lane < 2 * EDS_RED_N6) {
                        unsigned long long v[PER];
#pragma unroll
                        for (int i = 0; i < PER; ++i) {
                            const int m = wave + i * NW;
                            v[i] = m < VTEAM ? __hip_atomic_load(mb + m * EDS_TEAM_GRANULES + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
                        }
                        for (;;) {
                            bool late = false;
#pragma unroll
                            for (int i = 0; i < PER; ++i) late |= (unsigned)(v[i] >> 32) != tag;
                            if (!late) break;
                            if (__builtin_amdgcn_s_memrealtime() - t_start > EDS_TEAM_TIMEOUT_TICKS) { s_timeout = 1; break; }
                            __builtin_amdgcn_s_sleep(1);
                            for (int i = 0; i < PER; ++i)           // (constant trip count: unrolled without being asked; with the pragma PER = 1 draws a warning)
                                if ((unsigned)(v[i] >> 32) != tag)
                                    v[i] = __hip_atomic_load(mb + (wave + i * NW) * EDS_TEAM_GRANULES + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                        for (int i = 0; i < PER; ++i)
                            if (wave + i * NW < VTEAM) s_xchg[wave + i * NW][lane] = (unsigned)v[i];
                    }
                } else {
                for (int m = wave; m < TEAM; m += nthr >> 6) {
                    if (lane < 2 * EDS_RED_N6) {
                        const unsigned long long* g = mb + m * EDS_TEAM_GRANULES + lane;
                        unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        while ((unsigned)(v >> 32) != tag) {
                            if (__builtin_amdgcn_s_memrealtime() - t_start > EDS_TEAM_TIMEOUT_TICKS) { s_timeout = 1; break; }
                            __builtin_amdgcn_s_sleep(2);
                            v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        s_xchg[m][lane] = (unsigned)v;
                    }
                }
                }
                ++pass_no;
                __syncthreads();
            }
            if constexpr (GROUPS > 1) {
              if (wave == 0) {
                // ---- candidate groups: Solver6::on_eval replayed over the G candidates of this round, in order --------------------------
                EDS_STAMP(2);
                EDS_STAMP2(3);
                const int k = sp.k, max_iters = sv.max_iters;
                int have_cur = sv.have_cur, iter = sv.iter, ntrace = sv.ntrace;
                double lambda = sv.lambda;
                double cur_cost = sp.cur[EDS_RED_N6 - 1];
                int mode = edsp::MODE_SOLVE, nk = 0, acc = 0, last_ok = 1;
                bool decided = false;
                for (int g = 0; g < GROUPS; ++g) {
                    if (decided) break;
                    double s = 0.0;
                    if (lane < EDS_RED_N6) {
#pragma unroll
                        for (int m = 0; m < TEAM; ++m)
                            s += __longlong_as_double((long long)(((unsigned long long)s_xchg[g * TEAM + m][2 * lane + 1] << 32) | s_xchg[g * TEAM + m][2 * lane]));
                    }
                    const long long sb = __double_as_longlong(s);
                    const double cost = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(sb >> 32), EDS_RED_N6 - 1) << 32) |
                                                             (unsigned int)__builtin_amdgcn_readlane((int)(sb & 0xffffffffll), EDS_RED_N6 - 1));
                    const bool bad = lane < EDS_RED_N6 && !(fabs(s) < 1e300);
                    const bool fin = __ballot(bad) == 0ull;
                    if (!have_cur) {                    // first round: every group evaluated the start pose; group 0's sums are the linearisation
                        decided = true;
                        if (!fin) { mode = edsp::MODE_DONE; if (lane == 0) { sv.failed = 1; sv.done = 1; } }
                        else { have_cur = 1; acc = -1; if (lane < EDS_RED_N6) sp.cur[lane] = s; if (lane == 0) sv.initial_cost = cost; }
                        break;
                    }
                    const int kk = k + g;
                    if (!sp.spec[kk].ok) {              // damped matrix not positive definite: Solver6 gives up here (failed only at iteration 0)
                        decided = true; mode = edsp::MODE_DONE;
                        if (lane == 0) { sv.failed = (iter == 0); sv.final_cost = cur_cost; sv.done = 1; }
                        break;
                    }
                    const int ok = fin && (cost < cur_cost);
                    last_ok = ok;
                    if (lane == 0 && ntrace < EDS_MAX_TRACE) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) sv.tr_xi[ntrace][i] = sp.spec[kk].xi[i];
                        sv.tr_cost[ntrace] = cost; sv.tr_acc[ntrace] = ok;
                    }
                    if (ntrace < EDS_MAX_TRACE) ++ntrace;
                    ++iter;
                    if (ok) {
                        acc = 1 + g;
                        if (lane < EDS_RED_N6) sp.cur[lane] = s;
                        lambda *= 0.5;
                        if (lane == 0) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) sv.p[i] = sp.spec[kk].p[i];
#pragma unroll
                            for (int i = 0; i < 4; ++i) sv.q[i] = sp.spec[kk].q[i];
                        }
                    } else {
                        lambda = edsp::next_lambda_after_reject(lambda);
                    }
                    if (iter >= max_iters) {            // Solver6::finish with the residuals kept by the caller: done, no extra pass
                        decided = true; mode = edsp::MODE_DONE;
                        if (lane == 0) { sv.final_cost = ok ? cost : cur_cost; sv.done = 1; }
                    } else if (ok) {
                        decided = true;                 // fresh linearisation: new proposals (everything behind this candidate is discarded)
                    }
                }
                if (!decided && k + GROUPS < EDS_NSPEC) { mode = edsp::MODE_USE; nk = k + GROUPS; }      // a whole round rejected: the next G prepared candidates
                if (lane == 0) {
                    sv.lambda = lambda; sv.have_cur = have_cur; sv.iter = iter; sv.ntrace = ntrace; sv.last_accepted = last_ok;
                    sp.mode = mode; sp.k = nk;
                    s_accept = acc;
                    if (mode == edsp::MODE_DONE) s_state = 2;
                }
                if (mode == edsp::MODE_SOLVE) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < EDS_NSPEC) edsp::propose(sp.cur, lambda, lane, sv.p, sv.q, sp.spec[lane]);
                }
              }
            } else
            if (wave == 0) {
                double s = 0.0;
                if (lane < EDS_RED_N6) {
                    if (TEAM > 1) {
#pragma unroll
                        for (int m = 0; m < TEAM; ++m)
                            s += __longlong_as_double((long long)(((unsigned long long)s_xchg[m][2 * lane + 1] << 32) | s_xchg[m][2 * lane]));
                    } else {
                        float part[EDS_FUSED_MAX_WAVES];
#pragma unroll
                        for (int wv = 0; wv < EDS_FUSED_MAX_WAVES; ++wv) part[wv] = s_red[wv][lane];
#pragma unroll
                        for (int wv = 0; wv < EDS_FUSED_MAX_WAVES; ++wv) s += (double)part[wv];
                    }
                }
                EDS_STAMP(2);
                EDS_STAMP2(3);
                const int k = sp.k, max_iters = sv.max_iters;
                int have_cur = sv.have_cur, iter = sv.iter, ntrace = sv.ntrace;
                double lambda = sv.lambda;
                const double cur_cost = sp.cur[EDS_RED_N6 - 1];
                if (QUICK && lane == EDS_RED_N6 - 1) s = quick_cost;         // ONE cost per pass: the value every wavefront took its accept test on
                const long long sb = __double_as_longlong(s);
                const double cost = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(sb >> 32), EDS_RED_N6 - 1) << 32) |
                                                         (unsigned int)__builtin_amdgcn_readlane((int)(sb & 0xffffffffll), EDS_RED_N6 - 1));
                const bool bad = lane < EDS_RED_N6 && !(fabs(s) < 1e300);
                const bool fin = __ballot(bad) == 0ull;
                int mode = edsp::MODE_SOLVE, accepted = 1, nk = 0;
                bool store = false;
                if (!have_cur) {                    // first pass: linearise at the start pose
                    if (!fin) { mode = edsp::MODE_DONE; if (lane == 0) { sv.failed = 1; sv.done = 1; } }
                    else { store = true; have_cur = 1; if (lane == 0) sv.initial_cost = cost; }
                } else {
                    const int ok = fin && (cost < cur_cost);
                    accepted = ok;
                    if (lane == 0 && ntrace < EDS_MAX_TRACE) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) sv.tr_xi[ntrace][i] = sp.spec[k].xi[i];
                        sv.tr_cost[ntrace] = cost; sv.tr_acc[ntrace] = ok;
                    }
                    if (ntrace < EDS_MAX_TRACE) ++ntrace;
                    ++iter;
                    if (ok) {
                        store = true;
                        lambda *= 0.5;
                        if (lane == 0) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) sv.p[i] = sp.spec[k].p[i];
#pragma unroll
                            for (int i = 0; i < 4; ++i) sv.q[i] = sp.spec[k].q[i];
                        }
                    } else {
                        lambda = edsp::next_lambda_after_reject(lambda);
                    }
                    if (iter >= max_iters) {        // Solver6::finish with the residuals kept by the caller: done, no extra pass
                        mode = edsp::MODE_DONE;
                        if (lane == 0) { sv.final_cost = ok ? cost : cur_cost; sv.done = 1; }
                    } else if (!ok && k + 1 < EDS_NSPEC) {
                        mode = edsp::MODE_USE; nk = k + 1;      // the candidate prepared for exactly this lambda
                    }
                }
                if (store && lane < EDS_RED_N6) sp.cur[lane] = s;
                if (lane == 0) {
                    sv.lambda = lambda; sv.have_cur = have_cur; sv.iter = iter; sv.ntrace = ntrace; sv.last_accepted = accepted;
                    sp.mode = mode; sp.k = nk;
                    s_accept = accepted;
                    if (mode == edsp::MODE_DONE) s_state = 2;
                }
                if (mode == edsp::MODE_SOLVE) {     // fresh linearisation (or candidates used up): lane w solves it for the lambda
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // that w rejections in a row would lead to — one
                    __builtin_amdgcn_wave_barrier();                             // instruction stream, EDS_NSPEC lanes
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < EDS_NSPEC) edsp::propose(sp.cur, lambda, lane, sv.p, sv.q, sp.spec[lane]);
                }
            }
            EDS_STAMP(3);
            __syncthreads();
            if (TEAM > 1 && s_timeout) {            // a team member never showed up within the bound: give up cleanly, the host re-runs without teams
                if (tid == 0) { sv.failed = 2; sv.done = 1; }
                break;
            }
            if (GROUPS > 1 ? (s_accept < 0 || s_accept == 1 + group) : s_accept != 0) {
#pragma unroll
                for (int j = 0; j < NREG; ++j) racc[j] = rcand[j];
            }
            if (GROUPS > 1 && s_accept != 0) res_owner = s_accept < 0 ? 0 : s_accept - 1;
            if (QUICK) { k_r = sp.k; iter_r = sv.iter; have_cur_r = sv.have_cur; cur_cost_r = sp.cur[EDS_RED_N6 - 1]; }
            if (sp.mode == edsp::MODE_DONE) break;
            if (!sp.spec[sp.k].ok) {                // damped matrix not positive definite: Solver6 gives up here (failed only at iteration 0)
                if (tid == 0) { sv.failed = (sv.iter == 0); sv.final_cost = sp.cur[EDS_RED_N6 - 1]; sv.done = 1; }
                break;
            }
            continue;
        }
        if (tid < EDS_RED_N6) {          // cross-wavefront sum in fp64, unpacked straight into the solver's input
            float part[EDS_FUSED_MAX_WAVES];
#pragma unroll
            for (int wv = 0; wv < EDS_FUSED_MAX_WAVES; ++wv) part[wv] = s_red[wv][tid];   // 16 independent LDS reads in flight
            double s = 0.0;
#pragma unroll
            for (int wv = 0; wv < EDS_FUSED_MAX_WAVES; ++wv) s += (double)part[wv];
            if (tid < 21) {
                int a = 0, rem = tid;           // record index -> (a, b) of the upper triangle
                while (rem >= 6 - a) { rem -= 6 - a; ++a; }
                const int b = a + rem;
                s_sums.H[6 * a + b] = s;
                s_sums.H[6 * b + a] = s;
            } else if (tid < 27) {
                s_sums.b[tid - 21] = s;
            } else {
                s_sums.cost = s;
            }
        }
        // the 28 summing lanes and the solver lane all live in wavefront 0: LDS operations of one
        // wavefront retire in order, so a wavefront-scope fence replaces a second workgroup barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        EDS_STAMP(2);
        EDS_STAMP2(3);
        if (tid == 0) {
            sv.on_eval(s_sums);
            s_accept = sv.last_accepted;
            if (sv.done) {
                s_state = 2;
            } else {
                edsm::fill_pose_rt(sv.cp, sv.cq, s_pose);
                for (int i = 0; i < 9; ++i) s_posef[i] = (float)s_pose[EDS_PB_D + i];
                for (int i = 0; i < 3; ++i) s_posef[9 + i] = (float)s_pose[EDS_PB_T + i];
                s_state = sv.final_pass ? 1 : 0;
            }
        }
        EDS_STAMP(3);
        __syncthreads();
        if (PPT > 0 && s_accept) {      // the pass just consumed became the accepted pose: its residuals are the ones to keep
#pragma unroll
            for (int j = 0; j < NREG; ++j) racc[j] = rcand[j];
        }
        if (s_state == 2) break;
    }
    if (PPT > 0) {                      // residuals at the accepted pose (what Tracker.cpp:223-230 stores), kept in registers
#pragma unroll                          // all along: the damped solver needs no extra pass to produce them
        for (int j = 0; j < NREG; ++j) {
            const int i = tid + j * nthr;
            if (i < N && group == res_owner) A.r[base + i] = racc[j];
            if (A.rmap && i < N && group == res_owner) A.rmap[base + i] = racc[j];      // the caller reads them next (Tracker.cpp:223-233)
        }
    }
#ifdef EDS_FUSED_STAMPS
    // diagnostic build only: cycles of lane 0 in [point loop | reduction | solver] into the pad words
    if (tid == 0 && member == 0 && group == 0) { out[slot].pad[0] = (double)stamp_acc[0]; out[slot].pad[1] = (double)stamp_acc[1]; out[slot].pad[2] = (double)stamp_acc[2]; }
#if EDS_FUSED_STAMPS == 2
    if (tid == 0 && member == 0 && group == 0) { out[slot].pad[0] = (double)stamp2_acc[0]; out[slot].pad[1] = (double)stamp2_acc[1]; out[slot].pad[2] = (double)stamp2_acc[2]; }
#endif
#if EDS_FUSED_STAMPS == 3
    __syncthreads();
    if (tid == 0 && member == 0 && group == 0) { out[slot].pad[0] = (double)s_miss_total; out[slot].pad[1] = (double)s_pass_total; out[slot].pad[2] = (double)N; }
#endif
#endif

    // small solves: this workgroup's completion word (EdsArrays::done) — every thread's writes are out at system scope first
    unsigned* const done_word = A.done ? A.done + (size_t)slot * EDS_DONE_WORDS + (TEAM > 1 ? group * TEAM + member : 0) : nullptr;
    // (every wavefront waits for ITS stores to be acknowledged — the mirror is uncached host memory, nothing sits in the L2 — and the
    // barrier hands that to thread 0, whose release store at system scope follows; a __threadfence_system() per thread instead wrote the
    // L2 back and invalidated it on every wavefront of up to 32 workgroups: +2.7 us per solve, measured)
    if (done_word) { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); }
    __syncthreads();                    // the solver state as its last writer left it
    if (TEAM > 1 && (member != 0 || group != 0)) {          // every member holds the same result; member 0 (of group 0) reports it
        if (tid == 0 && done_word) __hip_atomic_store(done_word, A.done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (tid == 0) {
        EdsFusedOut& O = out[slot];
        for (int i = 0; i < 3; ++i) O.p[i] = sv.p[i];
        for (int i = 0; i < 4; ++i) O.q[i] = sv.q[i];
        O.initial_cost = sv.initial_cost; O.final_cost = sv.final_cost;
        O.iterations = sv.iter; O.ntrace = sv.ntrace; O.failed = sv.failed;
        int na = 0;
        for (int k = 0; k < sv.ntrace; ++k) na += sv.tr_acc[k];
        O.naccepted = na;
        O.t_end = __builtin_amdgcn_s_memrealtime();
        // (the completion word is released at SYSTEM scope — the record above is visible to a host that sees the word — and only where a
        // host polls it: a system-scope release writes the L2 back, and 4 096 workgroups of a batch doing that at their ends cost the
        // headline kernels 5-9 % when round 6 first released t_end that way in every launch: same-box A/B against round 5's build)
        if (done_word) __hip_atomic_store(done_word, A.done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // full solver state (trace) to HBM, cooperatively
    {
        const int nwords = (int)(sizeof(edss::Solver6) / sizeof(int));
        const int* src = reinterpret_cast<const int*>(&sv);
        int* dst = reinterpret_cast<int*>(sv_all + slot);
        for (int i = tid; i < nwords; i += nthr) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------------------------------
// Two translation units from this one source (Makefile).  The machine scheduler's ILP-first strategy is worth +2 % on the bicubic
// instantiations (and 6 % on a lone alignment) but costs the bilinear ones 9 % (20.6 -> 18.7 M iterations/s at 4 096 alignments),
// and the strategy is a per-TU compiler flag: eds_fused_bilinear.o (-DEDS_FUSED_BILINEAR_TU, register-pressure trackers only) holds
// the SAMPLING = 1 instantiations behind eds_fused6_launch_bilinear(); eds_fused.o holds the rest and the host side.
#if defined(EDS_FUSED_BILINEAR_TU) || defined(EDS_FUSED_ONE_TU)      // (ONE_TU: the diagnostic builds, one command line for every source)
void eds_fused6_launch_bilinear(const EdsFused6Launch& L, int ppt, int team, int groups) {
    edss::Solver6* svp = reinterpret_cast<edss::Solver6*>(L.sv);
#define EDS_BL(P, T, K)                                                                                                            \
    hipLaunchKernelGGL((eds_fused6_kernel<1, P, T, 0, K>), dim3(L.count * K - L.drop), dim3(L.threads), 0, L.st, *L.A, L.in, L.out, svp,      \
                       L.first, L.iters, L.damped, L.lambda0, L.tau, L.nb, L.mail, L.ticket, L.ticket_base, L.epoch)
    if (groups > 1) {           // candidate groups (eds_launch_rule.hpp: EDS_FUSED6_BILINEAR_GROUP_INSTANCES)
#define EDS_BLG_(S_, P_, T_, Q_, K_, G_)                                                                                            \
        if (ppt == P_ && team == K_ && groups == G_) {                                                                              \
            hipLaunchKernelGGL((eds_fused6_kernel<S_, P_, T_, Q_, K_, G_>), dim3(L.count * K_ * G_ - L.drop), dim3(L.threads), 0, L.st, *L.A, L.in, L.out, svp, \
                               L.first, L.iters, L.damped, L.lambda0, L.tau, L.nb, L.mail, L.ticket, L.ticket_base, L.epoch);       \
            return;                                                                                                                 \
        }
        EDS_FUSED6_BILINEAR_GROUP_INSTANCES(EDS_BLG_)
#undef EDS_BLG_
        return;
    }
    if (team > 1) {
        if (ppt == 1) { EDS_BL(1, 512, 4); }
        else if (team == 2) { EDS_BL(2, 512, 2); }
        else if (team == 4) { EDS_BL(2, 512, 4); }
        else if (team == 8) { EDS_BL(2, 512, 8); }
        else { EDS_BL(2, 512, 16); }
        return;
    }
    const bool big = L.threads > 512;
    switch (ppt) {
        case 1: if (big) EDS_BL(1, 1024, 1); else EDS_BL(1, 512, 1); break;
        case 2: if (big) EDS_BL(2, 1024, 1); else EDS_BL(2, 512, 1); break;
        case 4: if (big) EDS_BL(4, 1024, 1); else EDS_BL(4, 512, 1); break;
        default: if (big) EDS_BL(0, 1024, 1); else EDS_BL(0, 512, 1); break;
    }
#undef EDS_BL
}
#endif
#ifndef EDS_FUSED_BILINEAR_TU
int eds_fused_alloc(EdsFusedBuffers* fb, int B) {
    fb->B = B;
    // start states and compact results live in pinned host memory that the kernels read / write directly (a solve moves ~100 bytes
    // each way per alignment): no copy calls on the latency path of a single optimize — each HIP call costs 3-5 us of host time
    if (hipMalloc(&fb->d_sv, sizeof(edss::Solver6) * (size_t)B) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&fb->h_in, sizeof(EdsFusedIn) * B, hipHostMallocMapped) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&fb->h_out, sizeof(EdsFusedOut) * B, hipHostMallocMapped) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&fb->h_out12, sizeof(EdsFused12Out) * B, hipHostMallocMapped) != hipSuccess) return -1;
    if (hipHostGetDevicePointer((void**)&fb->d_in, fb->h_in, 0) != hipSuccess) return -1;
    if (hipHostGetDevicePointer((void**)&fb->d_out, fb->h_out, 0) != hipSuccess) return -1;
    if (hipHostGetDevicePointer((void**)&fb->d_out12, fb->h_out12, 0) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&fb->h_done, sizeof(unsigned) * EDS_DONE_WORDS * (size_t)B, hipHostMallocMapped) != hipSuccess) return -1;
    if (hipHostGetDevicePointer((void**)&fb->d_done, fb->h_done, 0) != hipSuccess) return -1;
    std::memset(fb->h_done, 0, sizeof(unsigned) * EDS_DONE_WORDS * (size_t)B);
    // team launches (eds_fused6_kernel TEAM > 1): granule mailboxes of EDS_TEAM_SLOTS alignments + the ticket counter
    if (hipMalloc((void**)&fb->d_mail, EDS_TEAM_MAIL_BYTES) != hipSuccess) return -1;
    if (hipMemset(fb->d_mail, 0, EDS_TEAM_MAIL_BYTES) != hipSuccess) return -1;
    if (hipMalloc((void**)&fb->d_ticket, sizeof(int)) != hipSuccess) return -1;
    if (hipMemset(fb->d_ticket, 0, sizeof(int)) != hipSuccess) return -1;
    fb->ticket_base = 0;
    fb->epoch = 0;
    std::memset(fb->h_out12, 0, sizeof(EdsFused12Out) * B);
    std::memset(fb->h_in, 0, sizeof(EdsFusedIn) * B);
    std::memset(fb->h_out, 0, sizeof(EdsFusedOut) * B);
    return 0;
}

// The tag of the next small launch (never 0).  When the 32-bit sequence wraps, the words are cleared: a slot that has not been solved for
// four thousand million launches must not meet its old tag again (nothing is in flight here: the previous batch was collected).
unsigned eds_next_done_tag(EdsFusedBuffers* fb) {
    if (++fb->done_seq == 0u) {
        std::memset(fb->h_done, 0, sizeof(unsigned) * EDS_DONE_WORDS * (size_t)fb->B);
        fb->done_seq = 1u;
    }
    return fb->done_seq;
}

bool eds_team_allowed(EdsFusedBuffers* fb) {
    if (fb->team_cooldown <= 0) return true;
    if (--fb->team_cooldown == 0) fb->team_clean = 0;      // re-armed: the next eligible solve forms teams again
    return false;
}
void eds_team_timed_out(eds_trk* h) {
    EdsFusedBuffers& fb = h->fused;
    // a time-out soon after a re-arm doubles the pause; after EDS_TEAM_REARM_CLEAN clean team launches it counts as a first one
    const int next = fb.team_backoff > 0 ? std::min(2 * fb.team_backoff, EDS_TEAM_COOLDOWN_MAX) : EDS_TEAM_COOLDOWN;
    fb.team_cooldown = fb.team_backoff = next;
    fb.team_clean = 0;
    // workgroups of the short launch may have taken tickets the host never accounted for (or the reverse): start over
    hipMemsetAsync(fb.d_ticket, 0, sizeof(int), h->st);
    fb.ticket_base = 0;
}
void eds_team_clean(EdsFusedBuffers* fb) {
    if (fb->team_backoff > 0 && ++fb->team_clean >= EDS_TEAM_REARM_CLEAN) { fb->team_backoff = 0; fb->team_clean = 0; }
}

void eds_fused_free(EdsFusedBuffers* fb) {
    if (fb->d_sv) hipFree(fb->d_sv);
    if (fb->h_in) hipHostFree(fb->h_in);
    if (fb->h_out) hipHostFree(fb->h_out);
    if (fb->d_mail) hipFree(fb->d_mail);
    if (fb->d_mail12) hipFree(fb->d_mail12);
    if (fb->d_ticket) hipFree(fb->d_ticket);
    if (fb->h_out12) hipHostFree(fb->h_out12);
    if (fb->h_done) hipHostFree(fb->h_done);
    *fb = EdsFusedBuffers();
}

int eds_fused_solve(eds_trk* h, int level, int first, int count) {
    // the persistent kernels are compiled for the tiled frame only; the row-major layout exists for the layout comparison of
    // the streaming kernels (EDS_FRAME_LAYOUT=rowmajor, tools/bench_layout.py) and is solved by the host-driven loop
    if (!h->tiled) return eds_internal_solve_host(h, level, first, count);
    const EdsKnobs& kn = h->knobs;                  // resolved at eds_trk_create / eds_trk_set_knob: no environment access on the solve path
    if (h->cfg.solver == EDS_SOLVER_REF12) {
        // The persistent REF12 kernels beat the host-driven loop at every batch size (one 2 000-point solve: 0.29 ms vs
        // 0.42 ms; B = 1024: 8.3 M vs 0.37 M LM iterations/s).  The host loop remains for what they do not cover
        // (more than 8 residual blocks).
        const bool want_device = kn.ref12_exec != 0;
        if (want_device && eds_fused12_supported(h, first, count)) return eds_fused12_solve(h, level, first, count);
        return eds_internal_solve_host(h, level, first, count);     // also: > 8 residual blocks
    }
    EdsFusedBuffers& fb = h->fused;
    if (fb.pending_count > 0) return eds_internal_fail(EDS_ERR_STATE, "previous batch not collected: call eds_trk_sync first");
    int lv = level < 0 ? 0 : (level >= EDS_MAX_LEVELS ? EDS_MAX_LEVELS - 1 : level);
    const int iters = h->cfg.max_num_iterations[lv];
    if (iters > EDS_MAX_TRACE) return eds_internal_fail(EDS_ERR_INVALID, "max_num_iterations exceeds EDS_MAX_TRACE for the device solver");
    int nb = h->cfg.num_blocks < 1 ? 1 : (h->cfg.num_blocks > EDS_MAX_BLOCKS ? EDS_MAX_BLOCKS : h->cfg.num_blocks);
    int maxN = 0;
    for (int s = first; s < first + count; ++s) {
        const Slot& sl = h->slots[s];
        if (!sl.has_kf || !sl.has_frame) return eds_internal_fail(EDS_ERR_STATE, "keyframe or event frame not set");
        if (sl.N > maxN) maxN = sl.N;
        EdsFusedIn& I = fb.h_in[s];
        std::memcpy(I.p, sl.p, sizeof(I.p)); std::memcpy(I.q, sl.q, sizeof(I.q)); std::memcpy(I.v, sl.v, sizeof(I.v));
    }
    hipError_t e = hipSuccess;
    EdsArrays A = h->arrays();
    const double tau = h->cfg.huber_tau > 0 ? h->cfg.huber_tau : 0.0;
    edss::Solver6* svp = reinterpret_cast<edss::Solver6*>(fb.d_sv);
    // WHICH kernel: eds_launch_rule.hpp (pure; table-tested on the CPU).  Its two questions with side effects are asked here: the
    // time-out policy of the teams (a cool-down is counted down when asked) and the strip copies (converted when a solve wants them).
    const EdsLm6In rin{maxN, count, h->cfg.sampling == EDS_SAMPLE_BICUBIC ? 1 : 0, iters, h->cfg.solver == EDS_SOLVER_LM6 ? 1 : 0, tau > 0 ? 1 : 0,
                       h->H, fb.pending_retry ? 1 : 0};
    EdsLm6Plan pl;
    eds_lm6_plan_begin(kn, rin, pl);
    const bool team_ok = pl.wants_team && eds_team_allowed(&fb);
    fb.pending_paused = pl.wants_team && !team_ok && !fb.pending_retry;
    eds_lm6_plan_team(kn, rin, team_ok ? 1 : 0, fb.team_cooldown > 0 ? 1 : 0, pl);
    fb.pending_ticks = count <= 64 && !(pl.team <= 1 && pl.stream);   // the latency regime: time stamps from inside the kernel instead of event packets
    if (!fb.pending_ticks) hipEventRecord(h->ev0, h->st);             // (a conversion of frames to strips that this solve asks for is inside the measured span)
    const bool strips = pl.strips_eligible && eds_strips_for_solve(h, first, count);
    eds_lm6_plan_finish(kn, rin, strips ? 1 : 0, pl);
    A.strips = h->dstrips; A.strip_phases = h->strip_phases;
    const int team = pl.kind == EDS_K6_TEAM ? pl.K : 1, damped = pl.damped, threads = pl.threads;
    const int groups = pl.kind == EDS_K6_TEAM ? pl.G : 1;       // candidate groups: G x K workgroups per alignment
    // test hook for the time-out path: a team launch goes out one workgroup short, so its last team never completes, reports a time-out
    // after EDS_TEAM_TIMEOUT_TICKS and eds_fused_collect re-runs the range without teams (tests/test_team_timeout_gpu.py)
    const int drop = kn.team_drop ? 1 : 0;
    // team launches (the latency regime) write the kept residuals into the pinned mirror themselves: one launch less behind the solve
    // (round 6: so do the one-CU kernels that keep their points in registers, when the launch reports through done words)
    const bool done_words = fb.pending_ticks && kn.poll_results && !drop && pl.kind != EDS_K6_STREAM && team * groups <= EDS_DONE_WORDS;
    const bool rmap_in_kernel = (team > 1 || (done_words && pl.kind == EDS_K6_FUSED && pl.P > 0)) && h->d_rmap && first + count <= EDS_RHOST_SLOTS;
    A.rmap = rmap_in_kernel ? h->d_rmap : nullptr;
    fb.pending_vteam = 0;
    if (done_words) { A.done = fb.d_done; A.done_tag = eds_next_done_tag(&fb); fb.pending_vteam = team * groups; }
    fb.pending_team = team; fb.pending_level = level;
    if (pl.kind != EDS_K6_STREAM && !eds_fused6_instance_exists(pl.S, pl.P, pl.T, pl.Q, pl.K, pl.bilinear_tu, groups))
        return eds_internal_fail(EDS_ERR_INVALID, "internal: the launch rule chose an instantiation the library does not hold");
    // one launch of the chosen instantiation over `cnt` alignments from slot `f0` (teams: cnt * K - drop workgroups of 512 threads)
    auto launch = [&](int f0, int cnt, unsigned ticket_base) {
        const int K = pl.K, wg = cnt * K * groups - (K > 1 ? drop : 0), block = K > 1 ? 512 : threads;
        if (groups > 1) std::snprintf(fb.last_kernel, sizeof(fb.last_kernel), "eds_fused6_kernel<%d, %d, %d, %d, %d, %d>", pl.S, pl.P, pl.note_T, pl.Q, K, groups);
        else std::snprintf(fb.last_kernel, sizeof(fb.last_kernel), "eds_fused6_kernel<%d, %d, %d, %d, %d>", pl.S, pl.P, pl.note_T, pl.Q, K);
        fb.last_workgroups = wg; fb.last_team = K * groups; fb.last_layout = pl.Q >= 3 ? 2 : 1;
        if (pl.bilinear_tu) {
            const EdsFused6Launch L{&A, fb.d_in, fb.d_out, fb.d_sv, f0, cnt, block, iters, damped, h->cfg.lambda0, tau, nb,
                                    K > 1 ? fb.d_mail : nullptr, K > 1 ? fb.d_ticket : nullptr, K > 1 ? ticket_base : 0u, K > 1 ? fb.epoch : 0u,
                                    K > 1 ? drop : 0, h->st};
            eds_fused6_launch_bilinear(L, pl.P, K, groups);
            return;
        }
        unsigned long long* mail = K > 1 ? fb.d_mail : nullptr;
        int* ticket = K > 1 ? fb.d_ticket : nullptr;
        const unsigned tb = K > 1 ? ticket_base : 0u, ep = K > 1 ? fb.epoch : 0u;
#define EDS_INST_LAUNCH_(S_, P_, T_, Q_, K_)                                                                                          \
        if (pl.S == S_ && pl.P == P_ && pl.T == T_ && pl.Q == Q_ && K == K_) {                                                       \
            hipLaunchKernelGGL((eds_fused6_kernel<S_, P_, T_, Q_, K_>), dim3(wg), dim3(block), 0, h->st, A, fb.d_in, fb.d_out, svp, f0, \
                               iters, damped, h->cfg.lambda0, tau, nb, mail, ticket, tb, ep);                                        \
            return;                                                                                                                   \
        }
        if (groups > 1) {
#define EDS_INST_LAUNCH6_(S_, P_, T_, Q_, K_, G_)                                                                                      \
            if (pl.S == S_ && pl.P == P_ && pl.T == T_ && pl.Q == Q_ && K == K_ && groups == G_) {                                    \
                hipLaunchKernelGGL((eds_fused6_kernel<S_, P_, T_, Q_, K_, G_>), dim3(wg), dim3(block), 0, h->st, A, fb.d_in, fb.d_out, svp, f0, \
                                   iters, damped, h->cfg.lambda0, tau, nb, mail, ticket, tb, ep);                                    \
                return;                                                                                                               \
            }
            EDS_FUSED6_GROUP_INSTANCES(EDS_INST_LAUNCH6_)
#undef EDS_INST_LAUNCH6_
            return;
        }
        EDS_FUSED6_MAIN_INSTANCES(EDS_INST_LAUNCH_)
#undef EDS_INST_LAUNCH_
    };
    if (pl.kind == EDS_K6_TEAM) {
        // one launch holds EDS_TEAM_MEMBERS workgroups (the mailboxes' capacity); a larger range goes out in several launches, in
        // stream order, each with its own launch number in the granule tags and its own stretch of tickets
        const int per_launch = EDS_TEAM_MEMBERS / (team * groups);
        for (int c0 = 0; c0 < count; c0 += per_launch) {
            const int f0 = first + c0, cnt = std::min(per_launch, count - c0);
            if (++fb.epoch >= (1u << 24)) {              // tags are (epoch << 8 | pass): start over with clean mailboxes
                hipMemsetAsync(fb.d_mail, 0, EDS_TEAM_MAIL_BYTES, h->st);
                fb.epoch = 1;
            }
            const unsigned ticket_base = fb.ticket_base;
            fb.ticket_base += (unsigned)(cnt * team * groups);
            for (int s = f0; s < f0 + cnt; ++s) { fb.h_out[s].failed = 2; fb.h_out[s].t_end = 0; }      // "no result yet": what a workgroup that never ran leaves behind reads as a time-out
            launch(f0, cnt, ticket_base);
        }
    } else if (pl.kind == EDS_K6_STREAM) {
        std::snprintf(fb.last_kernel, sizeof(fb.last_kernel), "eds_stream6_kernel<%d, %d, %d>", pl.S, pl.T, pl.P);
        fb.last_workgroups = count; fb.last_team = 1; fb.last_layout = 1;
        eds_stream6_launch(A, h->cfg.sampling, pl.wide ? 1 : 0, fb.d_in, fb.d_out, fb.d_sv, first, count, iters, damped, h->cfg.lambda0, tau, nb, h->st);
    } else {
        if (fb.pending_ticks) for (int s = first; s < first + count; ++s) fb.h_out[s].t_end = 0;      // (the completion word wait_stream polls)
        launch(first, count, 0u);
    }
    if (!fb.pending_ticks) hipEventRecord(h->ev1, h->st);
    fb.pending_host_r = rmap_in_kernel ? true : eds_mirror_residuals(h, first, count);
    if (fb.pending_host_r && !rmap_in_kernel) fb.pending_vteam = 0;      // a mirror launch follows the solve: only the stream says when IT is through
    e = hipGetLastError();
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    fb.pending_first = first;
    fb.pending_count = count;
    fb.pending_kind = 6;
    fb.last_first = first; fb.last_count = count; fb.last_kind = 6; fb.last_ticks = fb.pending_ticks;
    fb.launch_wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return EDS_OK;
}

int eds_fused_collect(eds_trk* h) {
    EdsFusedBuffers& fb = h->fused;
    if (fb.pending_count <= 0) return EDS_OK;
    if (fb.pending_kind == 12) return eds_fused12_collect(h);
    const double now = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    float dev_ms = 0.f;
    if (fb.pending_ticks) {
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) { t0 = std::min(t0, fb.h_out[s].t_begin); t1 = std::max(t1, fb.h_out[s].t_end); }
        dev_ms = t1 > t0 ? (float)((double)(t1 - t0) * 1e-5) : 0.f;        // 100 MHz ticks
    } else {
        hipEventElapsedTime(&dev_ms, h->ev0, h->ev1);
    }
    if (fb.pending_team == 1 && h->knobs.report) {     // diagnostic: what the workgroups' own begin / end stamps say about the launch
        std::vector<unsigned long long> te;
        unsigned long long t0 = ~0ull, t1 = 0; double busy = 0.0;
        for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) {
            t0 = std::min(t0, fb.h_out[s].t_begin); t1 = std::max(t1, fb.h_out[s].t_end);
            busy += (double)(fb.h_out[s].t_end - fb.h_out[s].t_begin); te.push_back(fb.h_out[s].t_end);
        }
        std::sort(te.begin(), te.end());
        double tail = 0.0; const int nc = std::min(256, (int)te.size());
        for (int i = 0; i < nc; ++i) tail += (double)(t1 - te[te.size() - 1 - i]);       // the last workgroup of each CU = the 256 latest ends
        const double span = (double)(t1 - t0);
        std::fprintf(stderr, "[eds_fused] %d alignments: span %.1f us, mean workgroup %.1f us, covered %.3f of 256 CUs x span, tail idle %.1f us per CU\n",
                     fb.pending_count, span * 1e-2, busy / fb.pending_count * 1e-2, span > 0 ? busy / (256.0 * span) : 0.0, tail / nc * 1e-2);
    }
    if (fb.pending_team > 1) {                        // a team whose members did not all become resident within the bound
        bool timed_out = false;
        for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) timed_out |= fb.h_out[s].failed == 2;
        if (timed_out) {                             // solve the range once more, one CU per alignment; teams pause on this handle (eds_fused.hpp)
            eds_team_timed_out(h);
            const int pf = fb.pending_first, pc = fb.pending_count;
            fb.pending_count = 0;
            fb.pending_retry = true;
            int rc = eds_fused_solve(h, fb.pending_level, pf, pc);
            if (rc != EDS_OK) { fb.pending_retry = false; return rc; }
            hipError_t e = hipStreamSynchronize(h->st);
            if (e != hipSuccess) { fb.pending_retry = false; return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e)); }
            return eds_fused_collect(h);
        }
        eds_team_clean(&fb);
    }
    for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) {
        Slot& sl = h->slots[s];
        const EdsFusedOut& O = fb.h_out[s];
        const bool ok = O.failed == 0;
        if (ok) { std::memcpy(sl.p, O.p, sizeof(sl.p)); std::memcpy(sl.q, O.q, sizeof(sl.q)); }
        sl.res_on_device = ok; sl.res_in_hostmap = ok && fb.pending_host_r;
        sl.trace_on_device = true;
        sl.residuals.clear();
        sl.ntrace = O.ntrace;
        eds_trk_info& in = sl.info;
        std::memset(&in, 0, sizeof(in));
        in.meas_time_us = now - fb.launch_wall_us;
        in.time_seconds = in.meas_time_us * 1e-6;
        in.device_time_us = dev_ms * 1e3;
        in.num_points = sl.N;
        in.num_iterations = O.iterations;
        in.success = ok;
        in.flags = (fb.pending_retry ? EDS_INFO_TEAM_TIMEOUT : 0) | (fb.pending_paused ? EDS_INFO_TEAMS_PAUSED : 0);
        in.termination = ok ? edss::TERM_NO_CONVERGENCE : edss::TERM_FAILURE;
        in.num_successful_steps = O.naccepted;
        in.num_unsuccessful_steps = O.ntrace - O.naccepted;
        in.initial_cost = 0.5 * O.initial_cost;
        in.final_cost = 0.5 * O.final_cost;
    }
#ifdef EDS_FUSED_STAMPS
    {
        double a[3] = {0, 0, 0};
        for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s)
            for (int k = 0; k < 3; ++k) a[k] += fb.h_out[s].pad[k];
        const double n = fb.pending_count, passes = fb.h_out[fb.pending_first].iterations + 2.0;
#if EDS_FUSED_STAMPS == 3
        fprintf(stderr, "[stamps3] patches gathered %.0f of %.0f point-passes (%.1f passes per alignment): miss rate %.3f (mean over %d slots)\n",
                a[0] / n, a[1] / n * a[2] / n, a[1] / n, a[0] / (a[1] * a[2] / n), fb.pending_count);
#else
        fprintf(stderr, "[stamps] lane-0 cycles per pass: points %.0f  reduce %.0f  solver %.0f  (mean over %d slots, %g passes)\n",
                a[0] / n / passes, a[1] / n / passes, a[2] / n / passes, fb.pending_count, passes);
#endif
    }
#endif
    fb.pending_count = 0;
    fb.pending_retry = false; fb.pending_paused = false;
    return EDS_OK;
}

// eds_trk_last_launch: what the last on-device solve launched, and (one CU per alignment) the digest of its workgroups' stamps
int eds_fused_last_launch(eds_trk* h, eds_trk_launch_info* out) {
    EdsFusedBuffers& fb = h->fused;
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->kernel, sizeof(out->kernel), "%s", fb.last_kernel);
    out->workgroups = fb.last_workgroups; out->cus_per_alignment = fb.last_team;
    out->first = fb.last_first; out->count = fb.last_count;
    out->layout = h->tiled ? fb.last_layout : 0;
    out->timing_source = fb.last_ticks ? 1 : 0;
    if (fb.last_count <= 0 || fb.last_team != 1 || fb.pending_count > 0) return EDS_OK;
    std::vector<unsigned long long> te;
    te.reserve(fb.last_count);
    unsigned long long t0 = ~0ull, t1 = 0; double busy = 0.0;
    for (int s = fb.last_first; s < fb.last_first + fb.last_count; ++s) {
        const unsigned long long b = fb.last_kind == 12 ? fb.h_out12[s].t_begin : fb.h_out[s].t_begin, e = fb.last_kind == 12 ? fb.h_out12[s].t_end : fb.h_out[s].t_end;
        if (e <= b) continue;
        t0 = std::min(t0, b); t1 = std::max(t1, e); busy += (double)(e - b); te.push_back(e);
    }
    if (te.empty() || t1 <= t0) return EDS_OK;
    std::sort(te.begin(), te.end());
    const int nc = std::min(256, (int)te.size());
    double tail = 0.0;
    for (int i = 0; i < nc; ++i) tail += (double)(t1 - te[te.size() - 1 - i]);          // the last workgroup of each CU = the 256 latest ends
    const double span = (double)(t1 - t0);
    out->span_us = span * 1e-2; out->mean_workgroup_us = busy / (double)te.size() * 1e-2;
    out->covered = busy / (256.0 * span); out->tail_idle_us = tail / nc * 1e-2;
    return EDS_OK;
}

int eds_fused_fetch_trace(eds_trk* h, int slot) {
    Slot& sl = h->slots[slot];
    if (!sl.trace_on_device) return EDS_OK;
    // (the kernel copies its solver state — the trace — to HBM BEHIND its completion word, and this copy runs on the null stream, which
    // does not wait for the handle's non-blocking one: a solve that was seen complete through the words has its stream waited for here)
    { const int rc_ = eds_stream_idle(h); if (rc_ != EDS_OK) return rc_; }
    edss::Solver6* tmp = new edss::Solver6();
    hipError_t e = hipMemcpy(tmp, reinterpret_cast<edss::Solver6*>(h->fused.d_sv) + slot, sizeof(edss::Solver6), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete tmp; return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e)); }
    sl.ntrace = tmp->ntrace;
    sl.tr_xi.assign(&tmp->tr_xi[0][0], &tmp->tr_xi[0][0] + 6 * tmp->ntrace);
    sl.tr_cost.assign(tmp->tr_cost, tmp->tr_cost + tmp->ntrace);
    sl.tr_acc.assign(tmp->tr_acc, tmp->tr_acc + tmp->ntrace);
    sl.trace_on_device = false;
    delete tmp;
    return EDS_OK;
}
#endif   // EDS_FUSED_BILINEAR_TU
