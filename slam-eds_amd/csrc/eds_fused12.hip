// Persistent on-device solve of THE REFERENCE PROBLEM (EDS_SOLVER_REF12 with EDS_EXEC_DEVICE):
// 12 local parameters (translation, quaternion-local, unit velocity), per-block model normalisation and
// robust loss, Ceres trust-region LM semantics (reference Tracker.cpp:104-241, PhotometricError.hpp:124-182).
//
// One workgroup owns one alignment and runs every evaluation of the solve without returning to the host:
//
//   every lane   per evaluation: its points' constants from HBM/L2 (36 B per point, coalesced — small next to the
//                ~200 B scattered frame read; keeping them in registers for the whole solve, as the pose-only kernel
//                does, spilled 0.7-2 KB per lane here), projection in fp32 through the small-displacement form of
//                eds_device.hpp, bicubic / bilinear sample from the LDS patch cache or HBM (two points per lane in
//                flight), residual and the 1x12 row in closed form
//   wavefront    its 64 rows [J | r] (13 floats, stride 17: conflict-free) go through a private LDS staging area into
//                the operand layout of v_mfma_f64_16x16x4_f64, which forms X^T X for the 64 x 13 matrix X in 16
//                instructions with FOUR accumulator registers per lane — 91 running sums per lane plus a 91-value
//                butterfly do not fit the register file next to a bicubic gather.  The matrix core is a register-free
//                reduction primitive here (products and sums in fp64), not a throughput device: the kernel stays
//                bound by the instruction stream around the scattered frame reads.  Tiles are added to the per-block
//                sums in LDS with fp64 atomics; the reference normalises the model and applies its loss per residual
//                block (options.num_threads blocks, Tracker.cpp:178-195), a wavefront's 64 consecutive points touch
//                one block, rarely two, and the tile is flushed when the block changes (up to EDS_DEV_MAX_BLOCKS = 8)
//   wavefront 0  runs the LM state machine of edss::Solver12 spread over its 64 lanes (eds_solver12_coop.hpp)
//
// Candidate residuals go to the (otherwise unused) mhat plane and are copied to the residual plane when the candidate
// is accepted, so no residual pass is needed at the end.  Any number of points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "eds_device.hpp"
#include "eds_fused.hpp"
#ifndef EDS_REF12_PREFETCH
#define EDS_REF12_PREFETCH 1
#endif
#include "eds_handle.hpp"
#include "eds_launch_rule.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"
#include "eds_solver12_coop.hpp"

using namespace edsd;

typedef double acc4d __attribute__((ext_vector_type(4)));

// TEAM = K > 1 (latency regime, see eds_fused.hip): K workgroups share one alignment; member m evaluates a contiguous slice of the
// points, the members exchange their per-block sums (157 doubles per residual block) as tagged granules after every evaluation,
// add them in member order, and all run the LM state machine on the identical totals.
// QUAD (bicubic): the quad-cooperative gather of eds_device.hpp / eds_fused.hip — lane j of a quad loads row j of each of the quad's
// four patches, the row splines run where the rows landed, a DPP transpose returns them to the point's own lane.  QUAD = 2 (round 3):
// the same on the STRIP copies of the frames (eds_layout.hpp) — one 16-byte read per patch row at a 4-byte-aligned address instead of
// two aligned pieces and a barrel shift, 2.5 instead of 3.06 sectors per patch.
//
// GROUPS = G > 1 (round 5; teams only): SPECULATIVE CANDIDATE GROUPS, as in eds_fused6_kernel.  Ceres rejects more than half of its steps on
// this problem, and the steps a run of rejections would walk through (the radius halved, quartered, ...) are prepared side by side
// whenever a linearisation is solved (EDS_NCAND of them, eds_solver12_coop.hpp).  G teams of K CUs evaluate prepared steps k .. k + G - 1 at
// the same time; every workgroup then reads the BLOCK COSTS of all G x K members (one number per residual block: all a rejection
// needs), replays Solver12's decisions over them in order (coop12_decide_head / coop12_walk / coop12_take), and only for the step that
// was accepted fetches the 157 sums per block of that group's K members for the linearisation.  (Collecting ALL groups' sums in one
// round of granules — no second trip — was measured slower: 111.6 against 106.8 us for one alignment, 150 against 126 with four
// residual blocks: G x K x 314 polled granules per workgroup cost more than the trip they save.)  Same decisions, radii and counters
// as one team alone; the sums are added in the same member order (inside a member the wavefronts' tiles meet in LDS by fp64 atomics, so
// the last bits vary from run to run — with and without groups).  12 evaluations become 6-7 rounds.  The residuals of the accepted point stay in
// the registers of the group that evaluated it, which writes them at the end.
// Entry e of the sums of the TEAM members whose mailboxes start at gb, added in member order: all granules of up to four members in
// flight at once, the late ones asked for again together until every one carries this round's tag (or the time-out strikes).
template <int TEAM>
__device__ __forceinline__ double team12_collect(const unsigned long long* __restrict__ gb, const int e, const unsigned tag, const unsigned long long t_start, int* timeout) {
    double tot = 0.0;
    constexpr int GM = TEAM < 4 ? TEAM : (TEAM == 16 ? 8 : 4);
#pragma unroll
    for (int m0 = 0; m0 < TEAM; m0 += GM) {
        unsigned long long v[2 * GM];
#pragma unroll
        for (int k = 0; k < 2 * GM; ++k)
            v[k] = __hip_atomic_load(gb + (size_t)(m0 + (k >> 1)) * EDS_TEAM12_GRANULES + 2 * e + (k & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            bool late = false;
#pragma unroll
            for (int k = 0; k < 2 * GM; ++k) late |= (unsigned)(v[k] >> 32) != tag;
            if (!late) break;
            if (__builtin_amdgcn_s_memrealtime() - t_start > EDS_TEAM_TIMEOUT_TICKS) { *timeout = 1; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int k = 0; k < 2 * GM; ++k)
                if ((unsigned)(v[k] >> 32) != tag)
                    v[k] = __hip_atomic_load(gb + (size_t)(m0 + (k >> 1)) * EDS_TEAM12_GRANULES + 2 * e + (k & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int m = 0; m < GM; ++m)
            tot += __longlong_as_double((long long)(((v[2 * m + 1] & 0xffffffffull) << 32) | (v[2 * m] & 0xffffffffull)));
    }
    return tot;
}

template <int SAMPLING, int NTHR, int CAP, bool NC, int TEAM, int QUAD, int GROUPS = 1>
__global__ __launch_bounds__(NTHR, 2) void eds_fused12_kernel(EdsArrays A, const EdsFusedIn* __restrict__ in,
                                                              EdsFused12Out* __restrict__ out, int first, int iters, int loss_type,
                                                              double loss_a, double ftol, double gtol, double ptol, int nb,
                                                              unsigned long long* __restrict__ mail, int* __restrict__ ticket, unsigned ticket_base, unsigned epoch) {
    const int tid = threadIdx.x;
    constexpr int nthr = NTHR;
    static_assert(TEAM == 1 || !NC, "the NC residual needs a second exchange (block norm of the sampled brightness): no teams");
    static_assert(GROUPS == 1 || (TEAM > 1 && GROUPS <= EDS_NCAND), "candidate groups: teams only, at most one per prepared step");
    constexpr int VTEAM = TEAM * GROUPS;              // workgroups per alignment
    // FULL (round 6): ONE alignment per CU with a patch-cache slot for EVERY point — 512 threads, CAP = 2 000 patches
    // (128 000 B of rows + 8 000 B of keys of the CU's 163 840).  What is left has to hold the solver: ONE residual block (the
    // reference problem of the bench; launches with more blocks take the other shapes), and the rows [J | r] of a wavefront go
    // through the matrix core's staging area 16 at a time instead of 64 (four rounds of: 16 lanes store their rows, the wavefront
    // reads them back as four operand pairs).
    constexpr bool FULL = NTHR == 512 && CAP == 2000;
    // HALF (round 6, second attempt): the paired shape — two 256-thread alignments per CU, one's solver phase under the other's point
    // phase — with the same slimming, which leaves each of the two a cache of 736 patches (37 % of 2 000 points: what two workgroups per CU leave of its 160 KB).
    constexpr bool HALF = NTHR == 256 && CAP == 736;
    constexpr bool SLIM = FULL || HALF;
    static_assert(!SLIM || (TEAM == 1 && !NC && GROUPS == 1), "the slim shapes: one workgroup per alignment, plain residual");
    constexpr int MAXB = SLIM ? 1 : EDS_DEV_MAX_BLOCKS;      // residual blocks this instantiation can hold
    constexpr int SROWS = SLIM ? 16 : 64;                   // rows of a wavefront staged at a time
    __shared__ int s_ticket, s_timeout;
    __shared__ double s_gcost[GROUPS][MAXB];      // GROUPS > 1: ||r_b||^2 of every group's candidate (summed over its members)
    __shared__ int s_gacc, s_kacc, s_linmode;                   // ... the group / prepared step that was accepted this round (-1: none), how
    int team_slot = blockIdx.x, member = 0, group = 0, res_owner = 0;
    if (TEAM > 1) {
        if (tid == 0) { s_ticket = (int)((unsigned)atomicAdd(ticket, 1) - ticket_base); s_timeout = 0; }   // the counter is never reset: the host knows how many tickets earlier launches took
        __syncthreads();
        if ((unsigned)s_ticket >= gridDim.x) return;      // counters out of step: see eds_fused6_kernel
        team_slot = s_ticket / VTEAM; member = s_ticket % TEAM; group = (s_ticket % VTEAM) / TEAM;
    }
    const int slot = first + team_slot;
    if (tid == 0 && member == 0 && group == 0) out[slot].t_begin = __builtin_amdgcn_s_memrealtime();
    if (nb > MAXB) {                    // (the launcher never asks for it: eds_ref12_force_feasible / the rule; a failed solve, not a wrong one)
        if (tid == 0) { out[slot].failed = 1; out[slot].termination = edss::TERM_FAILURE; out[slot].t_end = __builtin_amdgcn_s_memrealtime(); }
        return;
    }
    unsigned pass_no = 0;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NTAP = (SAMPLING == 0) ? 16 : 4;
    __shared__ edss::Solver12 sv;
    __shared__ edss::Sums12T<MAXB> sums;
    __shared__ edsc::Work12 work;
    __shared__ double s_pb[EDS_NCAND][EDS_POSE_STRIDE];      // pose blocks of the prepared steps (eds_solver12_coop.hpp); s_pb[s_k] is being evaluated
    __shared__ edsc::Cand12 s_cand[EDS_NCAND];
    __shared__ edsc::Step12 s_step[EDS_NCAND];
    __shared__ int s_k, s_head, s_walk;
    __shared__ float s_stage[(NTHR / 64)][SROWS * 17];
    __shared__ int s_state, s_accept;
    __shared__ __attribute__((aligned(16))) float s_patch[NTAP][CAP];
    static_assert(!QUAD || (SAMPLING == 0 && CAP % 4 == 0), "quad gather: bicubic, whole quads cached");
    __shared__ int s_cell[CAP];
    // The batch shape (two 256-thread workgroups per CU) gathers EVERY patch every evaluation: its 20 KB could cache 320 of 2 000 points
    // (3 % of the line fills at the pose-only kernel's 21 % hit rate), and probing, selecting and writing back for all points cost more
    // vector instructions than that returns (profiles/r04_sq_counters.txt: 523 per point-evaluation, a quarter of the wavefront-cycles wait
    // for an issue slot).  The quad gather of that shape therefore skips the cache (round 4); the lane gather and the 512-thread shapes keep it.
    constexpr bool QCACHE = !(QUAD != 0 && NTHR == 256) || HALF;
    // Candidate residuals of an evaluation: in LDS for the batch shape (two 256-thread workgroups per CU: 8 KB fit its 80 KB), copied to the
    // residual plane when the candidate is accepted — round 3 wrote every evaluation's candidates to the mhat plane and read them back on
    // acceptance (0.5 GB of writes per 4 096-alignment launch: profiles/r03_summary.md WRITE_SIZE).  Points beyond the buffer, and the
    // 512-thread shapes (whose LDS is full), keep the plane.
    constexpr int RC_CAP = (NTHR == 256) ? 2048 : 0;
    __shared__ float s_rc[RC_CAP > 0 ? RC_CAP : 1];
    __shared__ double s_G[MAXB * 36];
    __shared__ double s_nc[NC ? MAXB : 1][8];    // NC residual: per block 1/||E||, then sum_j E_j J'_j / ||E||^3

    const double* __restrict__ gpb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const double* __restrict__ Gg = A.G + (size_t)slot * EDS_MAX_BLOCKS * 36;
    const int N = (int)gpb[EDS_PB_N];
    const int ne = N / nb;
    const size_t base = (size_t)slot * A.Np;
    // this workgroup's slice of the points [lo, hi): everything for TEAM == 1; whole wavefront tiles otherwise
    const int chunk = TEAM > 1 ? (((N + TEAM - 1) / TEAM + 63) & ~63) : N;
    const int lo = TEAM > 1 ? (member * chunk < N ? member * chunk : N) : 0;
    const int hi = TEAM > 1 ? (lo + chunk < N ? lo + chunk : N) : N;
    const int fslot = (int)gpb[EDS_PB_FRAME];          // the slot whose frame storage is sampled (its own unless shared)
    const FrameView frame = make_frame_view(A.frame, fslot, A.H, A.W, A.Hp, A.Wp, 1);    // persistent kernels: tiled frames only (eds_fused_solve)

    if (wave == 0) {
        for (int k = lane; k < nb * 36; k += 64) s_G[k] = Gg[k];
        if (lane == 0) {
            const EdsFusedIn& I = in[slot];
            for (int w = 0; w < EDS_NCAND; ++w)
                for (int i = 0; i < 4; ++i) s_pb[w][EDS_PB_K + i] = gpb[EDS_PB_K + i];
            sv.init(iters, loss_type, loss_a, ftol, gtol, ptol, I.p, I.q, I.v);
            s_k = 0; s_head = 0; s_walk = edsc::W_EVAL;
            sv.skip_final = 1;              // accepted-point residuals are kept in the residual plane as the solve goes
            sums.nb = nb;
            s_state = 0; s_accept = 0;
#ifdef EDS_FUSED_STAMPS
            for (int k = 0; k < 8; ++k) work.st[k] = 0;
#endif
        }
        EDS_WSYNC();
        edsc::coop_fill_pose_block(sv.cp, sv.cq, sv.cv, s_G, nb, s_pb[0], lane);
    }
    static_assert(NTHR / 64 >= EDS_NCAND, "one wavefront per prepared step");
    for (int i = tid; i < CAP; i += nthr) s_cell[i] = 0x7fffffff;
    for (int k = tid; k < (NTHR / 64) * SROWS * 17; k += nthr) (&s_stage[0][0])[k] = 0.0f;    // columns 13..15 stay zero for good
    for (int k = tid; k < (int)(sizeof(sums) / sizeof(double)); k += nthr)
        if (k > 0) reinterpret_cast<double*>(&sums)[k] = 0.0;                               // word 0 holds nb
    __syncthreads();

    float* const stage = &s_stage[wave][0];
    auto flush = [&](const acc4d& C, int b) {       // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 r
        const int col = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r;
            const double v = C[r];
            if (row < 12 && col < 12) unsafeAtomicAdd(&sums.H[b][12 * row + col], v);
            else if (row < 12 && col == 12) unsafeAtomicAdd(&sums.g[b][row], v);
            else if (row == 12 && col == 12) unsafeAtomicAdd(&sums.s[b], v);
        }
    };

#ifdef EDS_FUSED_STAMPS
    unsigned long long st_acc[3] = {0, 0, 0}, st_t = __builtin_readcyclecounter();
    int st_n = 0, st_prop = 0;
    if (tid < 8) for (int w_ = 0; w_ < EDS_NCAND; ++w_) s_step[w_].pst[tid] = 0;
#define EDS12_STAMP(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc[k] += n_ - st_t; st_t = n_; pst_t = n_; } while (0)
    unsigned long long pst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pst_t = st_t;        // the point phase in five pieces (wavefront 0)
#define EDS12_PSTAMP(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); pst[k] += n_ - pst_t; pst_t = n_; } while (0)
#else
#define EDS12_STAMP(k) do { } while (0)
#define EDS12_PSTAMP(k) do { } while (0)
#endif
    float rkeep[FULL ? 4 : 2] = {};                          // candidate residuals of this lane's first two points (all sweeps write them; FULL: of all four)
    float racc[2] = {0.0f, 0.0f};                            // GROUPS > 1: ... of the accepted point, in the group that evaluated it
    for (;;) {
        EDS12_PSTAMP(5);                                     // residual copy of an accepted evaluation, loop back
        // the pose block under evaluation (candidate groups: group g takes prepared step s_k + g; the first round has one point, the start)
        const int k_eval = GROUPS > 1 ? (sv.started ? (s_k + group < EDS_NCAND ? s_k + group : EDS_NCAND - 1) : s_k) : s_k;
        const double* __restrict__ s_pose = s_pb[k_eval];
        PoseF ps;
        load_pose(s_pose, ps);
        float vf[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) vf[k] = uniformf((float)s_pose[EDS_PB_V + k]);
        const bool one_block = nb == 1;
        float inv_n1 = 0.0f, gv1[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (one_block) {
            const double* bk = s_pose + EDS_PB_BLK;
            inv_n1 = uniformf((float)bk[0]);
#pragma unroll
            for (int k = 0; k < 6; ++k) gv1[k] = uniformf((float)bk[1 + k]);
        }
        acc4d C = {0, 0, 0, 0}, C2 = {0, 0, 0, 0};
        int cb = -1;                    // residual block the tile currently belongs to (wave-uniform)
        EDS12_PSTAMP(6);                // pose block -> registers
        // One sweep over the points.  MODE 0: the plain residual (PhotometricError).  PhotometricErrorNC needs the block norm of
        // the sampled brightness before any row can be formed, so it sweeps twice: MODE 1 samples, forms the un-weighted pose
        // columns J' = -dE and accumulates [J' | E]^T [J' | E] (which holds sum E^2 and sum E J') while stashing E and J' in
        // seven planes of the otherwise unused Jacobian buffer; MODE 2 re-reads the stash (no frame access) and emits the rows
        //   w (J'/||E|| - E sum_j E_j J'_j/||E||^3),  velocity columns as for the plain residual,  r = w (m/||m|| - E/||E||).
        auto sweep = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            const size_t jplane = (size_t)A.B * A.Np;
        // (batch shape on the strips: the NEXT step's point constants are asked for while this step computes — a step was two trips to
        // memory in a row, constants then rows, on two wavefronts per SIMD)
        constexpr bool PREF = EDS_REF12_PREFETCH && QUAD == 2 && NTHR == 256 && MODE == 0;
        float pc[2][9];
        auto fetch_consts = [&](int j0_, float (&dst)[2][9]) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int i = j0_ + jj * nthr + tid;
                const float* __restrict__ c = A.kf + base + (i < hi ? i : 0);
                const size_t pl = A.kf_plane;
                dst[jj][0] = c[EDS_KF_X * pl]; dst[jj][1] = c[EDS_KF_Y * pl]; dst[jj][2] = c[EDS_KF_RHO * pl]; dst[jj][3] = c[EDS_KF_W * pl];
                dst[jj][4] = c[EDS_KF_GX * pl]; dst[jj][5] = c[EDS_KF_GY * pl]; dst[jj][6] = c[EDS_KF_F0X * pl]; dst[jj][7] = c[EDS_KF_F0Y * pl];
                dst[jj][8] = c[EDS_KF_CELL0 * pl];
            }
        };
        if (PREF) fetch_consts(lo, pc);
        // one step of the sweep: 2 x nthr points from j0 on (FULL: its 2 000 points are exactly two steps, and the candidate residuals of
        // all four points of a lane stay in registers — picked by the wave-uniform step index, two selects per point)
        auto sweep_step = [&](const int j0) {
            // phase A: two points per lane: constants from HBM/L2, projection, cache probe, gathers in flight
            float cc[2][9];
            if (PREF) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int k = 0; k < 9; ++k) cc[jj][k] = pc[jj][k];
                fetch_consts(j0 + 2 * nthr, pc);
            }
            PointKf kf[2];
            float kw[2], kgx[2], kgy[2];
            PointGeom pg[2];
            float tap[2][NTAP];
            bool miss[2];
            int org[2];                                                 // QUAD: packed patch origin | miss flag, per point
            float4 ra[2][4], rb[2][4];                                  // QUAD: this lane's row of the quad's four patches
            const int jr = lane & 3;
            const float* __restrict__ tiles = A.frame + (size_t)fslot * A.Hp * A.Wp;
            const unsigned copy_bytes = (unsigned)(eds_strips_copy_elems(A.Hp, A.Wp) * 4);
            const char* __restrict__ sbase = reinterpret_cast<const char*>(A.strips) + (size_t)(unsigned)fslot * ((size_t)(2 * A.strip_phases) * copy_bytes);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int i = j0 + jj * nthr + tid;
                const bool valid = i < hi;
                // a wavefront whose 64 points of this half all lie beyond the slice skips the half altogether (wave-uniform): a lone
                // alignment on 8 CUs has 250 points for 512 threads x 2 points — without this every lane dragged a second, silent
                // point through projection, gather and spline
                if (!QUAD && edsc::uniform_int(j0 + jj * nthr + wave * 64) >= hi) { miss[jj] = false; kw[jj] = 0.0f; continue; }
                const size_t o = base + (valid ? i : 0);
                const float* __restrict__ c = A.kf + o;                 // one base pointer, nine planes (eds_layout.hpp EDS_KF_*)
                const size_t pl = A.kf_plane;
                if (PREF) {
                    kf[jj].x = cc[jj][0]; kf[jj].y = cc[jj][1]; kf[jj].rhop = cc[jj][2] + 1e-5f;
                    kw[jj] = valid ? cc[jj][3] : 0.0f;
                    kgx[jj] = cc[jj][4]; kgy[jj] = cc[jj][5];
                } else {
                    kf[jj].x = c[EDS_KF_X * pl]; kf[jj].y = c[EDS_KF_Y * pl]; kf[jj].rhop = c[EDS_KF_RHO * pl] + 1e-5f;
                    kw[jj] = valid ? c[EDS_KF_W * pl] : 0.0f;
                    kgx[jj] = c[EDS_KF_GX * pl]; kgy[jj] = c[EDS_KF_GY * pl];
                }
                miss[jj] = false;
#ifdef EDS_FUSED_STAMPS
                if (jj == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); EDS12_PSTAMP(7); }     // (diagnostic builds wait for the constants here)
#endif
                if (MODE == 2) continue;                                // rows come from the stash: no projection, no gather
                if (PREF) { kf[jj].f0x = cc[jj][6]; kf[jj].f0y = cc[jj][7]; kf[jj].cell0 = __float_as_int(cc[jj][8]); }
                else { kf[jj].f0x = c[EDS_KF_F0X * pl]; kf[jj].f0y = c[EDS_KF_F0Y * pl]; kf[jj].cell0 = __float_as_int(c[EDS_KF_CELL0 * pl]); }
                project_point(ps, kf[jj], pg[jj]);
                const int li = i - lo;                                  // cache index: local to this workgroup's slice
                const bool cached = li < CAP;
                const int key = (pg[jj].r0 << 16) ^ (pg[jj].c0 & 0xffff);
                miss[jj] = !QCACHE || !(cached && s_cell[li] == key);
                if (QUAD) {
                    if (QCACHE && miss[jj] && cached) s_cell[li] = key;
                    if (QUAD == 2) {            // strips: the byte offset of the patch's first row (cached: offset 0, as below)
                        const int ra_ = clampi(pg[jj].r0, -2, frame.H) + (EDS_FRAME_MARGIN - 1), ca_ = clampi(pg[jj].c0, -2, frame.W) + (EDS_FRAME_MARGIN - 1);
                        org[jj] = miss[jj] ? (int)(eds_strips_row_offset(ra_, ca_, A.Hp, copy_bytes, A.strip_phases) | 0x80000000u) : 0;
                        continue;
                    }
                    org[jj] = miss[jj] ? (pack_origin(frame, pg[jj].r0, pg[jj].c0) | (int)0x80000000) : 0;     // cached: origin 0 — its (unused) row loads fall on the first line of the allocation: no branch around the loads (+3 %)
                    continue;
                }
                if (miss[jj]) {
                    if (SAMPLING == 0) load_patch16(frame, pg[jj].r0, pg[jj].c0, reinterpret_cast<float(&)[16]>(tap[jj]));
                    else load_patch4(frame, pg[jj].r0, pg[jj].c0, reinterpret_cast<float(&)[4]>(tap[jj]));
                    if (cached) s_cell[li] = key;
                } else {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) tap[jj][t] = s_patch[t][li];
                }
            }
            if (QUAD && MODE != 2) {                                    // the packed origins go round the quad; every lane puts its ROW of
#pragma unroll                                                          // each missing patch in flight
                for (int jj = 0; jj < 2; ++jj) {
                    const int o0 = quad_bcast_i<0>(org[jj]), o1 = quad_bcast_i<1>(org[jj]), o2 = quad_bcast_i<2>(org[jj]), o3 = quad_bcast_i<3>(org[jj]);
                    const int oq[4] = {o0, o1, o2, o3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (QUAD == 2) {
                            const float4u v = *reinterpret_cast<const float4u*>(sbase + (((unsigned)oq[q] & 0x7fffffffu) + 32u * (unsigned)jr));
                            ra[jj][q] = make_float4(v.x, v.y, v.z, v.w);
                        } else {
                            load_patch_row(tiles, frame.TW, oq[q], jr, ra[jj][q], rb[jj][q]);
                        }
                    }
                }
            }
            EDS12_PSTAMP(0);                                            // constants, projection, probe, gathers issued
            // phase B: residual + 1x12 row (closed forms of SURVEY §8a), rows through LDS into the MFMA
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int i = j0 + jj * nthr + tid;
                const bool valid = i < hi;
                const int i_first = j0 + jj * nthr + wave * 64;        // this wavefront's 64 consecutive points
                if (!QUAD && edsc::uniform_int(i_first) >= hi) continue;
                const int i_last = (i_first + 63 < hi) ? i_first + 63 : hi - 1;
                // (one residual block — nb is a kernel argument, the branch is scalar: no integer divisions for the block index, and the block's
                // constants were narrowed once per evaluation, in front of the sweep)
                const int b_lo = one_block ? 0 : edsc::uniform_int(block_of(i_first < hi ? i_first : 0, ne, nb));
                const int b_hi = one_block ? 0 : edsc::uniform_int(block_of(i_last > 0 ? i_last : 0, ne, nb));
                int myb = b_lo;
                float inv_n, gv[6];
                if (one_block) {
                    inv_n = inv_n1;
#pragma unroll
                    for (int k = 0; k < 6; ++k) gv[k] = gv1[k];
                } else if (b_lo == b_hi) {                           // the usual case: block constants are wave-uniform
                    const double* bk = s_pose + EDS_PB_BLK + EDS_PB_BLK_STRIDE * b_lo;
                    inv_n = uniformf((float)bk[0]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) gv[k] = uniformf((float)bk[1 + k]);
                } else {
                    myb = block_of(valid ? i : 0, ne, nb);
                    const double* bk = s_pose + EDS_PB_BLK + EDS_PB_BLK_STRIDE * myb;
                    inv_n = (float)bk[0];
#pragma unroll
                    for (int k = 0; k < 6; ++k) gv[k] = (float)bk[1 + k];
                }
                const float w = kw[jj];                              // 0 for out-of-range lanes: their rows vanish
                float x[13];
                float* __restrict__ st7 = A.J + base + (valid ? i : 0);      // NC stash: planes 0..5 J', plane 6 E
                if (MODE != 2) {
                    float E, Er, Ec;
                    if (QUAD) {
                        // branch-free: the cached row and the gathered row are both formed, a bit mask picks one; the row goes (back) to
                        // the cache ([point][row] units of 16 bytes, XOR-swizzled by the quad index: patch_unit), its spline runs here,
                        // the transposes return the four row results to the lane that owns the point
                        const int qli = (i & ~3) - lo;                   // first point of this quad, local index (quads are cached whole)
                        const bool qcached = qli < CAP;
                        float* __restrict__ cache = &s_patch[0][0];
                        const int o0 = quad_bcast_i<0>(org[jj]), o1 = quad_bcast_i<1>(org[jj]), o2 = quad_bcast_i<2>(org[jj]), o3 = quad_bcast_i<3>(org[jj]);
                        const int oq[4] = {o0, o1, o2, o3};
                        const float x0 = quad_bcast_f<0>(pg[jj].ax), x1 = quad_bcast_f<1>(pg[jj].ax), x2 = quad_bcast_f<2>(pg[jj].ax), x3 = quad_bcast_f<3>(pg[jj].ax);
                        const float xq[4] = {x0, x1, x2, x3};
                        float f[4], d[4];
                        float tq[4][4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float (&t)[4] = tq[q];
                            if (QUAD == 2) { t[0] = ra[jj][q].x; t[1] = ra[jj][q].y; t[2] = ra[jj][q].z; t[3] = ra[jj][q].w; }
                            else shift_patch_row(ra[jj][q], rb[jj][q], oq[q], t);
                            if (QCACHE) {
                                float* unit = cache + 4 * patch_unit(qcached ? qli + q : q, jr);
                                const float4 c = *reinterpret_cast<const float4*>(unit);
                                const int m = oq[q] >> 31;               // all ones: gathered this pass
                                t[0] = flag_select(m != 0, t[0], c.x); t[1] = flag_select(m != 0, t[1], c.y); t[2] = flag_select(m != 0, t[2], c.z); t[3] = flag_select(m != 0, t[3], c.w);
                                if (qcached) *reinterpret_cast<float4*>(unit) = make_float4(t[0], t[1], t[2], t[3]);
                            }
                        }
                        // the row splines of two patches per packed instruction, then the column spline of {value, column derivative} as one
                        // more pair (hermite_pair: what the pose-only kernel's point phase does since round 3) — 42 packed instructions where six
                        // scalar splines took ~100 (round 4: this kernel issued 523 vector instructions per point-evaluation)
                        {
                            const f2 x01 = {xq[0], xq[1]}, x23 = {xq[2], xq[3]};
                            f2 f01, d01, f23, d23;
                            hermite_pair((f2){tq[0][0], tq[1][0]}, (f2){tq[0][1], tq[1][1]}, (f2){tq[0][2], tq[1][2]}, (f2){tq[0][3], tq[1][3]}, x01, 0.5f * x01, 3.0f * x01, f01, d01);
                            hermite_pair((f2){tq[2][0], tq[3][0]}, (f2){tq[2][1], tq[3][1]}, (f2){tq[2][2], tq[3][2]}, (f2){tq[2][3], tq[3][3]}, x23, 0.5f * x23, 3.0f * x23, f23, d23);
                            f[0] = f01.x; f[1] = f01.y; f[2] = f23.x; f[3] = f23.y;
                            d[0] = d01.x; d[1] = d01.y; d[2] = d23.x; d[3] = d23.y;
                        }
                        quad_transpose(f, lane);
                        quad_transpose(d, lane);
                        {
                            const f2 y2 = (f2)(pg[jj].ay);
                            f2 EEc, dE;
                            hermite_pair((f2){f[0], d[0]}, (f2){f[1], d[1]}, (f2){f[2], d[2]}, (f2){f[3], d[3]}, y2, 0.5f * y2, 3.0f * y2, EEc, dE);
                            E = EEc.x; Ec = EEc.y; Er = dE.x;
                        }
                    } else {
                    if (miss[jj] && i - lo < CAP) {
#pragma unroll
                        for (int t = 0; t < NTAP; ++t) s_patch[t][i - lo] = tap[jj][t];
                    }
                    if (SAMPLING == 0) bicubic_patch(reinterpret_cast<float(&)[16]>(tap[jj]), pg[jj].ay, pg[jj].ax, E, Er, Ec);
                    else bilinear_patch(reinterpret_cast<float(&)[4]>(tap[jj]), pg[jj].ay, pg[jj].ax, E, Er, Ec);
                    }
                    EDS12_PSTAMP(1);                                    // taps arrived, spline done
                    PointProj pp;
                    finish_point(ps, pg[jj], E, Er, Ec, pp);
                    const float wp = MODE == 1 ? (valid ? 1.0f : 0.0f) : w;  // NC: pose columns un-weighted until the norm is known
                    x[0] = -wp * pp.g0; x[1] = -wp * pp.g1; x[2] = -wp * pp.g2;
                    const float rx = pp.Px - ps.t[0], ry = pp.Py - ps.t[1], rz = pp.Pz - ps.t[2];      // R X = P - t
                    const float w2 = -2.0f * wp;
                    x[3] = w2 * (ry * pp.g2 - rz * pp.g1);
                    x[4] = w2 * (rz * pp.g0 - rx * pp.g2);
                    x[5] = w2 * (rx * pp.g1 - ry * pp.g0);
                    x[12] = pp.E;
                }
                if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) x[6 + k] = 0.0f;
                    x[12] = valid ? x[12] : 0.0f;
                    if (valid) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) st7[k * jplane] = x[k];
                        st7[6 * jplane] = x[12];
                    }
                } else {
                    float ka[6];
                    model_row(kf[jj].x, kf[jj].y, kf[jj].rhop - 1e-5f, kgx[jj], kgy[jj], ka);
                    float m = 0.0f;
#pragma unroll
                    for (int k = 0; k < 6; ++k) m += ka[k] * vf[k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) x[6 + k] = w * (ka[k] * inv_n - m * gv[k]);         // projector applied by the solver
                    if (MODE == 0) {
                        x[12] = w * (m * inv_n - x[12]);
                    } else {                                         // NC rows from the stash and the block statistics
                        const double* nk = &s_nc[b_lo == b_hi ? b_lo : myb][0];
                        const float inv_e = (float)nk[0], E = st7[6 * jplane];
#pragma unroll
                        for (int k = 0; k < 6; ++k) x[k] = w * (st7[k * jplane] * inv_e - E * (float)nk[1 + k]);
                        x[12] = w * (m * inv_n - E * inv_e);
                    }
                    if (valid && GROUPS == 1 && !FULL) {              // candidate residual (candidate groups: registers only — the groups share the plane)
                        if (RC_CAP > 0 && i - lo < RC_CAP) s_rc[i - lo] = x[12]; else A.mhat[base + i] = x[12];
                    }
                    if (FULL) {                                       // (all four of a lane: no plane, no LDS)
                        const bool second = j0 != lo;
                        rkeep[jj] = second ? rkeep[jj] : x[12];
                        rkeep[2 + jj] = second ? x[12] : rkeep[2 + jj];
                    }
                    else if (j0 == lo) rkeep[jj] = x[12];             // (the first two of a lane also stay in registers: see the accept copy)
                }
                EDS12_PSTAMP(2);                                        // row formed, candidate residual stored
                if (i_first < hi) {
                    for (int b = b_lo; b <= b_hi; ++b) {
                        if (b != cb) {
                            if (cb >= 0) flush(C + C2, cb);
                            C = acc4d{0, 0, 0, 0}; C2 = acc4d{0, 0, 0, 0};
                            cb = b;
                        }
                        // (one block, plain residual: a lane without a point carries w = 0 and the constants of the slot's first point, so its row is
                        // all zeros already — no select in front of the 13 staging stores)
                        const bool on = (MODE == 0 && one_block) ? true : (valid && myb == b);
                        if constexpr (SROWS == 64) {
#pragma unroll
                            for (int c = 0; c < 13; ++c) stage[lane * 17 + c] = on ? x[c] : 0.0f;
                            EDS_WSYNC();
#pragma unroll
                            for (int mm = 0; mm < 16; mm += 2) {
                                const float a0 = stage[(4 * mm + (lane >> 4)) * 17 + (lane & 15)];
                                const float a1 = stage[(4 * mm + 4 + (lane >> 4)) * 17 + (lane & 15)];
                                C = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a0, (double)a0, C, 0, 0, 0);
                                C2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a1, (double)a1, C2, 0, 0, 0);
                            }
                            EDS_WSYNC();
                        } else {
                            // SROWS rows at a time (the full-cache shape: its staging area is a quarter of a wavefront's rows): the lanes of
                            // round q store, every lane reads the round's SROWS / 4 operand quadruples; same products, same order of the sums
#pragma unroll
                            for (int q = 0; q < 64 / SROWS; ++q) {
                                if ((lane / SROWS) == q) {
#pragma unroll
                                    for (int c = 0; c < 13; ++c) stage[(lane % SROWS) * 17 + c] = on ? x[c] : 0.0f;
                                }
                                EDS_WSYNC();
#pragma unroll
                                for (int mm = 0; mm < SROWS / 4; mm += 2) {
                                    const float a0 = stage[(4 * mm + (lane >> 4)) * 17 + (lane & 15)];
                                    const float a1 = stage[(4 * mm + 4 + (lane >> 4)) * 17 + (lane & 15)];
                                    C = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a0, (double)a0, C, 0, 0, 0);
                                    C2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a1, (double)a1, C2, 0, 0, 0);
                                }
                                EDS_WSYNC();
                            }
                        }
                    }
                }
            }
        };
        for (int j0 = lo; j0 < hi; j0 += 2 * nthr) sweep_step(j0);
        EDS12_PSTAMP(3);                                                // staging + matrix core
        if (cb >= 0) flush(C + C2, cb);
        C = acc4d{0, 0, 0, 0}; C2 = acc4d{0, 0, 0, 0}; cb = -1;
        EDS12_PSTAMP(4);                                                // tile added to the sums in LDS
        };
        if (!NC) {
            sweep(std::integral_constant<int, 0>());
        } else {
            sweep(std::integral_constant<int, 1>());
            __syncthreads();
            if (tid < nb) {                     // ||E||_block and sum_j E_j J'_j / ||E||^3 from the tile of sweep 1 (PhotometricErrorNC.hpp:151-186)
                const double SE = 1e-3 + sums.s[tid];
                const double inv = 1.0 / sqrt(SE);
                s_nc[tid][0] = inv;
                for (int c = 0; c < 6; ++c) s_nc[tid][1 + c] = sums.g[tid][c] * inv / SE;
            }
            __syncthreads();
            for (int k = 1 + tid; k < (int)(sizeof(sums) / sizeof(double)); k += nthr) reinterpret_cast<double*>(&sums)[k] = 0.0;
            __syncthreads();
            sweep(std::integral_constant<int, 2>());
        }
        EDS12_STAMP(0);
        __syncthreads();
        if (TEAM > 1) {
            // every thread publishes "its" entries of this member's sums, then collects the same entries of all members and leaves
            // their total (added in member order: identical on every member) in place
            const unsigned tag = (epoch << 8) | ((pass_no & 0x7f) + 1);
            unsigned long long* mb = mail + ((size_t)team_slot * 2 + (pass_no & 1)) * ((size_t)VTEAM * EDS_TEAM12_GRANULES);
            const int me = group * TEAM + member;
            const int nval = nb * EDS_TEAM12_VALUES;
            auto entry = [&](int e) -> double* {
                const int b = e / EDS_TEAM12_VALUES, r = e - b * EDS_TEAM12_VALUES;
                return r == 0 ? &sums.s[b] : (r <= 144 ? &sums.H[b][r - 1] : &sums.g[b][r - 145]);
            };
            for (int e = tid; e < nval; e += nthr) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(*entry(e));
                unsigned long long* g = mb + (size_t)me * EDS_TEAM12_GRANULES + 2 * e;
                __hip_atomic_store(g, ((unsigned long long)tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            // sums entry e of the K members of group `grp`, added in member order (polls until every granule carries this round's tag)
            auto collect = [&](int grp, int e, double& tot) {
                tot = team12_collect<TEAM>(mb + (size_t)grp * TEAM * EDS_TEAM12_GRANULES, e, tag, t_start, &s_timeout);
            };
            if constexpr (GROUPS > 1) {
                // ---- phase 1: the block costs of every group -----------------------------------------------------------------------------
                for (int idx = tid; idx < GROUPS * nb; idx += nthr) {
                    const int g = idx / nb, b = idx - g * nb;
                    double tot;
                    collect(g, b * EDS_TEAM12_VALUES, tot);
                    s_gcost[g][b] = tot;
                }
                ++pass_no;
                __syncthreads();
                if (s_timeout) {
                    if (tid == 0) { sv.termination = edss::TERM_FAILURE; sv.num_unsuccessful = -2; }
                    break;
                }
                EDS12_STAMP(1);
                // ---- Solver12's decisions replayed over the G candidates, in order (wavefront 0) ---------------------------------------------
                if (wave == 0) {
                    const int kbase = edsc::uniform_int(s_k);
                    int kcur = kbase, gacc = -1, walk = edsc::W_EVAL, mode = edsc::M_RETURN, head = 0;
                    for (;;) {
                        const int g = edsc::uniform_int(sv.started) ? kcur - kbase : 0;
                        mode = edsc::coop12_decide_head(sv, nb, s_gcost[g], work, lane);
                        if (mode == edsc::M_RETURN) { walk = edsc::W_RETURN; break; }
                        if (mode != edsc::M_ADVANCE) { gacc = g; break; }                 // M_LIN_ITER0 / M_LIN_ACCEPT: this one becomes the accepted point
                        const edsc::Walk12 wk = edsc::coop12_walk(sv, s_cand, kcur + 1, 0, lane);       // unsuccessful: one notch down the radius sequence
                        walk = wk.walk; head = wk.head;
                        if (wk.walk != edsc::W_EVAL) { kcur = wk.k; break; }             // the solve ended, or no step is prepared for this radius
                        edsc::coop12_take(sv, s_cand[wk.k], lane);
                        kcur = wk.k;
                        if (kcur - kbase >= GROUPS) break;                               // prepared, but not evaluated in this round: the next round starts there
                    }
                    if (lane == 0) {
                        s_gacc = gacc; s_kacc = kcur; s_linmode = mode;
                        s_accept = gacc >= 0 ? 1 : 0;
                        s_walk = walk; s_k = kcur; s_head = head;
                    }
                }
                __syncthreads();
                // ---- phase 2: the full sums of the accepted group, the linearisation, the walk on from there ------------------------------
                const int gacc = s_gacc;
                if (gacc >= 0) {
                    for (int e = tid; e < nval; e += nthr) { double tot; collect(gacc, e, tot); *entry(e) = tot; }
                    __syncthreads();
                    if (s_timeout) {            // the accepted group's 157 sums per block can time out as well: never linearise on a partial total
                        if (tid == 0) { sv.termination = edss::TERM_FAILURE; sv.num_unsuccessful = -2; }       // (the host re-runs the range without teams)
                        break;
                    }
                    if (wave == 0) {
                        const int mode = edsc::coop12_linearise(sv, sums, work, s_pb[edsc::uniform_int(s_kacc)], edsc::uniform_int(s_linmode), lane);
                        edsc::Walk12 wk{edsc::W_RETURN, 0, 0};
                        if (mode != edsc::M_RETURN) wk = edsc::coop12_walk(sv, s_cand, EDS_NCAND, 0, lane);      // fresh linearisation: every prepared step is stale
                        if (lane == 0) {
                            s_walk = wk.walk;
                            if (mode != edsc::M_RETURN) { s_k = wk.k; s_head = wk.head; }
                            if (mode == edsc::M_RETURN) s_accept = 0;
                        }
                        if (wk.walk == edsc::W_EVAL) edsc::coop12_take(sv, s_cand[wk.k], lane);
                    }
                }
            } else {
            for (int e = tid; e < nval; e += nthr) {
                // all 2 K granules of the entry in flight at once (the members sit on other XCDs: every load is a fabric round trip,
                // and 2 K of them one after the other were ~4 us per evaluation), then only the late ones are polled again
                double tot = 0.0;
                constexpr int GM = TEAM < 4 ? TEAM : (TEAM == 16 ? 8 : 4);             // members per round of loads (all 8 members of a team in one round: no faster)
#pragma unroll
                for (int m0 = 0; m0 < TEAM; m0 += GM) {
                    unsigned long long v[2 * GM];
#pragma unroll
                    for (int k = 0; k < 2 * GM; ++k)
                        v[k] = __hip_atomic_load(mb + (size_t)(m0 + (k >> 1)) * EDS_TEAM12_GRANULES + 2 * e + (k & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {                  // the late ones are asked for again TOGETHER (one after the other, every straggler cost
                        bool late = false;      // a fabric round trip of its own even when all of them had long arrived)
#pragma unroll
                        for (int k = 0; k < 2 * GM; ++k) late |= (unsigned)(v[k] >> 32) != tag;
                        if (!late) break;
                        if (__builtin_amdgcn_s_memrealtime() - t_start > EDS_TEAM_TIMEOUT_TICKS) { s_timeout = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
#pragma unroll
                        for (int k = 0; k < 2 * GM; ++k)
                            if ((unsigned)(v[k] >> 32) != tag)
                                v[k] = __hip_atomic_load(mb + (size_t)(m0 + (k >> 1)) * EDS_TEAM12_GRANULES + 2 * e + (k & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int m = 0; m < GM; ++m)
                        tot += __longlong_as_double((long long)(((v[2 * m + 1] & 0xffffffffull) << 32) | (v[2 * m] & 0xffffffffull)));
                }
                *entry(e) = tot;
            }
            ++pass_no;
            __syncthreads();
            if (s_timeout) {                    // a member never showed up: report it, the host solves the range again without teams
                if (tid == 0) { sv.termination = edss::TERM_FAILURE; sv.num_unsuccessful = -2; }
                break;
            }
            }
        }
        if (GROUPS == 1) EDS12_STAMP(1);
        if (GROUPS == 1 && wave == 0) {         // the LM state machine (eds_solver12_coop.hpp): what this evaluation means, the
            const int mode = edsc::coop12_decide(sv, sums, work, s_pose, lane);          // linearisation if it was accepted,
            // and the bookkeeping up to the next step — a prepared one, if there is one (every lane walks; lane 0 keeps the outcome)
            edsc::Walk12 wk{edsc::W_RETURN, 0, 0};
            if (mode != edsc::M_RETURN)
                wk = edsc::coop12_walk(sv, s_cand, (mode == edsc::M_ADVANCE) ? edsc::uniform_int(s_k) + 1 : EDS_NCAND, 0, lane);   // one notch down the radius sequence / all stale
            if (lane == 0) {
                s_accept = (mode == edsc::M_LIN_ITER0 || mode == edsc::M_LIN_ACCEPT) ? 1 : 0;
                s_walk = wk.walk;
                if (mode != edsc::M_RETURN) { s_k = wk.k; s_head = wk.head; }
            }
            if (wk.walk == edsc::W_EVAL) edsc::coop12_take(sv, s_cand[wk.k], lane);
        }
        __syncthreads();
        // the sums are consumed: every thread clears its share for the next evaluation (the barrier that ends the solver phase orders
        // this before the next sweep's atomics)
        for (int k = 1 + tid; k < (int)(sizeof(sums) / sizeof(double)); k += nthr) reinterpret_cast<double*>(&sums)[k] = 0.0;
        while (s_walk == edsc::W_NEED) {        // no step prepared for this radius: EDS_NCAND wavefronts prepare the next ones side by side
#ifdef EDS_FUSED_STAMPS
#define EDS12_SOLVE_STAMP(k) do { if (tid == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); work.st[k] += n_ - work.st_t; work.st_t = n_; } } while (0)
#else
#define EDS12_SOLVE_STAMP(k) do { } while (0)
#endif
            EDS12_SOLVE_STAMP(2);               // bookkeeping + barrier
#ifdef EDS_FUSED_STAMPS
            if (tid == 0) ++st_prop;
#endif
            // Round 6: the walk over the prepared steps needs their validity flags and the solver state, not their pose blocks — where the
            // workgroup has a wavefront to spare (512 threads) it runs on wavefront EDS_NCAND WHILE wavefronts 0 .. 3 fill the pose blocks
            // (one barrier moved); the 256-thread shape has none and keeps the order.
            constexpr bool SPARE = NTHR / 64 > EDS_NCAND;
            if (wave < EDS_NCAND) {
                edsc::coop12_propose(sv, wave, s_cand[wave], s_step[wave], lane);
                EDS12_SOLVE_STAMP(3);           // factorisation, substitutions, model cost change, candidate point
            }
            if (SPARE) __syncthreads();         // every candidate's validity is out
            if (wave < EDS_NCAND) {
                if (edsc::uniform_int(s_cand[wave].valid))
                    edsc::coop_fill_pose_block(s_cand[wave].cp, s_cand[wave].cq, s_cand[wave].cv, s_G, nb, s_pb[wave], lane);
                EDS12_SOLVE_STAMP(4);           // pose block
            }
            if (!SPARE) __syncthreads();
            EDS12_SOLVE_STAMP(5);               // waiting for the slowest of the proposing wavefronts
            if (wave == (SPARE ? EDS_NCAND : 0)) {
                const edsc::Walk12 wk = edsc::coop12_walk(sv, s_cand, 0, edsc::uniform_int(s_head), lane);
                if (lane == 0) { s_walk = wk.walk; s_k = wk.k; s_head = wk.head; }
                if (wk.walk == edsc::W_EVAL) edsc::coop12_take(sv, s_cand[wk.k], lane);
            }
            __syncthreads();            // (s_walk: the loop condition of every thread; the taken step: before its readers)
        }
        if (tid == 0) {
#ifdef EDS_FUSED_STAMPS
            const unsigned long long n_ = __builtin_readcyclecounter(); work.st[6] += n_ - work.st_t; work.st_t = n_;
#endif
            s_state = (s_walk == edsc::W_RETURN) ? 2 : 0;
        }
        __syncthreads();
        EDS12_STAMP(2);
#ifdef EDS_FUSED_STAMPS
        ++st_n;
#endif
        if (GROUPS > 1) {                       // candidate groups: the residuals of the accepted point stay in the registers of the group that evaluated it
            if (s_accept) {
                res_owner = s_gacc;
                if (group == s_gacc) { racc[0] = rkeep[0]; racc[1] = rkeep[1]; }
            }
        } else
        if (s_accept) {                         // the candidate became the accepted point: its residuals are the ones to keep
            // each thread copies what it wrote itself; its first two points out of registers — a lone alignment on 8 CUs has nothing
            // else, and the load of the value just stored (an L2 round trip) sat at the head of the next evaluation
#pragma unroll
            for (int jj = 0; jj < (FULL ? 4 : 2); ++jj) { const int i = lo + jj * nthr + tid; if (i < hi) A.r[base + i] = rkeep[jj]; }
            if (!FULL)
                for (int i = lo + 2 * nthr + tid; i < hi; i += nthr) A.r[base + i] = (RC_CAP > 0 && i - lo < RC_CAP) ? s_rc[i - lo] : A.mhat[base + i];
        }
        if (s_state == 2) break;
    }

#ifdef EDS_FUSED_STAMPS
    if (tid == 0 && blockIdx.x == 0) {
        printf("[stamps12] lane-0 cycles per evaluation: points %llu  reduce %llu  solver %llu  (%d evaluations, %d threads)\n",
               st_acc[0] / st_n, st_acc[1] / st_n, st_acc[2] / st_n, st_n, NTHR);
        printf("[stamps12]   point phase: copy+loop %llu  pose %llu  constants %llu  project+probe+issue %llu  taps+spline %llu  row %llu  stage+mfma %llu  flush %llu\n",
               pst[5] / st_n, pst[6] / st_n, pst[7] / st_n, pst[0] / st_n, pst[1] / st_n, pst[2] / st_n, pst[3] / st_n, pst[4] / st_n);
        printf("[stamps12]   propose (wavefront 0, per call): loads+damping %llu  factorisation %llu  substitutions %llu  A s %llu  model cost change %llu  candidate point %llu\n",
               s_step[0].pst[0] / (st_prop > 0 ? st_prop : 1), s_step[0].pst[1] / (st_prop > 0 ? st_prop : 1), s_step[0].pst[2] / (st_prop > 0 ? st_prop : 1),
               s_step[0].pst[3] / (st_prop > 0 ? st_prop : 1), s_step[0].pst[4] / (st_prop > 0 ? st_prop : 1), s_step[0].pst[5] / (st_prop > 0 ? st_prop : 1));
        printf("[stamps12]   solver split: decide %llu linearise %llu bookkeeping %llu propose %llu poseblock %llu wait %llu walk+rest %llu\n",
               work.st[0] / st_n, work.st[1] / st_n, work.st[2] / st_n, work.st[3] / st_n, work.st[4] / st_n, work.st[5] / st_n, work.st[6] / st_n);
    }
#endif
    if (GROUPS > 1) {                           // (a member's slice is at most 2 x NTHR points: the launcher forms groups only then)
        if (group == res_owner) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int i = lo + jj * nthr + tid;
                if (i < hi) { A.r[base + i] = racc[jj]; if (A.rmap) A.rmap[base + i] = racc[jj]; }
            }
        }
    } else
    if (A.rmap) {                               // the kept residuals into the pinned mirror (each thread: the entries it wrote itself)
        for (int i = lo + tid; i < hi; i += nthr) A.rmap[base + i] = A.r[base + i];
    }
    // small solves: this workgroup's completion word (EdsArrays::done) — every thread's writes are out at system scope first
    unsigned* const done_word = A.done ? A.done + (size_t)slot * EDS_DONE_WORDS + (TEAM > 1 ? group * TEAM + member : 0) : nullptr;
    if (done_word) { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); __syncthreads(); }      // (vmcnt(0): this wavefront's stores are acknowledged — see eds_fused6_kernel)
    if (TEAM > 1 && (member != 0 || group != 0)) {              // every member holds the same result; member 0 (of group 0) reports it
        if (tid == 0 && done_word) __hip_atomic_store(done_word, A.done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (tid == 0) {
        EdsFused12Out& O = out[slot];
        const bool ok = sv.termination != edss::TERM_FAILURE;
        for (int i = 0; i < 3; ++i) O.p[i] = ok ? sv.best_p[i] : sv.p[i];
        for (int i = 0; i < 4; ++i) O.q[i] = ok ? sv.best_q[i] : sv.q[i];
        for (int i = 0; i < 6; ++i) O.v[i] = ok ? sv.best_v[i] : sv.v[i];
        O.initial_cost = sv.initial_cost; O.final_cost = sv.minimum_cost;
        O.termination = sv.termination; O.num_successful = sv.num_successful; O.num_unsuccessful = sv.num_unsuccessful;
        O.failed = ok ? 0 : (TEAM > 1 && sv.num_unsuccessful == -2 ? 2 : 1);     // 2: team timeout
        O.t_end = __builtin_amdgcn_s_memrealtime();
        if (done_word) __hip_atomic_store(done_word, A.done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // (system scope only where a host polls: see eds_fused6_kernel)
    }
}

// ---------------------------------------------------------------------------------------
bool eds_fused12_supported(const eds_trk* h, int first, int count) {
    int nb = h->cfg.num_blocks < 1 ? 1 : h->cfg.num_blocks;
    return nb <= EDS_DEV_MAX_BLOCKS;            // any number of points
}

int eds_fused12_solve(eds_trk* h, int level, int first, int count) {
    EdsFusedBuffers& fb = h->fused;
    if (fb.pending_count > 0) return eds_internal_fail(EDS_ERR_STATE, "previous batch not collected: call eds_trk_sync first");
    int lv = level < 0 ? 0 : (level >= EDS_MAX_LEVELS ? EDS_MAX_LEVELS - 1 : level);
    const int iters = h->cfg.max_num_iterations[lv];
    const int nb = h->cfg.num_blocks < 1 ? 1 : h->cfg.num_blocks;
    for (int s = first; s < first + count; ++s) {
        const Slot& sl = h->slots[s];
        if (!sl.has_kf || !sl.has_frame) return eds_internal_fail(EDS_ERR_STATE, "keyframe or event frame not set");
        EdsFusedIn& I = fb.h_in[s];
        std::memcpy(I.p, sl.p, sizeof(I.p)); std::memcpy(I.q, sl.q, sizeof(I.q)); std::memcpy(I.v, sl.v, sizeof(I.v));
    }
    hipError_t e = hipSuccess;           // (start states and results: pinned host memory the kernel accesses directly, eds_fused_alloc)
    EdsArrays A = h->arrays();
    // WHICH kernel: eds_launch_rule.hpp (pure; table-tested on the CPU) — shape (one or two alignments per CU), team size, gather form.
    const EdsKnobs& kn = h->knobs;
    int maxN = 0;
    for (int s = first; s < first + count; ++s) maxN = std::max(maxN, h->slots[s].N);
    const EdsRef12In rin{maxN, count, h->cfg.sampling == EDS_SAMPLE_BICUBIC ? 1 : 0, h->cfg.nc ? 1 : 0, h->H, fb.pending_retry ? 1 : 0, nb};
    EdsRef12Plan pl;
    eds_ref12_plan_begin(kn, rin, pl);
    const bool team_ok = pl.wants_team && eds_team_allowed(&fb);         // the time-out policy of eds_fused.hpp
    fb.pending_paused = pl.wants_team && !team_ok;
    eds_ref12_plan_team(kn, rin, team_ok ? 1 : 0, fb.team_cooldown > 0 ? 1 : 0, pl);
    if (pl.team > 1 && !fb.d_mail12) {
        if (hipMalloc((void**)&fb.d_mail12, EDS_TEAM12_MAIL_BYTES) != hipSuccess) {       // no mailboxes: one CU per alignment
            (void)hipGetLastError();
            EdsKnobs k1 = kn; k1.ref12_team = 1;
            eds_ref12_plan_team(k1, rin, 0, 1, pl);
        } else hipMemsetAsync(fb.d_mail12, 0, EDS_TEAM12_MAIL_BYTES, h->st);
    }
    const int team = pl.team;
    if (team > 1) {
        if (++fb.epoch >= (1u << 24)) {
            hipMemsetAsync(fb.d_mail, 0, EDS_TEAM_MAIL_BYTES, h->st);
            hipMemsetAsync(fb.d_mail12, 0, EDS_TEAM12_MAIL_BYTES, h->st);
            fb.epoch = 1;
        }
    }
    fb.pending_team = team; fb.pending_level = level;
    const int drop = kn.team_drop ? 1 : 0;            // test hook for the time-out path (see eds_fused_solve)
    fb.pending_ticks = count <= 64;                  // as eds_fused_solve
    const bool done_words = fb.pending_ticks && kn.poll_results && !drop;                     // (see eds_fused_solve; the group count is only known below)
    const bool rmap_in_kernel = (team > 1 || done_words) && h->d_rmap && first + count <= EDS_RHOST_SLOTS;       // (see eds_fused_solve)
    A.rmap = rmap_in_kernel ? h->d_rmap : nullptr;
    const unsigned ticket_base = fb.ticket_base;
    if (!fb.pending_ticks) hipEventRecord(h->ev0, h->st);
    // the strip copies of the frames: asked for only where an instantiation reads them (teams of up to 4, both one-CU shapes)
    const bool strips = pl.strips_eligible && eds_strips_for_solve(h, first, count);
    eds_ref12_plan_finish(kn, rin, strips ? 1 : 0, pl);
    A.strips = h->dstrips; A.strip_phases = h->strip_phases;
    const int groups = pl.K > 1 ? pl.G : 1;          // candidate groups: G x K workgroups per alignment
    fb.pending_vteam = 0;
    if (done_words && pl.K * groups <= EDS_DONE_WORDS) {
        A.done = fb.d_done; A.done_tag = eds_next_done_tag(&fb); fb.pending_vteam = pl.K * groups;
    }
    if (!eds_fused12_instance_exists(pl.S, pl.T, pl.CAP, pl.NC, pl.K, pl.Q, groups))
        return eds_internal_fail(EDS_ERR_INVALID, "internal: the launch rule chose an instantiation the library does not hold");
    if (team > 1) {
        fb.ticket_base += (unsigned)(count * team * groups);
        for (int s = first; s < first + count; ++s) fb.h_out12[s].failed = 2;    // "no result yet" reads as a time-out (eds_fused_solve)
    }
    if (fb.pending_ticks) for (int s = first; s < first + count; ++s) fb.h_out12[s].t_end = 0;      // (the completion word wait_stream polls)
    if (groups > 1) std::snprintf(fb.last_kernel, sizeof(fb.last_kernel), "eds_fused12_kernel<%d, %d, %d, %s, %d, %d, %d>", pl.S, pl.T, pl.CAP, pl.NC ? "true" : "false", pl.K, pl.Q, groups);
    else std::snprintf(fb.last_kernel, sizeof(fb.last_kernel), "eds_fused12_kernel<%d, %d, %d, %s, %d, %d>", pl.S, pl.T, pl.CAP, pl.NC ? "true" : "false", pl.K, pl.Q);
    fb.last_workgroups = count * pl.K * groups - (pl.K > 1 ? drop : 0); fb.last_team = pl.K * groups; fb.last_layout = pl.Q == 2 ? 2 : 1;
    bool launched = false;
#define EDS_INST_LAUNCH12G_(S_, T_, C_, N_, K_, Q_, G_)                                                                               \
    if (!launched && groups == G_ && pl.S == S_ && pl.T == T_ && pl.CAP == C_ && (pl.NC != 0) == N_ && pl.K == K_ && pl.Q == Q_) {   \
        launched = true;                                                                                                              \
        hipLaunchKernelGGL((eds_fused12_kernel<S_, T_, C_, N_, K_, Q_, G_>), dim3(count * K_ * G_ - drop), dim3(T_), 0, h->st, A, fb.d_in, \
                           fb.d_out12, first, iters, h->cfg.loss_type, h->cfg.loss_param, h->cfg.function_tolerance, h->cfg.gradient_tolerance, \
                           h->cfg.parameter_tolerance, nb, fb.d_mail12, fb.d_ticket, ticket_base, fb.epoch);                         \
    }
    if (groups > 1) { EDS_FUSED12_GROUP_INSTANCES(EDS_INST_LAUNCH12G_) }
#undef EDS_INST_LAUNCH12G_
#define EDS_INST_LAUNCH12_(S_, T_, C_, N_, K_, Q_)                                                                                    \
    if (!launched && groups == 1 && pl.S == S_ && pl.T == T_ && pl.CAP == C_ && (pl.NC != 0) == N_ && pl.K == K_ && pl.Q == Q_) {    \
        launched = true;                                                                                                              \
        hipLaunchKernelGGL((eds_fused12_kernel<S_, T_, C_, N_, K_, Q_>), dim3(count * K_ - ((K_) > 1 ? drop : 0)), dim3(T_), 0, h->st, A, fb.d_in, \
                           fb.d_out12, first, iters, h->cfg.loss_type, h->cfg.loss_param, h->cfg.function_tolerance, h->cfg.gradient_tolerance, \
                           h->cfg.parameter_tolerance, nb, fb.d_mail12, fb.d_ticket, ticket_base, fb.epoch);                         \
    }
    EDS_FUSED12_INSTANCES(EDS_INST_LAUNCH12_)
#undef EDS_INST_LAUNCH12_
    if (!fb.pending_ticks) hipEventRecord(h->ev1, h->st);
    fb.pending_host_r = rmap_in_kernel ? true : eds_mirror_residuals(h, first, count);
    if (fb.pending_host_r && !rmap_in_kernel) fb.pending_vteam = 0;      // (a mirror launch follows: see eds_fused_solve)
    e = hipGetLastError();
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    fb.pending_first = first;
    fb.pending_count = count;
    fb.pending_kind = 12;
    fb.last_first = first; fb.last_count = count; fb.last_kind = 12; fb.last_ticks = fb.pending_ticks;
    fb.launch_wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return EDS_OK;
}

int eds_fused12_collect(eds_trk* h) {
    EdsFusedBuffers& fb = h->fused;
    if (fb.pending_team > 1) {                        // a team whose members did not all become resident within the bound
        bool timed_out = false;
        for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) timed_out |= fb.h_out12[s].failed == 2;
        if (timed_out) {
            eds_team_timed_out(h);
            const int pf = fb.pending_first, pc = fb.pending_count;
            fb.pending_count = 0;
            fb.pending_retry = true;
            int rc = eds_fused12_solve(h, fb.pending_level, pf, pc);
            if (rc != EDS_OK) { fb.pending_retry = false; return rc; }
            hipError_t e = hipStreamSynchronize(h->st);
            if (e != hipSuccess) { fb.pending_retry = false; return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e)); }
            return eds_fused12_collect(h);
        }
        eds_team_clean(&fb);
    }
    const double now = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    float dev_ms = 0.f;
    if (fb.pending_ticks) {
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) { t0 = std::min(t0, fb.h_out12[s].t_begin); t1 = std::max(t1, fb.h_out12[s].t_end); }
        dev_ms = t1 > t0 ? (float)((double)(t1 - t0) * 1e-5) : 0.f;        // 100 MHz ticks
    } else {
        hipEventElapsedTime(&dev_ms, h->ev0, h->ev1);
    }
    for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) {
        Slot& sl = h->slots[s];
        const EdsFused12Out& O = fb.h_out12[s];
        const bool ok = O.failed == 0;
        if (ok) { std::memcpy(sl.p, O.p, sizeof(sl.p)); std::memcpy(sl.q, O.q, sizeof(sl.q)); std::memcpy(sl.v, O.v, sizeof(sl.v)); }
        sl.res_on_device = ok; sl.res_in_hostmap = ok && fb.pending_host_r;
        sl.trace_on_device = false;
        sl.residuals.clear();
        sl.ntrace = 0;
        eds_trk_info& in = sl.info;
        std::memset(&in, 0, sizeof(in));
        in.meas_time_us = now - fb.launch_wall_us;
        in.time_seconds = in.meas_time_us * 1e-6;
        in.device_time_us = dev_ms * 1e3;
        in.num_points = sl.N;
        in.num_successful_steps = O.num_successful;
        in.num_unsuccessful_steps = O.num_unsuccessful;
        in.num_iterations = O.num_successful + O.num_unsuccessful;       // Tracker.cpp:211
        in.success = ok;
        in.flags = (fb.pending_retry ? EDS_INFO_TEAM_TIMEOUT : 0) | (fb.pending_paused ? EDS_INFO_TEAMS_PAUSED : 0);
        in.termination = O.termination;
        in.initial_cost = O.initial_cost;
        in.final_cost = O.final_cost;
    }
    fb.pending_count = 0;
    fb.pending_kind = 0;
    fb.pending_retry = false; fb.pending_paused = false;
    return EDS_OK;
}
