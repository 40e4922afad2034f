// Which kernel does a solve launch?  The rule, as PURE functions of the handle's tuning knobs and of the shape of the range —
// no HIP, no handle, no environment access on the solve path — so that tests/host_logic can table-test it on the CPU
// (tests/test_host_logic.py::test_launch_rule_*), and so that two handles with different knobs coexist in one process.
//
// Knobs (EdsKnobs): every former per-solve getenv() of eds_fused.hip / eds_fused12.hip / eds_strips.hip / eds_capi.hip.  They are
// resolved ONCE, at eds_trk_create (eds_knobs_from_env), live in the handle, and can be changed per handle with eds_trk_set_knob
// (include/eds_hip.h) — same names, same values as the environment variables.  Defaults are the measured best; the knobs exist for
// A/B runs and tests.
//
// The pose-only rule runs in three steps because two of its inputs have side effects the launcher owns: the time-out policy of the
// teams counts a cool-down down when it is asked (eds_team_allowed), and the strip copies of the frames are converted when a solve
// asks for them (eds_strips_for_solve).  So:   begin (shape, does it want teams?)  ->  team (given the policy's answer: team size,
// would it gather from strips?)  ->  finish (given the copies' answer: the template arguments).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <initializer_list>

#define EDS_RULE_TEAM6_MAX 16          // = EDS_TEAM6_MAX (eds_fused.hpp; static_assert there)
#define EDS_RULE_TEAM_SLOTS 128        // = EDS_TEAM_SLOTS
#define EDS_RULE_TEAM12_SLOTS 64       // = EDS_TEAM12_SLOTS
#define EDS_RULE_TEAM12_MEMBERS 512    // = EDS_TEAM12_MEMBERS
#define EDS_RULE_TEAM_MEMBERS 4096     // = EDS_TEAM_MEMBERS
#define EDS_RULE_CACHE_CAP 2048        // = EDS_CACHE_CAP (eds_fused.hip)
#define EDS_RULE_CUS 256               // compute units of an MI355X: the default of EdsKnobs::cus (the handle takes its device's real count at eds_trk_create)

struct EdsKnobs {
    int cus = EDS_RULE_CUS;     // (not a knob) compute units of the handle's device: candidate groups are formed only while every workgroup of the launch gets a CU of its own
    int ref12_exec = -1;        // EDS_REF12_EXEC      device | host            -1: the rule (device wherever the kernel covers the problem)
    int fused_threads = 0;      // EDS_FUSED_THREADS   64 .. 1024, multiple of 64   0: the rule
    int fused_ppt_set = 0;      // EDS_FUSED_PPT       points per lane; set but infeasible -> 0 (constants re-read per pass)
    int fused_ppt = 0;
    int lm6_spec = 1;           // EDS_LM6_SPEC        0: the serial solver lane of round 1
    int lm6_kernel = 0;         // EDS_LM6_KERNEL      0: the rule, 1 resident (or any other word), 2 paired / stream, 3 wide
    int layout_tiles = 0;       // EDS_FUSED_LAYOUT    tiles: never gather from the strip copies
    int team_drop = 0;          // EDS_TEAM_TEST_DROP_MEMBER   test hook: a team launch goes out one workgroup short
    int lm6_team = 0;           // EDS_LM6_TEAM        1 | 2 | 4 | 8 | 16          0: the rule
    int team_wide = -1;         // EDS_TEAM_WIDE       0 | 1                      -1: the rule
    int gather = 0;             // EDS_FUSED_GATHER    0: the rule, 1 quad (any word but "lane"), 2 lane
    int report = 0;             // EDS_FUSED_REPORT    per-launch digest of the workgroups' stamps on stderr
    int ref12_kernel = 0;       // EDS_REF12_KERNEL    0: the rule, 1 wide, 2 paired
    int ref12_team = 0;         // EDS_REF12_TEAM      1 | 2 | 4 | 8 | 16          0: the rule
    int strips_phases = 0;      // EDS_STRIPS_PHASES   1 | 2 | 4                   0: the rule (4 for batch handles, 1 below 32 slots)
    int strips_policy = 0;      // EDS_STRIPS_POLICY   0 reuse, 1 eager, 2 never
    int strips_budget_pct = 50; // EDS_STRIPS_BUDGET_PCT  the strip copies may take at most this share of the device memory that is FREE when
                                //                     they are first allocated (1 .. 95); fewer row phases, or none, beyond it
    int no_spin = 0;            // EDS_NO_SPIN         block in the stream wait from the start
    int poll_results = 1;       // EDS_POLL_RESULTS    0: small solves are waited for through the stream (hipStreamQuery) instead of their result records
    int upload_bands = 0;       // EDS_UPLOAD=bands    one launch per band of a host frame (round 2's upload)
    int frame_rowmajor = 0;     // EDS_FRAME_LAYOUT=rowmajor   (read at create only: it decides the allocation)
    int reduce_ppl = 4;         // EDS_REDUCE_PPL      4 | 8: points a lane of eds_reduce_kernel<6> folds (16-byte loads)
    int upload_threads = 0;     // EDS_UPLOAD_THREADS  1 .. 64: host threads that narrow the frames of eds_trk_set_event_frames   0: the rule (4)
    int force6_set = 0, force6[6] = {0, 0, 0, 0, 0, 0};        // EDS_FORCE_FUSED6   "S,P,T,Q,K,G": launch exactly this instantiation of eds_fused6_kernel
    int force12_set = 0, force12[6] = {0, 0, 0, 0, 0, 0};      // EDS_FORCE_FUSED12  "S,T,CAP,NC,K,Q": ... of eds_fused12_kernel.  Test hooks (tests/test_instances_gpu.py
                                                               // walks the compiled lists with them); honoured only where the instantiation exists and fits the range
    int ref12_groups = 0;       // EDS_REF12_GROUPS    1 | 2 | 4: candidate groups of a REF12 team launch (eds_fused12.hip)   0: the rule
    int upload_streams = 0;     // EDS_UPLOAD_STREAMS  1: the batch upload keeps to the handle's stream (default: alternate frames on a second one)
    int upload_dma = 0;         // EDS_UPLOAD_DMA      1: the batch upload moves the staged frames with the copy engine (hipMemcpyAsync) instead of kernels reading pinned memory
    int lm6_groups = 0;         // EDS_LM6_GROUPS      1 | 2 | 4 | 8: candidate groups of a team launch (eds_fused.hip)   0: the rule
};

// returns 0, -1 for a name that is not a knob, -2 for a value the knob does not take (nothing is changed then).  value == nullptr or ""
// resets the knob to its default.  (Round 5, ADVICE r4: values used to be coerced silently — EDS_REDUCE_PPL=abc became 4.)
static inline int eds_knobs_set(EdsKnobs* k, const char* name, const char* value) {
    const EdsKnobs d;
    const bool unset = !value || !value[0];
    auto is = [&](const char* w) { return !unset && strcmp(value, w) == 0; };
    // a whole decimal integer, or "not a number"
    bool num = false; long iv = 0;
    if (!unset) { char* e = nullptr; iv = strtol(value, &e, 10); num = e != value && *e == 0; }
    auto one_of = [&](std::initializer_list<long> ok) { if (!num) return false; for (long v : ok) if (v == iv) return true; return false; };
    auto flag = [&](int* dst, int dflt) { if (unset) { *dst = dflt; return 0; } if (!one_of({0, 1})) return -2; *dst = (int)iv; return 0; };
    if (!strcmp(name, "EDS_REF12_EXEC")) { if (unset) k->ref12_exec = d.ref12_exec; else if (is("device")) k->ref12_exec = 1; else if (is("host")) k->ref12_exec = 0; else return -2; }
    else if (!strcmp(name, "EDS_FUSED_THREADS")) { if (unset) k->fused_threads = 0; else if (num && iv >= 64 && iv <= 1024 && iv % 64 == 0) k->fused_threads = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_FUSED_PPT")) { if (unset) { k->fused_ppt_set = 0; k->fused_ppt = 0; } else if (num && iv >= 0 && iv <= 64) { k->fused_ppt_set = 1; k->fused_ppt = (int)iv; } else return -2; }
    else if (!strcmp(name, "EDS_LM6_SPEC")) return flag(&k->lm6_spec, 1);
    else if (!strcmp(name, "EDS_LM6_KERNEL")) { if (unset) k->lm6_kernel = 0; else if (is("wide")) k->lm6_kernel = 3; else if (is("paired") || is("stream")) k->lm6_kernel = 2; else if (is("resident")) k->lm6_kernel = 1; else return -2; }
    else if (!strcmp(name, "EDS_FUSED_LAYOUT")) { if (unset || is("strips")) k->layout_tiles = 0; else if (is("tiles")) k->layout_tiles = 1; else return -2; }
    else if (!strcmp(name, "EDS_TEAM_TEST_DROP_MEMBER")) return flag(&k->team_drop, 0);
    else if (!strcmp(name, "EDS_LM6_TEAM")) { if (unset) k->lm6_team = 0; else if (one_of({1, 2, 4, 8, 16})) k->lm6_team = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_TEAM_WIDE")) return flag(&k->team_wide, -1);
    else if (!strcmp(name, "EDS_FUSED_GATHER")) { if (unset) k->gather = 0; else if (is("lane")) k->gather = 2; else if (is("quad")) k->gather = 1; else return -2; }
    else if (!strcmp(name, "EDS_FUSED_REPORT")) return flag(&k->report, 0);
    else if (!strcmp(name, "EDS_REF12_KERNEL")) { if (unset) k->ref12_kernel = 0; else if (is("wide")) k->ref12_kernel = 1; else if (is("paired")) k->ref12_kernel = 2; else if (is("full")) k->ref12_kernel = 3; else if (is("half")) k->ref12_kernel = 4; else return -2; }
    else if (!strcmp(name, "EDS_REF12_TEAM")) { if (unset) k->ref12_team = 0; else if (one_of({1, 2, 4, 8, 16})) k->ref12_team = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_STRIPS_PHASES")) { if (unset) k->strips_phases = 0; else if (one_of({1, 2, 4})) k->strips_phases = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_STRIPS_POLICY")) { if (unset || is("reuse")) k->strips_policy = 0; else if (is("eager")) k->strips_policy = 1; else if (is("never")) k->strips_policy = 2; else return -2; }
    else if (!strcmp(name, "EDS_STRIPS_BUDGET_PCT")) { if (unset) k->strips_budget_pct = d.strips_budget_pct; else if (num && iv >= 1 && iv <= 95) k->strips_budget_pct = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_NO_SPIN")) return flag(&k->no_spin, 0);
    else if (!strcmp(name, "EDS_POLL_RESULTS")) return flag(&k->poll_results, 1);
    else if (!strcmp(name, "EDS_UPLOAD")) { if (unset) k->upload_bands = 0; else if (is("bands")) k->upload_bands = 1; else return -2; }
    else if (!strcmp(name, "EDS_FRAME_LAYOUT")) { if (unset || is("tiles")) k->frame_rowmajor = 0; else if (is("rowmajor")) k->frame_rowmajor = 1; else return -2; }
    else if (!strcmp(name, "EDS_REDUCE_PPL")) { if (unset) k->reduce_ppl = 4; else if (one_of({4, 8})) k->reduce_ppl = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_LM6_GROUPS")) { if (unset) k->lm6_groups = 0; else if (one_of({1, 2, 4, 8})) k->lm6_groups = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_REF12_GROUPS")) { if (unset) k->ref12_groups = 0; else if (one_of({1, 2, 4})) k->ref12_groups = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_UPLOAD_THREADS")) { if (unset) k->upload_threads = 0; else if (num && iv >= 1 && iv <= 64) k->upload_threads = (int)iv; else return -2; }
    else if (!strcmp(name, "EDS_UPLOAD_DMA")) return flag(&k->upload_dma, 0);
    else if (!strcmp(name, "EDS_UPLOAD_STREAMS")) { if (unset) k->upload_streams = 0; else if (one_of({1, 2})) k->upload_streams = iv == 1 ? 1 : 0; else return -2; }
    else if (!strcmp(name, "EDS_FORCE_FUSED6") || !strcmp(name, "EDS_FORCE_FUSED12")) {
        int* dst = name[15] == '6' ? k->force6 : k->force12;
        int* set = name[15] == '6' ? &k->force6_set : &k->force12_set;
        if (unset) { *set = 0; return 0; }
        int v[6], n = 0;
        const char* c = value;
        while (n < 6) {
            char* e = nullptr;
            v[n] = (int)strtol(c, &e, 10);
            if (e == c) break;
            ++n; c = e;
            if (*c == ',') ++c; else break;
        }
        if (n != 6 || *c) return -2;                    // not six integers
        for (int i = 0; i < 6; ++i) dst[i] = v[i];
        *set = 1;
    }
    else return -1;
    return 0;
}

#define EDS_KNOB_NAMES(X)                                                                                                              \
    X("EDS_REF12_EXEC") X("EDS_FUSED_THREADS") X("EDS_FUSED_PPT") X("EDS_LM6_SPEC") X("EDS_LM6_KERNEL") X("EDS_FUSED_LAYOUT")        \
    X("EDS_TEAM_TEST_DROP_MEMBER") X("EDS_LM6_TEAM") X("EDS_TEAM_WIDE") X("EDS_FUSED_GATHER") X("EDS_FUSED_REPORT")                   \
    X("EDS_REF12_KERNEL") X("EDS_REF12_TEAM") X("EDS_STRIPS_PHASES") X("EDS_STRIPS_POLICY") X("EDS_STRIPS_BUDGET_PCT")               \
    X("EDS_NO_SPIN") X("EDS_POLL_RESULTS") X("EDS_UPLOAD") X("EDS_FRAME_LAYOUT") X("EDS_REDUCE_PPL") X("EDS_LM6_GROUPS") X("EDS_UPLOAD_THREADS") X("EDS_UPLOAD_DMA") X("EDS_UPLOAD_STREAMS") X("EDS_FORCE_FUSED6") X("EDS_FORCE_FUSED12") X("EDS_REF12_GROUPS")

// the process environment, read once per handle (eds_trk_create)
// Returns the name of the first variable whose value its knob does not take (nullptr: none) — eds_trk_create refuses to make a handle
// whose environment it would otherwise silently ignore (EDS_NO_SPIN=yes used to "work").
static inline const char* eds_knobs_from_env(EdsKnobs* k) {
    const char* rejected = nullptr;
#define EDS_KNOB_ENV_(N) if (const char* v_ = getenv(N)) { if (eds_knobs_set(k, N, v_) != 0 && !rejected) rejected = N; }
    EDS_KNOB_NAMES(EDS_KNOB_ENV_)
#undef EDS_KNOB_ENV_
    return rejected;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// pose-only solvers (GN6 / LM6): eds_fused6_kernel<S, P, T, Q, K>, eds_stream6_kernel
// ---------------------------------------------------------------------------------------------------------------------------------
enum { EDS_K6_FUSED = 0, EDS_K6_TEAM = 1, EDS_K6_STREAM = 2 };

struct EdsLm6In {
    int maxN, count;        // most points of a slot in the range, alignments in the range
    int bicubic;            // cfg.sampling == EDS_SAMPLE_BICUBIC
    int iters;              // cfg.max_num_iterations[level]
    int lm6;                // cfg.solver == EDS_SOLVER_LM6 (else GN6)
    int huber;              // cfg.huber_tau > 0
    int H;                  // frame height
    int retry;              // this launch is the one-CU re-run of a timed-out team launch
};

struct EdsLm6Plan {
    // begin
    int threads, ppt, damped, stream, wide;
    int team_eligible, wants_team;
    // team
    int team, q, qany, quad, strips_eligible;
    // finish: the launch
    int kind;               // EDS_K6_*
    int S, P, T, Q, K;      // template arguments of eds_fused6_kernel (kind != STREAM); T is MAXT, the block has `threads` threads
    int G;                  // candidate groups (sixth template argument; 1 unless kind == TEAM): G x K workgroups per alignment
    int bilinear_tu;        // the instantiation lives in eds_fused_bilinear.o (eds_fused6_launch_bilinear)
    int wide_members;       // teams with members of 2 048 points
    int note_T;             // the T eds_trk_last_launch prints (the lane kernel of the bilinear sampler prints the block size)
};

static inline bool eds_fused6_instance_exists(int S, int P, int T, int Q, int K, int bilinear_tu, int G);
static inline bool eds_fused12_instance_exists(int S, int T, int CAP, int NC, int K, int Q, int G);
struct EdsRef12In;
static inline bool eds_lm6_force_feasible(const EdsKnobs& kn, const EdsLm6In& in, const EdsLm6Plan& p);
static inline bool eds_ref12_force_feasible(const EdsKnobs& kn, const EdsRef12In& in);

static inline void eds_lm6_plan_begin(const EdsKnobs& kn, const EdsLm6In& in, EdsLm6Plan& p) {
    memset(&p, 0, sizeof(p));
    // geometry: one alignment owns a CU's LDS (patch cache), so it also gets all 16 wave slots; the points-per-lane variant is picked
    // from the largest N of the range.  8 wavefronts x 4 points per lane measured 12 % faster than 16 x 2 at N = 2 000
    int threads = in.maxN <= 2048 ? 512 : 1024;
    if (kn.fused_threads >= 64 && kn.fused_threads <= 1024 && kn.fused_threads % 64 == 0) threads = kn.fused_threads;
    while (threads > 64 && threads / 2 >= in.maxN) threads /= 2;
    int ppt = (in.maxN + threads - 1) / threads;
    ppt = ppt <= 1 ? 1 : (ppt <= 2 ? 2 : (ppt <= 4 ? 4 : 0));
    if (kn.fused_ppt_set) ppt = (kn.fused_ppt > 0 && kn.fused_ppt * threads >= in.maxN) ? kn.fused_ppt : 0;
    int damped = in.lm6 ? 1 : 0;
    if (damped && !kn.lm6_spec) damped = 2;
    // Streaming variants (eds_stream6.hip: constants re-read per pass, any N): beyond 2 048 points, unless a handful of very large
    // alignments (the 1 024-thread resident kernel is faster there); teams replace them wherever prepared-candidate LM6 runs
    bool stream = in.maxN <= 2048 ? false : !(in.maxN > 4096 && in.count < 16);
    bool wide = in.maxN > 2048;
    if (kn.lm6_kernel) { stream = kn.lm6_kernel >= 2; wide = kn.lm6_kernel == 3; }
    p.threads = threads; p.ppt = ppt; p.damped = damped; p.stream = stream; p.wide = wide;
    // (513 .. 1 024 points: a member's slice is PPT x 512 = 1 024 points, a team of two would leave its second member without a point)
    p.team_eligible = damped == 1 && in.iters > 0 && in.maxN > 1024 && !in.retry;
    p.wants_team = p.team_eligible && (in.maxN > 2048 ? in.maxN <= 1024 * EDS_RULE_TEAM6_MAX : in.count <= EDS_RULE_TEAM_SLOTS);
}

// team_ok: wants_team and the handle's time-out policy allows teams now; cooldown_active: a cool-down is running (after the policy was asked)
static inline void eds_lm6_plan_team(const EdsKnobs& kn, const EdsLm6In& in, int team_ok, int cooldown_active, EdsLm6Plan& p) {
    int team = 1;
    if (team_ok && in.maxN <= 2048) {
        if (in.count <= 64) team = 4;
        else if (in.count <= EDS_RULE_TEAM_SLOTS) team = 2;
    } else if (team_ok && in.maxN <= 1024 * EDS_RULE_TEAM6_MAX) {
        team = in.maxN <= 4096 ? 4 : (in.maxN <= 8192 ? 8 : 16);      // 1 024 points per CU, at ANY batch size (the alternative streams)
    }
    if (kn.lm6_team) {
        const int v = kn.lm6_team;
        const bool feasible = p.damped == 1 && in.iters > 0 && (v == 2 || v == 4 || v == 8 || v == 16) && in.maxN <= (v == 2 ? 2048 : 1024 * v) &&
                              !in.retry && !cooldown_active;          // the override does not reach past the time-out policy
        if (v == 1 || feasible) team = v;
    }
    if (team > 1) { p.stream = 0; p.wide = 0; }
    p.team = team;
    if (kn.force6_set && kn.force6[3] >= 3 && eds_lm6_force_feasible(kn, in, p)) { p.strips_eligible = 1; return; }      // (the launcher then asks for the copies)
    const bool want_strips = !kn.layout_tiles;
    if (team > 1) {
        p.qany = in.count * team >= 128 && in.H < 8000;              // enough gathers in flight for the quad-cooperative form to pay
        p.q = in.bicubic && p.qany;                                  // (on tiles: the bicubic sampler only)
        p.strips_eligible = p.qany && !(team == 4 && in.maxN <= 2048) && want_strips;
    } else if (!p.stream) {
        bool quad = in.count >= 32;
        if (kn.gather) quad = kn.gather == 1;
        quad = quad && p.ppt > 0 && p.threads * p.ppt <= EDS_RULE_CACHE_CAP && in.H < 8000;     // every point's patch has a cache line of its own; 13-bit row field
        p.quad = quad;
        p.strips_eligible = quad && (p.ppt == 2 || p.ppt == 4) && want_strips;
    }
}

// EDS_FORCE_FUSED6: is the forced instantiation one the library holds AND one that can solve this range?  (sampler and Huber variant are
// the configuration's, every point needs a lane slot, teams need prepared-candidate LM6, the quad gather its 13-bit row field)
static inline bool eds_lm6_force_feasible(const EdsKnobs& kn, const EdsLm6In& in, const EdsLm6Plan& p) {
    if (!kn.force6_set) return false;
    const int S = kn.force6[0], P = kn.force6[1], T = kn.force6[2], Q = kn.force6[3], K = kn.force6[4], G = kn.force6[5];
    const int btu = (S == 1 && Q == 0) ? 1 : 0;
    if (G < 1 || K < 1 || !eds_fused6_instance_exists(S, P, T, Q, K, btu, G)) return false;
    if (S != (in.bicubic ? 0 : 1)) return false;
    if ((Q == 2 || Q == 4) != (in.huber != 0) && Q != 0) return false;            // Q = 0 takes the threshold at run time
    if (P > 0 && (long long)P * (K > 1 ? 512 : T) * K < in.maxN) return false;
    if (K > 1 && (p.damped != 1 || in.iters <= 0 || in.retry)) return false;
    if (K > 1 && (long long)in.count * K * G > EDS_RULE_TEAM_MEMBERS && G > 1) return false;
    if (Q != 0 && in.H >= 8000) return false;
    return true;
}

// strips: the copies of the range are current (asked only when strips_eligible)
static inline void eds_lm6_plan_finish(const EdsKnobs& kn, const EdsLm6In& in, int strips, EdsLm6Plan& p) {
    strips = strips && p.strips_eligible;
    if (eds_lm6_force_feasible(kn, in, p) && (kn.force6[3] < 3 || strips)) {      // the strip instantiations only on current copies
        const int S = kn.force6[0], P = kn.force6[1], T = kn.force6[2], Q = kn.force6[3], K = kn.force6[4], G = kn.force6[5];
        p.kind = K > 1 ? EDS_K6_TEAM : EDS_K6_FUSED;
        p.S = S; p.P = P; p.T = T; p.Q = Q; p.K = K; p.G = G; p.team = K; p.stream = 0; p.wide = 0;
        p.bilinear_tu = (S == 1 && Q == 0) ? 1 : 0; p.wide_members = (K > 1 && P == 4) ? 1 : 0;
        if (K == 1) p.threads = T;
        p.note_T = T;
        return;
    }
    const int bic = in.bicubic, hub = in.huber;
    p.bilinear_tu = 0; p.wide_members = 0;
    if (p.team > 1) {
        p.kind = EDS_K6_TEAM; p.T = 512;
        // Members of 2 048 points (4 per lane) once the members of 1 024 are more than one workgroup per CU
        const bool wide_ok = (p.q || (!bic && strips)) && in.maxN > 2048;
        bool wide_members = wide_ok && in.count * p.team > 256;
        if (kn.team_wide >= 0) wide_members = wide_ok && kn.team_wide != 0;
        if (wide_members) { p.team /= 2; p.wide_members = 1; }
        p.K = p.team;
        if (wide_members) {
            p.P = 4; p.S = bic ? 0 : 1;
            if (!bic || strips) p.Q = hub ? 4 : 3; else p.Q = hub ? 2 : 1;
        } else if (p.team == 4 && in.maxN <= 2048) {                 // 512 points per member, one per lane
            p.P = 1;
            if (bic) { p.S = 0; p.Q = p.q ? 1 : 0; } else { p.S = 1; p.Q = 0; p.bilinear_tu = 1; }
        } else {                                                     // 1 024 points per member, two per lane
            p.P = 2;
            if (!bic && strips) { p.S = 1; p.Q = hub ? 4 : 3; }
            else if (!bic) { p.S = 1; p.Q = 0; p.bilinear_tu = 1; }
            else if (!p.q) { p.S = 0; p.Q = 0; }
            else if (strips) { p.S = 0; p.Q = hub ? 4 : 3; }
            else { p.S = 0; p.Q = hub ? 2 : 1; }
        }
        p.note_T = 512;
        // Candidate groups (the latency regime proper: every workgroup of the launch on a CU of its own, the other CUs idle): G teams
        // evaluate G prepared candidates at once.  Instantiated for the members of 512 points (one per lane), lane / quad gather on tiles.
        p.G = 1;
        if (((p.P == 1 && p.K == 4) || (p.P == 2 && p.K == 2 && in.maxN <= 2048)) && !p.wide_members) {
            // measured (tools/check_groups.py, MI355X): a launch that takes HALF the CUs is faster than one that takes all of them
            // (8 alignments: 47.9 us with 4 groups on 128 CUs, 50.6 with 8 on 256), and two groups on all 256 still beat none
            // (32 alignments: 52.5 against 64.1 us); beyond the CU count the workgroups queue and a round waits for the queue
            int g = 1;
            for (int c = 8; c >= 2; c >>= 1)
                if (in.count * p.K * c <= kn.cus / 2) { g = c; break; }
            if (g == 1 && in.count * p.K * 2 <= kn.cus) g = 2;
            if (kn.lm6_groups == 1 || kn.lm6_groups == 2 || kn.lm6_groups == 4 || kn.lm6_groups == 8) g = kn.lm6_groups;
            if (in.count * p.K * g > EDS_RULE_TEAM_MEMBERS || !eds_fused6_instance_exists(p.S, p.P, p.T, p.Q, p.K, p.bilinear_tu, g)) g = 1;     // the mailboxes' capacity; compiled?
            p.G = g;
        }
        return;
    }
    p.G = 1;
    if (p.stream) { p.kind = EDS_K6_STREAM; p.S = bic ? 0 : 1; p.T = p.wide ? 512 : 256; p.P = p.wide ? 2048 : 1024; p.K = 1; p.note_T = p.T; return; }
    p.kind = EDS_K6_FUSED; p.K = 1;
    const int Tt = p.threads > 512 ? 1024 : 512;
    p.P = (p.ppt == 1 || p.ppt == 2 || p.ppt == 4) ? p.ppt : 0; p.T = Tt;     // any other count: constants re-read per pass (PPT = 0)
    if (strips) {
        p.S = bic ? 0 : 1; p.Q = hub ? 4 : 3;
        if (p.ppt == 4) p.T = 512;
    } else if (!bic) {
        p.S = 1; p.Q = 0; p.bilinear_tu = 1;
    } else {
        p.S = 0;
        switch (p.ppt) {
            case 1: p.Q = p.quad ? 1 : 0; break;
            case 2: p.Q = p.quad ? (hub ? 2 : 1) : 0; break;
            case 4: p.Q = p.quad ? (hub ? 2 : 1) : 0; if (p.quad) p.T = 512; break;
            default: p.P = 0; p.Q = 0; break;
        }
    }
    p.note_T = p.bilinear_tu ? p.threads : p.T;
}

// every instantiation of eds_fused6_kernel the library holds: X(S, P, T, Q, K).  The launcher dispatches over this list and the
// CPU test checks that the rule never leaves it.
#define EDS_FUSED6_MAIN_INSTANCES(X)                                                                                                  \
    X(0, 2, 512, 4, 1) X(0, 2, 512, 3, 1) X(0, 2, 1024, 4, 1) X(0, 2, 1024, 3, 1) X(0, 4, 512, 4, 1) X(0, 4, 512, 3, 1)             \
    X(1, 2, 512, 4, 1) X(1, 2, 512, 3, 1) X(1, 2, 1024, 4, 1) X(1, 2, 1024, 3, 1) X(1, 4, 512, 4, 1) X(1, 4, 512, 3, 1)             \
    X(0, 1, 512, 1, 1) X(0, 1, 512, 0, 1) X(0, 1, 1024, 1, 1) X(0, 1, 1024, 0, 1)                                                   \
    X(0, 2, 512, 2, 1) X(0, 2, 512, 1, 1) X(0, 2, 512, 0, 1) X(0, 2, 1024, 2, 1) X(0, 2, 1024, 1, 1) X(0, 2, 1024, 0, 1)             \
    X(0, 4, 512, 2, 1) X(0, 4, 512, 1, 1) X(0, 4, 512, 0, 1) X(0, 4, 1024, 0, 1) X(0, 0, 512, 0, 1) X(0, 0, 1024, 0, 1)             \
    X(1, 4, 512, 4, 2) X(1, 4, 512, 3, 2) X(0, 4, 512, 4, 2) X(0, 4, 512, 3, 2) X(0, 4, 512, 2, 2) X(0, 4, 512, 1, 2)               \
    X(1, 4, 512, 4, 4) X(1, 4, 512, 3, 4) X(0, 4, 512, 4, 4) X(0, 4, 512, 3, 4) X(0, 4, 512, 2, 4) X(0, 4, 512, 1, 4)               \
    X(1, 4, 512, 4, 8) X(1, 4, 512, 3, 8) X(0, 4, 512, 4, 8) X(0, 4, 512, 3, 8) X(0, 4, 512, 2, 8) X(0, 4, 512, 1, 8)               \
    X(0, 1, 512, 1, 4) X(0, 1, 512, 0, 4)                                                                                             \
    X(1, 2, 512, 4, 2) X(1, 2, 512, 3, 2) X(0, 2, 512, 0, 2) X(0, 2, 512, 4, 2) X(0, 2, 512, 3, 2) X(0, 2, 512, 2, 2) X(0, 2, 512, 1, 2) \
    X(1, 2, 512, 4, 4) X(1, 2, 512, 3, 4) X(0, 2, 512, 0, 4) X(0, 2, 512, 4, 4) X(0, 2, 512, 3, 4) X(0, 2, 512, 2, 4) X(0, 2, 512, 1, 4) \
    X(1, 2, 512, 4, 8) X(1, 2, 512, 3, 8) X(0, 2, 512, 0, 8) X(0, 2, 512, 4, 8) X(0, 2, 512, 3, 8) X(0, 2, 512, 2, 8) X(0, 2, 512, 1, 8) \
    X(1, 2, 512, 4, 16) X(1, 2, 512, 3, 16) X(0, 2, 512, 0, 16) X(0, 2, 512, 4, 16) X(0, 2, 512, 3, 16) X(0, 2, 512, 2, 16) X(0, 2, 512, 1, 16)
// ... and the lane kernels of the bilinear sampler, compiled into eds_fused_bilinear.o
#define EDS_FUSED6_BILINEAR_INSTANCES(X)                                                                                              \
    X(1, 1, 512, 0, 1) X(1, 1, 1024, 0, 1) X(1, 2, 512, 0, 1) X(1, 2, 1024, 0, 1) X(1, 4, 512, 0, 1) X(1, 4, 1024, 0, 1)             \
    X(1, 0, 512, 0, 1) X(1, 0, 1024, 0, 1) X(1, 1, 512, 0, 4) X(1, 2, 512, 0, 2) X(1, 2, 512, 0, 4) X(1, 2, 512, 0, 8) X(1, 2, 512, 0, 16)

// ... and the candidate-group instantiations X(S, P, T, Q, K, G), G > 1 (teams of four members of 512 points)
#define EDS_FUSED6_GROUP_INSTANCES(X)                                                                                                 \
    X(0, 1, 512, 0, 4, 2) X(0, 1, 512, 0, 4, 4) X(0, 1, 512, 0, 4, 8) X(0, 1, 512, 1, 4, 2) X(0, 1, 512, 1, 4, 4) X(0, 1, 512, 1, 4, 8) \
    X(0, 2, 512, 0, 2, 2) X(0, 2, 512, 1, 2, 2) X(0, 2, 512, 3, 2, 2) X(0, 2, 512, 2, 2, 2) X(0, 2, 512, 4, 2, 2) X(1, 2, 512, 3, 2, 2) X(1, 2, 512, 4, 2, 2) \
    X(0, 2, 512, 0, 2, 4) X(0, 2, 512, 1, 2, 4) X(0, 2, 512, 3, 2, 4)
#define EDS_FUSED6_BILINEAR_GROUP_INSTANCES(X)                                                                                        \
    X(1, 1, 512, 0, 4, 2) X(1, 1, 512, 0, 4, 4) X(1, 1, 512, 0, 4, 8) X(1, 2, 512, 0, 2, 2)

static inline bool eds_fused6_instance_exists(int S, int P, int T, int Q, int K, int bilinear_tu, int G) {
#define EDS_INST_EQ_(s, p, t, q, k) if (S == s && P == p && T == t && Q == q && K == k) return true;
#define EDS_INST_EQ6_(s, p, t, q, k, g) if (S == s && P == p && T == t && Q == q && K == k && G == g) return true;
    if (G > 1) {
        if (bilinear_tu) { EDS_FUSED6_BILINEAR_GROUP_INSTANCES(EDS_INST_EQ6_) }
        else { EDS_FUSED6_GROUP_INSTANCES(EDS_INST_EQ6_) }
        return false;
    }
    if (bilinear_tu) { EDS_FUSED6_BILINEAR_INSTANCES(EDS_INST_EQ_) }
    else { EDS_FUSED6_MAIN_INSTANCES(EDS_INST_EQ_) }
#undef EDS_INST_EQ_
#undef EDS_INST_EQ6_
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the reference problem (REF12): eds_fused12_kernel<S, T, CAP, NC, K, Q>
// ---------------------------------------------------------------------------------------------------------------------------------
struct EdsRef12In {
    int maxN, count;
    int bicubic, nc;        // cfg.sampling == bicubic; cfg.nc (the PhotometricErrorNC residual)
    int H;
    int retry;
    int nb;                 // residual blocks (cfg.num_blocks; 0 reads as 1): the full-cache shape holds the sums of one
};
#define EDS_RULE_HALF_CAP 736          // ... and of the paired slim shape (256 threads, two alignments per CU, 736 of the points cached)
#define EDS_RULE_FULL_CAP 2000         // = the CAP of the full-cache instantiations of eds_fused12_kernel (512 threads, one alignment per CU)
static inline bool eds_ref12_full_fits(const EdsRef12In& in) { return in.bicubic && !in.nc && in.nb <= 1 && in.maxN <= EDS_RULE_FULL_CAP && in.H < 8000; }
struct EdsRef12Plan {
    int wide, wants_team;               // begin
    int team, quad, strips_eligible;    // team
    int S, T, CAP, NC, K, Q;            // finish
    int G;                              // candidate groups (seventh template argument): G x K workgroups per alignment
};

static inline void eds_ref12_plan_begin(const EdsKnobs& kn, const EdsRef12In& in, EdsRef12Plan& p) {
    memset(&p, 0, sizeof(p));
    // Two shapes of the same kernel.  Up to one workgroup per CU (count <= 256) an alignment gets the whole CU: 512 threads, 88 KB
    // patch cache; beyond, 256-thread workgroups with a small cache so that TWO alignments share a CU and one's solver phase overlaps
    // the other's point phase
    bool wide = in.count <= 256;
    if (kn.ref12_kernel == 1 || kn.ref12_kernel == 3) wide = true; else if (kn.ref12_kernel == 2 || kn.ref12_kernel == 4) wide = false;
    p.wide = wide;
    p.wants_team = wide && !in.nc && in.maxN > 512 && in.count <= EDS_RULE_TEAM12_SLOTS && !in.retry;
}

static inline void eds_ref12_plan_team(const EdsKnobs& kn, const EdsRef12In& in, int team_ok, int cooldown_active, EdsRef12Plan& p) {
    int team = 1;
    if (team_ok) {
        team = (in.count <= 64 && in.maxN > 1024) ? 4 : 2;
        // a handful of alignments: 8 CUs each — up to 8 of them; 9 .. 16 of at most 2 048 points run faster on 4 CUs x 4 candidate groups than
        // on 8 x 2 (round 5, tools/bench_ref12_b16.py: 110.5 / 117.6 against 117.8 / 122.9 us at 9 / 16 alignments)
        if (in.count <= (in.maxN <= 2048 && !kn.ref12_groups ? 8 : 16) && in.maxN > 1024) team = 8;
        if (in.maxN > 8192 && in.count * 16 <= EDS_RULE_TEAM12_MEMBERS) team = 16;                // the finer pyramid levels: ~1 000 points per CU
        else if (in.maxN > 4096 && in.count * 8 <= EDS_RULE_TEAM12_MEMBERS) team = 8;
    }
    if (kn.ref12_team) {
        const int v = kn.ref12_team;
        if (v == 1 || ((v == 2 || v == 4 || v == 8 || v == 16) && p.wide && !in.nc && in.count <= EDS_RULE_TEAM12_SLOTS &&
                       in.count * v <= EDS_RULE_TEAM12_MEMBERS && !in.retry && !cooldown_active)) team = v;
    }
    if (eds_ref12_force_feasible(kn, in) && (kn.force12[4] == 1 || (team_ok && !cooldown_active) || kn.force12[4] == team)) {
        p.team = kn.force12[4];
        p.quad = kn.force12[5] != 0;
        p.strips_eligible = kn.force12[5] == 2;
        return;
    }
    p.team = team;
    // the quad-cooperative gather pays once the gather, not the instruction stream, bounds the point phase: on the tiles from 1 024
    // alignments, on the strips (one load per row, no shift) from 64
    const bool want_strips = !in.nc && !kn.layout_tiles;
    bool quad = in.bicubic && in.count >= (want_strips ? 64 : 1024);
    if (kn.gather) quad = in.bicubic && kn.gather == 1;
    quad = quad && in.H < 8000;                                       // 13-bit row field of the packed origins
    p.quad = quad;
    // teams of 8 and 16 have no strip instantiation: their frames are neither converted nor marked as solved-on-strips
    p.strips_eligible = quad && want_strips && team <= 4;
}

static inline bool eds_ref12_force_feasible(const EdsKnobs& kn, const EdsRef12In& in) {
    if (!kn.force12_set) return false;
    const int S = kn.force12[0], T = kn.force12[1], CAP = kn.force12[2], NC = kn.force12[3], K = kn.force12[4], Q = kn.force12[5];
    if (K < 1 || !eds_fused12_instance_exists(S, T, CAP, NC, K, Q, 1)) return false;
    if (S != (in.bicubic ? 0 : 1) || (NC != 0) != (in.nc != 0)) return false;
    if (K > 1 && (in.nc || in.retry || in.count > EDS_RULE_TEAM12_SLOTS || in.count * K > EDS_RULE_TEAM12_MEMBERS)) return false;
    if (Q != 0 && in.H >= 8000) return false;
    if ((CAP == EDS_RULE_FULL_CAP || CAP == EDS_RULE_HALF_CAP) && !eds_ref12_full_fits(in)) return false;          // the slim shapes: one residual block
    return true;
}

static inline void eds_ref12_plan_finish(const EdsKnobs& kn, const EdsRef12In& in, int strips, EdsRef12Plan& p) {
    strips = strips && p.strips_eligible;
    // Candidate groups (the latency regime proper: the launch leaves CUs idle): G teams evaluate G prepared steps at once — as many as
    // give every workgroup a CU of its own (measured, tools/check_groups12.py: 8 alignments x 8 CUs x 4 groups on all 256 CUs 114.8 us,
    // 2 groups 119.8, none 122.0; unlike the pose-only kernel, whose rounds are short enough for a full chip to cost more than it gives).
    // Only where a member's slice is at most 512 points (its patch cache in this shape; the kept residuals stay in registers) and an
    // instantiation exists.
    auto groups_for = [&](EdsRef12Plan& q) {
        q.G = 1;
        if (q.K <= 1 || q.NC || q.T != 512 || (long long)in.maxN > 512ll * q.K) return;
        int g = 1;
        for (int c = 4; c >= 2; c >>= 1)
            if (in.count * q.K * c <= kn.cus) { g = c; break; }
        if (kn.ref12_groups) g = kn.ref12_groups;
        if (g > 1 && (in.count * q.K * g > EDS_RULE_TEAM12_MEMBERS || !eds_fused12_instance_exists(q.S, q.T, 512, q.NC, q.K, q.Q, g))) g = 1;
        q.G = g;
        if (g > 1) q.CAP = 512;
    };
    if (eds_ref12_force_feasible(kn, in) && (kn.force12[5] != 2 || strips) && p.team == kn.force12[4]) {
        p.S = kn.force12[0]; p.T = kn.force12[1]; p.CAP = kn.force12[2]; p.NC = kn.force12[3]; p.K = kn.force12[4]; p.Q = kn.force12[5];
        p.G = 1;
        if (kn.ref12_groups > 1) groups_for(p);          // (a forced instantiation forms groups only when asked to)
        return;
    }
    const bool want_strips = !in.nc && !kn.layout_tiles;
    bool quad = p.quad;
    if (want_strips && !strips && in.count < 1024) quad = false;     // (no copies to read: the tiles' rule)
    p.K = p.team; p.NC = 0;
    p.S = in.bicubic ? 0 : 1;
    if (p.team > 1 || p.wide) { p.T = 512; p.CAP = 1408; } else { p.T = 256; p.CAP = 320; }
    if (p.team >= 8 || !in.bicubic) p.Q = 0;
    else p.Q = strips ? 2 : (quad ? 1 : 0);
    if (p.team == 1 && !(in.bicubic && strips)) p.NC = in.nc ? 1 : 0;
    // EDS_REF12_KERNEL=full (round 6, A/B knob): one alignment per CU with a cache slot for every point (quad gather, tiles or strips)
    if (kn.ref12_kernel == 3 && p.team == 1 && eds_ref12_full_fits(in)) { p.T = 512; p.CAP = EDS_RULE_FULL_CAP; p.Q = strips ? 2 : 1; p.NC = 0; }
    // The paired shape with 736 cache slots per alignment (round 6): what the batch launches for the reference problem (one residual block,
    // up to 2 000 points) on frames that are NEW for the solve — the tile gather runs at the fabric's line-fill rate, and the cache takes
    // 13 % of its requests away (4 096 alignments: 194 M -> 169 M, 3.64 -> 3.48 ms, profiles/r06_ref12_shapes.txt).  On the strip copies the
    // same cache LOSES (2.68 -> 3.07 ms: that gather is bound by its instruction stream, and probing costs instructions): EDS_REF12_KERNEL=half
    // forces it there too, =paired keeps the cache-less shape everywhere.
    if (p.team == 1 && !p.wide && eds_ref12_full_fits(in) && p.T == 256 && ((p.Q == 1 && kn.ref12_kernel != 2) || (kn.ref12_kernel == 4 && p.Q != 0)))
        { p.CAP = EDS_RULE_HALF_CAP; p.NC = 0; }
    groups_for(p);
}

// (the full-cache shape fills the CU's LDS to 40 bytes: the diagnostic builds, whose scratch structures carry stamps, leave it out)
#ifdef EDS_FUSED_STAMPS
#define EDS_FUSED12_FULL_INSTANCES(X)
#else
#define EDS_FUSED12_FULL_INSTANCES(X) X(0, 512, 2000, false, 1, 1) X(0, 512, 2000, false, 1, 2)
#endif
#define EDS_FUSED12_INSTANCES(X)                                                                                                      \
    X(0, 512, 1408, false, 16, 0) X(1, 512, 1408, false, 16, 0) X(0, 512, 1408, false, 8, 0) X(1, 512, 1408, false, 8, 0)           \
    X(0, 512, 1408, false, 4, 2) X(0, 512, 1408, false, 4, 1) X(0, 512, 1408, false, 4, 0) X(1, 512, 1408, false, 4, 0)             \
    X(0, 512, 1408, false, 2, 2) X(0, 512, 1408, false, 2, 1) X(0, 512, 1408, false, 2, 0) X(1, 512, 1408, false, 2, 0)             \
    X(0, 512, 1408, false, 1, 2) X(0, 512, 1408, false, 1, 1) X(0, 512, 1408, true, 1, 1) X(0, 512, 1408, false, 1, 0)              \
    X(0, 512, 1408, true, 1, 0) X(1, 512, 1408, false, 1, 0) X(1, 512, 1408, true, 1, 0)                                            \
    X(0, 256, 320, false, 1, 2) X(0, 256, 320, false, 1, 1) X(0, 256, 320, true, 1, 1) X(0, 256, 320, false, 1, 0)                  \
    X(0, 256, 320, true, 1, 0) X(1, 256, 320, false, 1, 0) X(1, 256, 320, true, 1, 0)                                               \
    EDS_FUSED12_FULL_INSTANCES(X) X(0, 256, 736, false, 1, 1) X(0, 256, 736, false, 1, 2)

// ... and the candidate-group instantiations X(S, T, CAP, NC, K, Q, G), G > 1: a member's slice is at most 512 points here, so is its patch
// cache (CAP = 512) — the LDS that leaves holds the G sets of sums of a round
#define EDS_FUSED12_GROUP_INSTANCES(X)                                                                                                \
    X(0, 512, 512, false, 8, 0, 2) X(0, 512, 512, false, 8, 0, 4) X(1, 512, 512, false, 8, 0, 2) X(1, 512, 512, false, 8, 0, 4)   \
    X(0, 512, 512, false, 4, 0, 2) X(1, 512, 512, false, 4, 0, 2) X(0, 512, 512, false, 4, 0, 4) X(1, 512, 512, false, 4, 0, 4)

static inline bool eds_fused12_instance_exists(int S, int T, int CAP, int NC, int K, int Q, int G) {
#define EDS_INST_EQ_(s, t, c, n, k, q) if (S == s && T == t && CAP == c && (NC != 0) == n && K == k && Q == q) return true;
#define EDS_INST_EQ7_(s, t, c, n, k, q, g) if (S == s && T == t && CAP == c && (NC != 0) == n && K == k && Q == q && G == g) return true;
    if (G > 1) { EDS_FUSED12_GROUP_INSTANCES(EDS_INST_EQ7_) return false; }
    EDS_FUSED12_INSTANCES(EDS_INST_EQ_)
#undef EDS_INST_EQ_
#undef EDS_INST_EQ7_
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// strip copies: how many row phases fit the budget (eds_strips.hip).  copy_bytes_per_slot = 2 x one copy of one frame.
// returns 4, 2, 1, or 0 (none: remembered on the handle, not retried on every solve)
// ---------------------------------------------------------------------------------------------------------------------------------
static inline int eds_strips_phases_for_budget(int wanted_phases, unsigned long long slots, unsigned long long two_copies_bytes,
                                               unsigned long long free_bytes, int budget_pct) {
    const unsigned long long budget = free_bytes / 100ull * (unsigned long long)budget_pct;
    for (int ph = wanted_phases; ph >= 1; ph >>= 1)
        if (slots * (unsigned long long)ph * two_copies_bytes <= budget) return ph;
    return 0;
}
