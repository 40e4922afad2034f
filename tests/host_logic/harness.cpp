// CPU harness for the PRODUCT's host-side logic (slam-eds_amd/csrc/eds_math.hpp, eds_solver.hpp):
// the state machines are fed with reduced sums computed by the oracle's evaluator, so that the
// solver logic can be checked against the oracle's own solvers without a GPU.  Test-only code.
#include <cstring>
#include <string>
#include <vector>

#include "../../oracle/eds_oracle.hpp"
#include "../../slam-eds_amd/csrc/eds_math.hpp"
#include "../../slam-eds_amd/csrc/eds_layout.hpp"
#include "../../slam-eds_amd/csrc/eds_launch_rule.hpp"
#include "../../slam-eds_amd/csrc/eds_solver.hpp"

using namespace eds_oracle;

extern "C" {

struct hl_problem {
    int32_t N, H, W, _pad;
    const double *grad, *norm_coord, *idp, *weights, *frame;
    double fx, fy, cx, cy;
};
static Problem to_pb(const hl_problem* p) {
    Problem pb; pb.N = p->N; pb.H = p->H; pb.W = p->W; pb.grad = p->grad; pb.norm_coord = p->norm_coord;
    pb.idp = p->idp; pb.weights = p->weights; pb.frame = p->frame; pb.fx = p->fx; pb.fy = p->fy; pb.cx = p->cx; pb.cy = p->cy;
    return pb;
}

// product Solver12 driven by oracle-evaluated sums
int hl_solver12_run(const hl_problem* p, int sampling, int nb, int loss_type, double loss_a, int max_iters, double ftol,
                    double gtol, double ptol, double* px, double* qx, double* vx, int32_t* out5, double* costs2) {
    Problem pb = to_pb(p);
    SolveConfig cfg; cfg.sampling = sampling; cfg.num_blocks = nb;
    edss::Solver12* sv = new edss::Solver12();
    edss::Sums12* S = new edss::Sums12();
    sv->init(max_iters, loss_type, loss_a, ftol, gtol, ptol, px, qx, vx);
    Evaluation ev;
    int passes = 0;
    while (!sv->done) {
        evaluate(pb, cfg, sv->cp, sv->cq, sv->cv, true, &ev);
        ++passes;
        S->nb = nb;
        for (int k = 0; k < nb; ++k) {
            int start, n; block_range(pb.N, nb, k, &start, &n);
            for (int i = 0; i < 144; ++i) S->H[k][i] = 0; for (int i = 0; i < 12; ++i) S->g[k][i] = 0; S->s[k] = 0;
            for (int i = start; i < start + n; ++i) {
                const double* J = &ev.jac_local_raw[(size_t)i * 12]; const double r = ev.raw_residuals[i];
                for (int a = 0; a < 12; ++a) { S->g[k][a] += J[a] * r; for (int b = 0; b < 12; ++b) S->H[k][12 * a + b] += J[a] * J[b]; }
                S->s[k] += r * r;
            }
        }
        sv->on_eval(*S);
    }
    const bool ok = sv->termination != edss::TERM_FAILURE;
    if (ok) { std::memcpy(px, sv->best_p, 24); std::memcpy(qx, sv->best_q, 32); std::memcpy(vx, sv->best_v, 48); }
    out5[0] = sv->termination; out5[1] = sv->num_successful; out5[2] = sv->num_unsuccessful; out5[3] = passes; out5[4] = sv->iteration;
    costs2[0] = sv->initial_cost; costs2[1] = sv->minimum_cost;
    delete sv; delete S;
    return ok ? 0 : -1;
}

// product Solver6 driven by oracle-evaluated sums
int hl_solver6_run(const hl_problem* p, int sampling, int nb, int damped, int max_iters, double lambda0, double huber_tau,
                   double* px, double* qx, const double* vx, double* inc, double* costs, int32_t* acc, int32_t* out3) {
    Problem pb = to_pb(p);
    SolveConfig cfg; cfg.sampling = sampling; cfg.num_blocks = nb;
    edss::Solver6* sv = new edss::Solver6();
    sv->init(damped, max_iters, lambda0, px, qx);
    Pose6Eval ev;
    int passes = 0;
    while (!sv->done) {
        pose6_eval(pb, cfg, sv->cp, sv->cq, vx, huber_tau, &ev);
        ++passes;
        edss::Sums6 S; std::memcpy(S.H, ev.H, sizeof(S.H)); std::memcpy(S.b, ev.b, sizeof(S.b)); S.cost = ev.cost;
        sv->on_eval(S);
    }
    std::memcpy(px, sv->p, 24); std::memcpy(qx, sv->q, 32);
    for (int i = 0; i < sv->ntrace; ++i) { std::memcpy(inc + 6 * i, sv->tr_xi[i], 48); costs[i] = sv->tr_cost[i]; acc[i] = sv->tr_acc[i]; }
    out3[0] = sv->ntrace; out3[1] = passes; out3[2] = sv->failed;
    delete sv;
    return 0;
}

void hl_se3_left_update(const double* xi, double* t, double* q) { edsm::se3_left_update(xi, t, q); }
void hl_state_plus12(const double* p, const double* q, const double* v, const double* d, double* po, double* qo, double* vo) { edsm::state_plus12(p, q, v, d, po, qo, vo); }
int hl_cholesky(int n, const double* A, const double* b, double* x) { return edsm::cholesky_solve(n, A, b, x) ? 1 : 0; }
void hl_quat_to_R(const double* q, double* R) { edsm::quat_to_R(q, R); }
void hl_fill_pose_block(const double* p, const double* q, const double* v, const double* G, int nb, double* pb) { edsm::fill_pose_block(p, q, v, G, nb, pb); }
void hl_loss_eval(int type, double a, double s, double* out2) { edss::loss_eval(type, a, s, &out2[0], &out2[1]); }
int hl_pose_stride(void) { return EDS_POSE_STRIDE; }

// Frame allocation of the product (eds_layout.hpp): walks every logical pixel of the padded + margin range through
// eds_frame_index and reports how many elements of the Hp x Wp allocation were not hit exactly once, plus the index of
// logical pixel (r, c) for spot checks.
int hl_frame_layout(int H, int W, int tiled, int* Hp_out, int* Wp_out, int r, int c, long long* index_rc) {
    const int Hp = eds_frame_extent(H), Wp = eds_frame_extent(W);
    *Hp_out = Hp; *Wp_out = Wp;
    std::vector<int> hits((size_t)Hp * Wp, 0);
    int bad = 0;
    for (int y = -EDS_FRAME_MARGIN; y < Hp - EDS_FRAME_MARGIN; ++y)
        for (int x = -EDS_FRAME_MARGIN; x < Wp - EDS_FRAME_MARGIN; ++x) {
            const size_t o = eds_frame_index(y, x, Wp, tiled);
            if (o >= hits.size()) ++bad; else ++hits[o];
        }
    for (int h : hits) bad += (h != 1);
    *index_rc = (long long)eds_frame_index(r, c, Wp, tiled);
    return bad;
}

// Strip copies of the frames (eds_layout.hpp, round 3).  Builds the copies of an Hp x Wp allocation whose element (r, c) holds the
// value r * Wp + c with the rule of the conversion kernel (eds_strips.hip: copy 2 p + cc holds allocation row r at position r - p,
// strips cut at column 4 cc), then walks every patch position a kernel can ask for (first row ra in [1, Hp - 4], first column ca in
// [1, Wp - 4]) through eds_strips_row_offset and checks the 16 taps; returns the number of wrong taps and, in stats, the number of
// patches and how many of them start on a 32 * phases byte boundary / lie inside ONE 128-byte line.
int hl_strips_decide(int policy, int stale, int fresh, int count) { return eds_strips_decide(policy, stale, fresh, count); }

int hl_strips_layout(int Hp, int Wp, int phases, long long* stats) {
    const int NS = eds_strips_count(Wp);
    const size_t copy_elems = eds_strips_copy_elems(Hp, Wp);
    std::vector<float> alloc((size_t)Hp * Wp), strips((size_t)2 * phases * copy_elems, -1.0f);
    for (int r = 0; r < Hp; ++r)
        for (int c = 0; c < Wp; ++c) alloc[(size_t)r * Wp + c] = (float)(r * Wp + c);
    for (int p = 0; p < phases; ++p)
        for (int cc = 0; cc < 2; ++cc)
            for (int sidx = 0; sidx < NS; ++sidx)
                for (int pos = 0; pos < Hp; ++pos)
                    for (int j = 0; j < 8; ++j) {
                        int row = pos + p; if (row > Hp - 1) row = Hp - 1;
                        int col = 8 * sidx + 4 * cc + j; if (col > Wp - 1) col = Wp - 1;      // (the kernel clamps whole 4-column pieces: never sampled either way)
                        strips[(size_t)(2 * p + cc) * copy_elems + ((size_t)sidx * Hp + pos) * 8 + j] = alloc[(size_t)row * Wp + col];
                    }
    long long bad = 0, patches = 0, aligned = 0, one_line = 0;
    const unsigned copy_bytes = (unsigned)(copy_elems * 4);
    for (int ra = 1; ra <= Hp - 4; ++ra)
        for (int ca = 1; ca <= Wp - 4; ++ca) {
            const unsigned off = eds_strips_row_offset(ra, ca, Hp, copy_bytes, phases);
            ++patches;
            aligned += (off & ~31u) % (32u * phases) == 0 ? 1 : 0;          // the 32-byte strip row the patch starts in
            one_line += ((off & ~31u) / 128u == ((off & ~31u) + 127u) / 128u) ? 1 : 0;
            for (int k = 0; k < 4; ++k)
                for (int j = 0; j < 4; ++j) {
                    const size_t e = (size_t)(off / 4) + 8 * k + j;
                    if (e >= strips.size() || strips[e] != alloc[(size_t)(ra + k) * Wp + ca + j]) ++bad;
                }
        }
    stats[0] = patches; stats[1] = aligned; stats[2] = one_line;
    return (int)(bad > 2000000000ll ? 2000000000ll : bad);
}

// The launch rule of the product (eds_launch_rule.hpp).  knobs: "NAME=value;NAME=value" (the names of the environment variables).
// in7 = {maxN, count, bicubic, iters, lm6, huber, H}; flags: bit 0 retry, bit 1 the time-out policy allows teams, bit 2 a cool-down is
// running, bit 3 the strip copies are (or can be made) current.
// out = {kind, S, P, T, Q, K, bilinear_tu, wide_members, threads, ppt, strips_eligible, wants_team, instance_exists, note_T, G}
static int parse_knobs(const char* spec, EdsKnobs* kn) {
    std::string s(spec ? spec : "");
    size_t a = 0;
    while (a < s.size()) {
        size_t b = s.find(';', a); if (b == std::string::npos) b = s.size();
        const std::string item = s.substr(a, b - a);
        const size_t eq = item.find('=');
        if (eq != std::string::npos && item.substr(0, eq) == "cus") { kn->cus = std::atoi(item.substr(eq + 1).c_str()); a = b + 1; continue; }     // (not a knob: the device's CU count)
        if (eq != std::string::npos && eds_knobs_set(kn, item.substr(0, eq).c_str(), item.substr(eq + 1).c_str()) != 0) return -1;
        a = b + 1;
    }
    return 0;
}
int hl_lm6_rule(const char* knobs, const int32_t* in7, int flags, int32_t* out) {
    EdsKnobs kn;
    if (parse_knobs(knobs, &kn)) return -1;
    const EdsLm6In in{in7[0], in7[1], in7[2], in7[3], in7[4], in7[5], in7[6], (flags & 1) ? 1 : 0};
    EdsLm6Plan p;
    eds_lm6_plan_begin(kn, in, p);
    const int team_ok = p.wants_team && (flags & 2);
    eds_lm6_plan_team(kn, in, team_ok, (flags & 4) ? 1 : 0, p);
    const int strips = p.strips_eligible && (flags & 8);
    eds_lm6_plan_finish(kn, in, strips, p);
    const int exists = p.kind == EDS_K6_STREAM ? 1 : (eds_fused6_instance_exists(p.S, p.P, p.T, p.Q, p.K, p.bilinear_tu, p.G) ? 1 : 0);
    const int32_t o[15] = {p.kind, p.S, p.P, p.T, p.Q, p.K, p.bilinear_tu, p.wide_members, p.threads, p.ppt, p.strips_eligible, p.wants_team, exists, p.note_T, p.G};
    std::memcpy(out, o, sizeof(o));
    return 0;
}
// in5 = {maxN, count, bicubic, nc, H}; flags as above.  out = {S, T, CAP, NC, K, Q, strips_eligible, wants_team, instance_exists, G}
int hl_ref12_rule(const char* knobs, const int32_t* in5, int flags, int32_t* out) {
    EdsKnobs kn;
    if (parse_knobs(knobs, &kn)) return -1;
    const EdsRef12In in{in5[0], in5[1], in5[2], in5[3], in5[4], (flags & 1) ? 1 : 0, (flags >> 8) & 0xff};       // flags bits 8..15: residual blocks (0 reads as 1)
    EdsRef12Plan p;
    eds_ref12_plan_begin(kn, in, p);
    const int team_ok = p.wants_team && (flags & 2);
    eds_ref12_plan_team(kn, in, team_ok, (flags & 4) ? 1 : 0, p);
    const int strips = p.strips_eligible && (flags & 8);
    eds_ref12_plan_finish(kn, in, strips, p);
    const int32_t o[10] = {p.S, p.T, p.CAP, p.NC, p.K, p.Q, p.strips_eligible, p.wants_team, eds_fused12_instance_exists(p.S, p.T, p.CAP, p.NC, p.K, p.Q, p.G) ? 1 : 0, p.G};
    std::memcpy(out, o, sizeof(o));
    return 0;
}
// the environment as eds_trk_create reads it: the first variable whose value its knob refuses (nullptr: none)
const char* hl_knobs_from_env(void) { EdsKnobs kn; return eds_knobs_from_env(&kn); }
int hl_knob_set(const char* name, const char* value) { EdsKnobs kn; return eds_knobs_set(&kn, name, value); }
int hl_strips_phases_for_budget(int wanted, long long slots, long long two_copies_bytes, long long free_bytes, int pct) {
    return eds_strips_phases_for_budget(wanted, (unsigned long long)slots, (unsigned long long)two_copies_bytes, (unsigned long long)free_bytes, pct);
}

}  // extern "C"
