// CPU harness for the PRODUCT's host-side logic (slam-eds_amd/csrc/eds_math.hpp, eds_solver.hpp):
// the state machines are fed with reduced sums computed by the oracle's evaluator, so that the
// solver logic can be checked against the oracle's own solvers without a GPU.  Test-only code.
#include <cstring>
#include <vector>

#include "../../oracle/eds_oracle.hpp"
#include "../../slam-eds_amd/csrc/eds_math.hpp"
#include "../../slam-eds_amd/csrc/eds_layout.hpp"
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

}  // extern "C"
