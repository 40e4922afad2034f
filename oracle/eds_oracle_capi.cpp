// C entry points over eds_oracle.hpp so tests/ and bench.py's cpu_baseline leg can
// drive the CPU oracle through ctypes.  TEST INFRASTRUCTURE ONLY — see the header
// of eds_oracle.hpp (parity unpinned; never linked into the product library).
#include "eds_oracle.hpp"
#include "eds_cpu_fast.hpp"

#include <atomic>
#include <chrono>
#include <thread>

using namespace eds_oracle;

extern "C" {

struct eds_oracle_problem {
    int32_t N, H, W, _pad;
    const double* grad;         // N x 2
    const double* norm_coord;   // N x 2
    const double* idp;          // N
    const double* weights;      // N
    const double* frame;        // H x W
    double fx, fy, cx, cy;
};

struct eds_oracle_config {
    int32_t sampling;       // 0 bicubic (reference), 1 bilinear
    int32_t nc;             // 1: PhotometricErrorNC
    int32_t num_blocks;     // reference num_threads
    int32_t loss_type;      // 0 none, 1 Huber, 2 Cauchy
    double loss_param;
    int32_t max_num_iterations;
    int32_t eval_threads;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
};

struct eds_oracle_summary {
    int32_t termination, num_successful_steps, num_unsuccessful_steps, num_residuals;
    double initial_cost, final_cost, seconds;
};

static Problem to_problem(const eds_oracle_problem* p) {
    Problem pb;
    pb.N = p->N; pb.H = p->H; pb.W = p->W;
    pb.grad = p->grad; pb.norm_coord = p->norm_coord; pb.idp = p->idp; pb.weights = p->weights; pb.frame = p->frame;
    pb.fx = p->fx; pb.fy = p->fy; pb.cx = p->cx; pb.cy = p->cy;
    return pb;
}
static SolveConfig to_config(const eds_oracle_config* c) {
    SolveConfig cfg;
    cfg.sampling = c->sampling; cfg.nc = c->nc != 0; cfg.num_blocks = c->num_blocks > 0 ? c->num_blocks : 1;
    cfg.loss_type = c->loss_type; cfg.loss_param = c->loss_param;
    cfg.max_num_iterations = c->max_num_iterations; cfg.eval_threads = c->eval_threads > 0 ? c->eval_threads : 1;
    cfg.function_tolerance = c->function_tolerance; cfg.gradient_tolerance = c->gradient_tolerance;
    cfg.parameter_tolerance = c->parameter_tolerance;
    return cfg;
}

// 12-column reference problem.  Any output pointer may be null.
int eds_oracle_eval12(const eds_oracle_problem* p, const eds_oracle_config* c, const double* px, const double* qx,
                      const double* vx, double* r_raw, double* r_corrected, double* jac_global /*N x 13*/,
                      double* jac_local_raw /*N x 12*/, double* jac_local /*N x 12*/, double* cost, double* gradient12) {
    Problem pb = to_problem(p); SolveConfig cfg = to_config(c);
    Evaluation ev;
    const bool wj = jac_global || jac_local_raw || jac_local || gradient12;
    evaluate(pb, cfg, px, qx, vx, wj, &ev);
    const size_t N = pb.N;
    if (r_raw) std::memcpy(r_raw, ev.raw_residuals.data(), N * 8);
    if (r_corrected) std::memcpy(r_corrected, ev.residuals.data(), N * 8);
    if (jac_global) std::memcpy(jac_global, ev.jac_global.data(), N * 13 * 8);
    if (jac_local_raw) std::memcpy(jac_local_raw, ev.jac_local_raw.data(), N * 12 * 8);
    if (jac_local) std::memcpy(jac_local, ev.jac_local.data(), N * 12 * 8);
    if (cost) *cost = ev.cost;
    if (gradient12) std::memcpy(gradient12, ev.gradient, 12 * 8);
    return ev.ok ? 0 : -1;
}

// ceres::Solve restatement.  p,q,v are in/out (left untouched when not usable).
int eds_oracle_solve_lm(const eds_oracle_problem* p, const eds_oracle_config* c, double* px, double* qx, double* vx,
                        eds_oracle_summary* out) {
    Problem pb = to_problem(p); SolveConfig cfg = to_config(c);
    SolveSummary s;
    const auto t0 = std::chrono::steady_clock::now();
    solve_lm(pb, cfg, px, qx, vx, &s);
    const auto t1 = std::chrono::steady_clock::now();
    if (out) {
        out->termination = s.termination; out->num_successful_steps = s.num_successful_steps;
        out->num_unsuccessful_steps = s.num_unsuccessful_steps; out->num_residuals = s.num_residuals;
        out->initial_cost = s.initial_cost; out->final_cost = s.final_cost;
        out->seconds = std::chrono::duration<double>(t1 - t0).count();
    }
    return s.usable() ? 0 : -1;
}

// Pose-only 6-DoF evaluation (autodiff wrt the SE(3) left perturbation).
int eds_oracle_pose6_eval(const eds_oracle_problem* p, const eds_oracle_config* c, const double* px, const double* qx,
                          const double* vx, double huber_tau, double* r, double* J /*N x 6*/, double* hw, double* H36,
                          double* b6, double* cost) {
    Problem pb = to_problem(p); SolveConfig cfg = to_config(c);
    Pose6Eval ev;
    pose6_eval(pb, cfg, px, qx, vx, huber_tau, &ev);
    const size_t N = pb.N;
    if (r) std::memcpy(r, ev.r.data(), N * 8);
    if (J) std::memcpy(J, ev.J.data(), N * 6 * 8);
    if (hw) std::memcpy(hw, ev.hw.data(), N * 8);
    if (H36) std::memcpy(H36, ev.H, 36 * 8);
    if (b6) std::memcpy(b6, ev.b, 6 * 8);
    if (cost) *cost = ev.cost;
    return 0;
}

// `iters` Gauss-Newton iterations; returns the number executed.  p,q in/out.
int eds_oracle_pose6_gn(const eds_oracle_problem* p, const eds_oracle_config* c, double* px, double* qx,
                        const double* vx, double huber_tau, int iters, double* increments /*iters x 6*/,
                        double* costs /*iters*/, double* seconds) {
    Problem pb = to_problem(p); SolveConfig cfg = to_config(c);
    const auto t0 = std::chrono::steady_clock::now();
    const int n = pose6_gauss_newton(pb, cfg, px, qx, vx, huber_tau, iters, increments, costs);
    const auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    return n;
}

// DSO-style damped Gauss-Newton on the 6-DoF problem; returns iterations executed.
int eds_oracle_pose6_lm(const eds_oracle_problem* p, const eds_oracle_config* c, double* px, double* qx,
                        const double* vx, double huber_tau, int iters, double lambda0, double* increments,
                        double* costs, int32_t* accepted, double* initial_cost, double* seconds) {
    Problem pb = to_problem(p); SolveConfig cfg = to_config(c);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> acc(iters > 0 ? iters : 1, 0);
    const int n = pose6_lm(pb, cfg, px, qx, vx, huber_tau, iters, lambda0, increments, costs, acc.data(), initial_cost);
    const auto t1 = std::chrono::steady_clock::now();
    if (accepted) for (int i = 0; i < n; ++i) accepted[i] = acc[i];
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    return n;
}

// Tracker::getLossParams; `residuals` is reordered in place exactly like the reference.
// Optimised CPU variant of pose6_lm (eds_cpu_fast.hpp): fp32 sampling, analytic rows, SoA — bench.py's second CPU baseline.
// prepare() is the counterpart of the GPU library's set_keyframe / set_event_frame (inputs converted once, outside the timed
// solves); the handle is freed with eds_oracle_fast_free.
void* eds_oracle_fast_prepare(const eds_oracle_problem* p, const double* vx) {
    Problem pb = to_problem(p);
    eds_cpu_fast::Prepared* P = new eds_cpu_fast::Prepared();
    eds_cpu_fast::prepare(pb, vx, P);
    return P;
}
void eds_oracle_fast_free(void* h) { delete static_cast<eds_cpu_fast::Prepared*>(h); }
int eds_oracle_fast_lm6(const void* h, double* px, double* qx, int iters, double lambda0, int32_t* accepted, double* seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    const int n = eds_cpu_fast::lm6(*static_cast<const eds_cpu_fast::Prepared*>(h), px, qx, iters, lambda0, accepted);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return n;
}

int eds_oracle_fast_lm6_scalar(const void* h, double* px, double* qx, int iters, double lambda0, int32_t* accepted) {
    return eds_cpu_fast::lm6(*static_cast<const eds_cpu_fast::Prepared*>(h), px, qx, iters, lambda0, accepted, true);
}
int eds_oracle_fast_is_vectorised(void) {
#ifdef EDS_CPU_FAST_AVX2
    return 1;
#else
    return 0;
#endif
}

// All-core throughput of the pose-only solve for bench.py's cpu_baseline leg, WITHOUT the interpreter in the loop (round 4 drove the
// threads from a Python ThreadPoolExecutor with a barrier every cores x 4 tasks: 256 threads reached 10x one core).  `threads`
// std::threads are started once; each takes solves off a shared atomic counter — solve k works on problem k % nprob from its start
// pose — until `budget_s` seconds have passed.  mode 0: the oracle's pose6_lm (Jet autodiff), 1: the optimised variant on handles of
// eds_oracle_fast_prepare.  Returns the number of solves; iterations / elapsed seconds through the pointers.
long long eds_oracle_bench_lm6(int nprob, const eds_oracle_problem* problems, const void* const* fast_handles, const eds_oracle_config* c,
                               const double* p0 /*nprob x 3*/, const double* q0 /*nprob x 4*/, const double* v0 /*nprob x 6*/, int iters,
                               double lambda0, int threads, double budget_s, int mode, long long* iterations, double* elapsed_s) {
    if (nprob < 1 || threads < 1 || iters < 0) return -1;
    std::atomic<long long> next{0}, its{0}, done{0};
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        std::vector<double> inc((size_t)std::max(iters, 1) * 6), costs(std::max(iters, 1));
        std::vector<int> acc(std::max(iters, 1));
        long long my_its = 0, my_done = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < budget_s) {
            const int k = (int)(next.fetch_add(1) % nprob);
            double p[3], q[4];
            std::memcpy(p, p0 + 3 * k, sizeof(p)); std::memcpy(q, q0 + 4 * k, sizeof(q));
            if (mode == 1) {
                my_its += eds_cpu_fast::lm6(*static_cast<const eds_cpu_fast::Prepared*>(fast_handles[k]), p, q, iters, lambda0, acc.data());
            } else {
                Problem pb = to_problem(&problems[k]); SolveConfig cfg = to_config(c);
                double c0 = 0.0;
                my_its += pose6_lm(pb, cfg, p, q, v0 + 6 * k, 0.0, iters, lambda0, inc.data(), costs.data(), acc.data(), &c0);
            }
            ++my_done;
        }
        its += my_its; done += my_done;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (iterations) *iterations = its.load();
    if (elapsed_s) *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return done.load();
}

double eds_oracle_loss_param(double* residuals, int n, int method, double current) {
    std::vector<double> r(residuals, residuals + n);
    const double tau = loss_param(r, method, current);
    std::memcpy(residuals, r.data(), sizeof(double) * n);
    return tau;
}

void eds_oracle_bicubic(const double* frame, int H, int W, double row, double col, double* out3) {
    Problem pb; pb.frame = frame; pb.H = H; pb.W = W;
    bicubic(pb, row, col, &out3[0], &out3[1], &out3[2]);
}
void eds_oracle_bilinear(const double* frame, int H, int W, double row, double col, double* out3) {
    Problem pb; pb.frame = frame; pb.H = H; pb.W = W;
    bilinear(pb, row, col, &out3[0], &out3[1], &out3[2]);
}

void eds_oracle_state_plus(const double* x13, const double* d12, double* out13) { state_plus(x13, d12, out13); }
void eds_oracle_quat_plus_jacobian(const double* q, double* J12) { quat_plus_jacobian(q, J12); }
void eds_oracle_unit_plus_jacobian(const double* v, double* J36) { unit_plus_jacobian(v, J36); }
void eds_oracle_quat_to_R(const double* q, double* R9) { quat_to_R<double>(q, R9); }
void eds_oracle_se3_exp(const double* xi, double* t, double* q) { se3_exp(xi, t, q); }
void eds_oracle_se3_log(const double* t, const double* q, double* xi) { se3_log(t, q, xi); }
void eds_oracle_se3_left_update(const double* xi, double* t, double* q) { se3_left_update(xi, t, q); }
double eds_oracle_se3_distance(const double* ta, const double* qa, const double* tb, const double* qb) { return se3_distance(ta, qa, tb, qb); }
void eds_oracle_loss_eval(int type, double a, double s, double* rho3) { loss_eval(type, a, s, rho3); }

}  // extern "C"
