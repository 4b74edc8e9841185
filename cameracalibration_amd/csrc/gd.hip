// gd.hip -- gradient-descent refinement (host loop, batched device passes).
//
// Restates the optimiser the reference instantiates at
// src/camera_calibration_optimization.cpp:428-445 and :486-505:
//   gdc::GradientDescent<double, Functor, gdc::WolfeBacktracking<double>>
// with central finite differences (the functor never fills the gradient),
// momentum 0.4, maxIt 10000, minGradientLength 1e-6, minStepLength 1e-6.
// gdcpp itself is an unpinned third-party header that is not part of the
// reference tree; the algorithm below is this library's own statement of it:
//
//   loop while it < maxIt and |g| >= minGrad and |step| >= minStep:
//       x -= step
//       f, g   = objective and central-difference gradient at x
//       alpha  = Wolfe backtracking from maxStep (x0.8 per trial) until
//                f(x-alpha g) <= f - c1 alpha g.g  and  g.g(x-alpha g) <= c2 g.g
//       step   = momentum*step + (1-momentum)*alpha*g
//
// Objective values come from batched calls into the exact objective; several
// independent descents advance in lock step and share those calls.  Three facts
// keep the batches small without changing a single iterate:
//   * the backtracking trials of one line search are known in advance
//     (alpha = 0.8^k), so the centres of a run of them are evaluated in one pass
//     and consumed in order with the sequential accept/continue logic;
//   * a trial that fails the Armijo test is rejected whatever its gradient, so
//     the 2*dim finite-difference points of a trial are evaluated only once its
//     centre has passed Armijo; the main stencil of the next iteration rides
//     along with them, computed as if the curvature test passes (it is dropped
//     if it does not).
//   * most backtracking trials fail the Armijo test by a wide margin, and a few pass it by one:
//     the centres of a line search are first scored by the folded FP32 pass, which comes with a
//     bound on its distance from the exact objective (ccal::eval_screen).  A trial whose FP32
//     value minus the bound is above the Armijo threshold fails whatever its exact value, one
//     whose FP32 value plus the bound is below it passes; only the trials in between are
//     evaluated exactly.  The decisions, and so the iterates, are those of the exact values.
//     The FP32 values are far closer to the exact ones than their bound, so they also say which
//     trial to expect to pass: the undecided centres up to it go into one exact request together
//     with its finite-difference points and the next main stencil (two dependent device steps
//     per iteration; the speculation is dropped when another trial passes).
//   * a lone descent tells the exact request what it expects every chain to sum to (the screening pass's
//     per-cloud sums of the trial; of the next main point too while the descent still moves far in one
//     iteration): the requests' chains are cut into parts whose units are guessed from an earlier
//     evaluation, and the guesses follow the expectation.  Expectations steer guesses, never values.
// Values past the accepted trial are dropped.  The evaluation count reported is
// that of the sequential algorithm (1 + 2*dim per trial it visits).
#include "ccal_internal.hpp"

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

namespace {

struct Machine {
    int D = 6;
    ccal_gd_opts o;
    double x[10], grad[10], step[10];
    double fval = 0.0, grad_len = 0.0, step_len = 0.0;
    int it = 0;
    int64_t n_evals = 0;
    int64_t n_requested = 0;   // evaluations asked of the device, dropped speculation included
    int64_t n_requests = 0;
    int64_t n_trials = 0, n_armijo_fail = 0, n_wolfe_only_fail = 0;
    // line search
    double alpha = 0.0, gg = 0.0, xn[10];
    int ls_it = 0;
    // TRIAL_F: objective values at the centres of the next trials (Armijo needs nothing else);
    // TRIAL_G: the +-eps/2 points of the one trial that passed Armijo (curvature test)
    // SCREEN:  FP32 values and bounds at the centres of the next trials (settles the clear cases)
    // TRIAL_FG: the centres of the undecided trials up to the one the FP32 values expect to pass,
    //          with that trial's +-eps/2 points (and the next main stencil) in the same request
    enum Phase { MAIN, SCREEN, TRIAL_F, TRIAL_FG, TRIAL_G, DONE } phase = DONE;
    bool screening = false;
    int fg_trials = 0;         // TRIAL_FG: centres in the request; the last one is the expected pass
    int screen_first = -1;     // TRIAL_FG / TRIAL_G straight after a screening request: that request's number of the trial at xn
    int64_t n_screen_requests = 0, n_screened_fail = 0, n_screened_pass = 0, n_expected_hits = 0, n_expected_misses = 0;
    // speculation: trials per request (first request of a line search / later ones)
    int last_trials = 3;
    bool first_request = true;
    static constexpr int kMaxSpeculated = 64;
    int64_t n_main_hits = 0;
    bool ride_along = true;   // off for the rest of a line search whose curvature test has failed once
    int chains_per_pose = 0, round_chains = 0;   // set for a single descent: clouds, compute units
    // Set on a COPY of a machine that looks ahead: the curvature test of the trial whose gradient points are
    // still being evaluated is taken to pass (it has, on every trial measured), so that the next iteration's
    // screening request can be issued early.  The original machine runs the test for real afterwards and
    // the look-ahead is kept only if both arrive at the same state.
    bool assume_curvature_passes = false;

    int stencil() const { return 1 + 2 * D; }

    // trials whose centres the next TRIAL_F request evaluates (>= 1)
    int planned_trials() const
    {
        // centres are cheap (one pose each) and a short request costs about as much as a
        // dozen of them, so speculate a few trials past the previous line search's count
        int want = first_request ? last_trials + 4 : 8;
        // the device works through one chain per (pose, cloud) in rounds of `round_chains`
        // workgroups: more centres are free up to the end of the round the request needs anyway
        if (round_chains > 0 && chains_per_pose > 0) {
            const int rounds = (want * chains_per_pose + round_chains - 1) / round_chains;
            const int fill = rounds * round_chains / chains_per_pose;
            if (fill > want) want = fill;
        }
        if (want > kMaxSpeculated) want = kMaxSpeculated;
        if (want < 1) want = 1;
        int n = 1;
        double a = alpha;
        int lit = ls_it + 1;                     // the first trial of the request is always allowed
        while (n < want && (o.wolfe_max_it <= 0 || lit < o.wolfe_max_it) && a * o.wolfe_decrease >= o.wolfe_min_step) {
            a = o.wolfe_decrease * a;
            ++lit;
            ++n;
        }
        return n;
    }

    // trials whose centres the next SCREEN request scores (>= 1): the pass costs little more for a
    // few dozen poses than for one
    int planned_screen() const
    {
        // the pass works through the poses in tiles of 16, with one of 8 at the end when that is all
        // that is left: a started tile costs the same full or not
        int want = (last_trials + 3 + 7) / 8 * 8;
        if (!first_request) want = 16;
        if (want > kMaxSpeculated) want = kMaxSpeculated;
        int n = 1;
        double a = alpha;
        int lit = ls_it + 1;
        while (n < want && (o.wolfe_max_it <= 0 || lit < o.wolfe_max_it) && a * o.wolfe_decrease >= o.wolfe_min_step) {
            a = o.wolfe_decrease * a;
            ++lit;
            ++n;
        }
        return n;
    }

    // the main stencil of the next iteration, should the trial with step length a be accepted
    void next_main_point(double a, double* xs) const
    {
        for (int k = 0; k < D; ++k) {
            const double st = o.momentum * step[k] + (1 - o.momentum) * a * grad[k];
            xs[k] = x[k] - st;
        }
    }

    int request_points() const
    {
        if (phase == MAIN) return stencil();
        if (phase == TRIAL_G) return 2 * D + (ride_along ? stencil() : 0);
        if (phase == TRIAL_FG) return fg_trials + 2 * D + (ride_along ? stencil() : 0);
        if (phase == SCREEN) return planned_screen();
        return planned_trials();
    }

    bool keep_going() const
    {
        return (o.max_iterations <= 0 || it < o.max_iterations) &&
               grad_len >= o.min_gradient_len && step_len >= o.min_step_len;
    }

    void start(int dim, const double* x0, const ccal_gd_opts& opts)
    {
        D = dim;
        o = opts;
        std::memset(x, 0, sizeof x);
        std::memset(grad, 0, sizeof grad);
        std::memset(step, 0, sizeof step);
        std::memcpy(x, x0, sizeof(double) * D);
        grad_len = o.min_gradient_len + 1;
        step_len = o.min_step_len + 1;
        it = 0;
        n_evals = 0;
        begin_iteration();
    }

    void begin_iteration()
    {
        if (!keep_going()) { phase = DONE; return; }
        for (int k = 0; k < D; ++k) x[k] -= step[k];
        phase = MAIN;
    }

    // the 2*D points c +- eps/2 along each axis (plus first)
    void emit_axis_points(const double* c, double* out) const
    {
        for (int i = 0; i < 2 * D; ++i) {
            double* p = out + (size_t)i * D;
            std::memcpy(p, c, sizeof(double) * D);
            if (i % 2 == 0) p[i / 2] += o.fd_eps / 2;
            else            p[i / 2] -= o.fd_eps / 2;
        }
    }

    // writes request_points() points of D doubles each
    void emit(double* out) const
    {
        if (phase == MAIN) {
            std::memcpy(out, x, sizeof(double) * D);
            emit_axis_points(x, out + D);
            return;
        }
        if (phase == TRIAL_G) {
            // a trial that passed Armijo has (so far always) passed the curvature test too, so the
            // next iteration's main stencil rides along with its gradient points
            double xs[10];
            emit_axis_points(xn, out);
            if (!ride_along) return;
            next_main_point(alpha, xs);
            std::memcpy(out + (size_t)2 * D * D, xs, sizeof(double) * D);
            emit_axis_points(xs, out + (size_t)(2 * D + 1) * D);
            return;
        }
        const int n = phase == SCREEN ? planned_screen() : (phase == TRIAL_FG ? fg_trials : planned_trials());
        double a = alpha, c[10];
        for (int t = 0; t < n; ++t) {
            if (t > 0) a = o.wolfe_decrease * a;   // same recurrence as next_trial()
            for (int k = 0; k < D; ++k) c[k] = x[k] - a * grad[k];
            std::memcpy(out + (size_t)t * D, c, sizeof(double) * D);
        }
        if (phase == TRIAL_FG) {
            // c and a are those of the last centre, the trial expected to pass
            double xs[10];
            double* rest = out + (size_t)n * D;
            emit_axis_points(c, rest);
            if (!ride_along) return;
            next_main_point(a, xs);
            std::memcpy(rest + (size_t)2 * D * D, xs, sizeof(double) * D);
            emit_axis_points(xs, rest + (size_t)(2 * D + 1) * D);
        }
    }

    // a.b the way Eigen sums it for the VectorXd operands gdcpp works on (VectorXd::dot, ::norm,
    // ::squaredNorm all end in Redux.h's linear vectorised traversal; x86-64 without -march: packets of
    // two doubles, two packet accumulators, the odd packet added last, then low + high lane).
    // For n = 6: ((v0 + v2) + v4) + ((v1 + v3) + v5), v_i = a_i b_i.
    static double dot(const double* a, const double* b, int n)
    {
        if (n < 2) return n == 1 ? a[0] * b[0] : 0.0;
        const int aligned2 = (n / 4) * 4, aligned = (n / 2) * 2;
        double lo0 = a[0] * b[0], hi0 = a[1] * b[1];
        if (aligned > 2) {
            double lo1 = a[2] * b[2], hi1 = a[3] * b[3];
            for (int i = 4; i < aligned2; i += 4) {
                lo0 = lo0 + a[i] * b[i];
                hi0 = hi0 + a[i + 1] * b[i + 1];
                lo1 = lo1 + a[i + 2] * b[i + 2];
                hi1 = hi1 + a[i + 3] * b[i + 3];
            }
            lo0 = lo0 + lo1;
            hi0 = hi0 + hi1;
            if (aligned > aligned2) {
                lo0 = lo0 + a[aligned2] * b[aligned2];
                hi0 = hi0 + a[aligned2 + 1] * b[aligned2 + 1];
            }
        }
        double r = lo0 + hi0;
        for (int i = aligned; i < n; ++i) r = r + a[i] * b[i];
        return r;
    }

    bool trial_allowed() const
    {
        return (o.wolfe_max_it <= 0 || ls_it < o.wolfe_max_it) &&
               alpha * o.wolfe_decrease >= o.wolfe_min_step;
    }

    void next_trial()
    {
        alpha = o.wolfe_decrease * alpha;
        for (int k = 0; k < D; ++k) xn[k] = x[k] - alpha * grad[k];
        phase = TRIAL_F;
    }

    void finish_iteration()
    {
        last_trials = ls_it;
        first_request = true;
        for (int k = 0; k < D; ++k)
            step[k] = o.momentum * step[k] + (1 - o.momentum) * alpha * grad[k];
        step_len = std::sqrt(dot(step, step, D));
        ++it;
        begin_iteration();
    }

    // f: values at emit_axis_points order
    void gradient_of(const double* f, double* g) const
    {
        for (int k = 0; k < D; ++k) g[k] = (f[2 * k] - f[2 * k + 1]) / o.fd_eps;
    }

    // The trial at xn passed Armijo and g is the gradient there: curvature test, then either the
    // iteration ends or the next trial is set up (phase TRIAL_F).  Returns true if the iteration ended.
    bool conclude_trial(const double* g)
    {
        const bool wolfe = assume_curvature_passes || dot(grad, g, D) <= o.wolfe_c2 * gg;
        ++ls_it;
        ++n_trials;
        n_evals += stencil();
        if (!wolfe) { ++n_wolfe_only_fail; ride_along = false; }
        if (!wolfe && trial_allowed()) { next_trial(); return false; }
        finish_iteration();
        return true;
    }

    // f: objective at x and at its 2*D axis points
    void feed_main(const double* f)
    {
        double g[10];
        n_evals += stencil();
        gradient_of(f + 1, g);
        fval = f[0];
        std::memcpy(grad, g, sizeof(double) * D);
        gg = dot(grad, grad, D);
        grad_len = std::sqrt(gg);
        alpha = o.wolfe_max_step / o.wolfe_decrease;
        ls_it = 0;
        first_request = true;
        ride_along = true;
        if (trial_allowed()) {
            next_trial();
            if (screening) phase = SCREEN;
        } else {
            finish_iteration();
        }
    }

    // The trial at xn has failed the Armijo test: count it and move to the next one.  Returns false
    // if the line search (and with it the iteration) ended instead.
    bool fail_trial()
    {
        ++ls_it;
        ++n_trials;
        ++n_armijo_fail;
        n_evals += stencil();
        if (!trial_allowed()) { finish_iteration(); return false; }
        next_trial();
        return true;
    }

    // fast[t], bound[t]: FP32 objective at the centre of the t-th next trial and a bound on its
    // distance from the exact value (request_points() of them, in emit() order).
    void feed_screen(const double* fast, const double* bound)
    {
        const int n = planned_screen();
        ++n_screen_requests;
        first_request = false;
        screen_first = -1;
        int t = 0;
        for (; t < n; ++t) {
            const double thr = fval - o.wolfe_c1 * alpha * gg;
            const bool sure_fail = std::isfinite(fast[t]) && std::isfinite(bound[t]) && fast[t] - bound[t] > thr;
            if (!sure_fail) break;
            ++n_screened_fail;
            if (!fail_trial()) return;          // the iteration ended: phase is MAIN or DONE
        }
        if (t == n) { phase = SCREEN; return; } // all of them failed for sure: screen the next ones
        // The trial at xn is number t of this request.  If it passes for sure, only its gradient is
        // wanted.  Otherwise the FP32 values (far closer to the exact ones than their bound) say
        // which trial to expect to pass: the undecided centres up to it are evaluated exactly
        // together with its gradient points.
        const double thr_t = fval - o.wolfe_c1 * alpha * gg;
        if (std::isfinite(fast[t]) && std::isfinite(bound[t]) && fast[t] + bound[t] <= thr_t) {
            ++n_screened_pass;
            phase = TRIAL_G;                    // xn passes Armijo whatever its exact value
            screen_first = t;
            return;
        }
        double a = alpha;
        for (int u = t; u < n; ++u) {
            if (std::isfinite(fast[u]) && fast[u] <= fval - o.wolfe_c1 * a * gg) {
                fg_trials = u - t + 1;
                phase = TRIAL_FG;
                screen_first = t;
                return;
            }
            a = o.wolfe_decrease * a;
        }
        phase = TRIAL_F;                        // no pass in sight: exact centres, as without screening
    }

    // The trial at xn passed Armijo; fg: objective at its 2*D axis points, then (if it rode along)
    // at the next main stencil.
    void feed_gradient(const double* fg)
    {
        double g[10], xs[10];
        next_main_point(alpha, xs);
        gradient_of(fg, g);
        first_request = false;
        const bool rode = ride_along;
        if (conclude_trial(g) && rode && phase == MAIN && std::memcmp(xs, x, sizeof(double) * D) == 0) {
            ++n_main_hits;
            feed_main(fg + 2 * D);   // the stencil that rode along is the one now due
        }
    }

    // f holds request_points() objective values in emit() order.  Every trial the sequential
    // algorithm evaluates (centre and gradient, 1 + 2*D values) is counted in n_evals whether or
    // not its gradient was needed here: a trial that fails Armijo is rejected whatever its gradient.
    void feed(const double* f)
    {
        n_requested += request_points();
        ++n_requests;
        screen_first = -1;
        if (phase == MAIN) {
            feed_main(f);
        } else if (phase == TRIAL_G) {
            feed_gradient(f);
        } else if (phase == TRIAL_FG) {
            const int n = fg_trials;
            first_request = false;
            for (int t = 0; t < n; ++t) {
                const bool armijo = f[t] <= fval - o.wolfe_c1 * alpha * gg;
                if (t + 1 < n) {
                    // an earlier trial than expected passes: its gradient is still to be evaluated
                    if (armijo) { ++n_expected_misses; phase = TRIAL_G; return; }
                    if (!fail_trial()) return;
                } else if (armijo) {
                    ++n_expected_hits;
                    feed_gradient(f + n);
                } else {
                    ++n_expected_misses;
                    if (fail_trial() && screening) phase = SCREEN;
                }
            }
        } else if (phase == TRIAL_F) {
            const int n = planned_trials();
            first_request = false;
            for (int t = 0; t < n; ++t) {
                const bool armijo = f[t] <= fval - o.wolfe_c1 * alpha * gg;
                if (armijo) { phase = TRIAL_G; return; }   // xn is this trial's point
                if (!fail_trial()) return;
            }
        }
    }

    // Everything the next request and its interpretation depend on.
    bool same_state(const Machine& m) const
    {
        return phase == m.phase && it == m.it && ls_it == m.ls_it && last_trials == m.last_trials &&
               first_request == m.first_request && ride_along == m.ride_along && fg_trials == m.fg_trials &&
               std::memcmp(x, m.x, sizeof x) == 0 && std::memcmp(grad, m.grad, sizeof grad) == 0 &&
               std::memcmp(step, m.step, sizeof step) == 0 && std::memcmp(xn, m.xn, sizeof xn) == 0 &&
               std::memcmp(&fval, &m.fval, sizeof fval) == 0 && std::memcmp(&gg, &m.gg, sizeof gg) == 0 &&
               std::memcmp(&alpha, &m.alpha, sizeof alpha) == 0 && std::memcmp(&grad_len, &m.grad_len, sizeof grad_len) == 0 &&
               std::memcmp(&step_len, &m.step_len, sizeof step_len) == 0;
    }

    void result(ccal_gd_result* r) const
    {
        r->iterations = it;
        r->converged = grad_len < o.min_gradient_len || step_len < o.min_step_len;
        r->fval = fval;
        std::memset(r->xval, 0, sizeof r->xval);
        std::memcpy(r->xval, x, sizeof(double) * D);
        r->n_evals = n_evals;
    }
};

} // namespace

extern "C" void ccal_gd_default_opts(ccal_gd_opts* o)
{
    if (!o) return;
    o->max_iterations = 10000;
    o->min_gradient_len = 1e-6;
    o->min_step_len = 1e-6;
    o->momentum = 0.4;
    o->fd_eps = std::sqrt(DBL_EPSILON);
    o->wolfe_decrease = 0.8;
    o->wolfe_c1 = 1e-4;
    o->wolfe_c2 = 0.9;
    o->wolfe_min_step = 1e-10;
    o->wolfe_max_step = 1.0;
    o->wolfe_max_it = 0;
}

extern "C" int ccal_gd_minimize_multi(ccal_ctx* ctx, int which, int dim, int n_starts,
                                      const double* x0, const ccal_gd_opts* opts,
                                      ccal_gd_result* results)
{
    if (!ctx) return CCAL_ERR_ARG;
    if ((dim != 6 && dim != 10) || n_starts <= 0 || !x0 || !results) {
        ctx->err = "gd_minimize: dim must be 6 or 10 and arguments non-null";
        return CCAL_ERR_ARG;
    }
    if (which != CCAL_DOWNSAMPLED && which != CCAL_RAW) {
        ctx->err = "cloud-set selector must be CCAL_DOWNSAMPLED or CCAL_RAW";
        return CCAL_ERR_ARG;
    }
    if (!ctx->sets[which].ready) { ctx->err = "cloud set not uploaded"; return CCAL_ERR_STATE; }
    if (dim == 6 && !ctx->plane_set) { ctx->err = "gd_minimize: plane not set"; return CCAL_ERR_STATE; }
    ccal_gd_opts o;
    if (opts) o = *opts;
    else ccal_gd_default_opts(&o);
    if (!(o.fd_eps > 0) || !(o.wolfe_decrease > 0 && o.wolfe_decrease < 1)) {
        ctx->err = "gd_minimize: fd_eps must be > 0 and wolfe_decrease in (0,1)";
        return CCAL_ERR_ARG;
    }

    std::vector<Machine> ms(n_starts);
    for (int s = 0; s < n_starts; ++s) {
        ms[s].screening = ctx->gd_screening;
        ms[s].start(dim, x0 + (size_t)s * dim, o);
    }
    if (n_starts == 1) {
        ms[0].chains_per_pose = ctx->sets[which].n_clouds;
        ms[0].round_chains = ctx->n_cu;
    }

    // One round: the descents that want exact values share one call, those that want the screening
    // pass share another; every descent is fed what it asked for and moves on.
    std::vector<double> pts, poses, planes, f, bound;
    std::vector<int> active;
    std::vector<size_t> offs;
    auto gather = [&](bool screen) {
        active.clear();
        offs.clear();
        size_t n = 0;
        for (int s = 0; s < n_starts; ++s)
            if (ms[s].phase != Machine::DONE && (ms[s].phase == Machine::SCREEN) == screen) {
                active.push_back(s);
                offs.push_back(n);
                n += (size_t)ms[s].request_points();
            }
        if (active.empty()) return (size_t)0;
        pts.resize(n * dim);
        for (size_t a = 0; a < active.size(); ++a) ms[active[a]].emit(pts.data() + offs[a] * dim);
        poses.resize(n * 6);
        f.resize(n);
        if (dim == 10) {
            planes.resize(n * 4);
            for (size_t i = 0; i < n; ++i) {
                std::memcpy(&poses[i * 6], &pts[i * 10], 6 * sizeof(double));
                std::memcpy(&planes[i * 4], &pts[i * 10 + 6], 4 * sizeof(double));
            }
        } else {
            std::memcpy(poses.data(), pts.data(), n * 6 * sizeof(double));
        }
        return n;
    };
    double t_screen = 0.0, t_exact = 0.0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    int64_t n_lookahead = 0, n_lookahead_kept = 0;
    // points (dim doubles each) -> the pose / plane arrays the objective takes
    auto to_poses = [&](const std::vector<double>& p, size_t n, std::vector<double>& po, std::vector<double>& pl) {
        po.resize(n * 6);
        if (dim == 10) {
            pl.resize(n * 4);
            for (size_t i = 0; i < n; ++i) {
                std::memcpy(&po[i * 6], &p[i * 10], 6 * sizeof(double));
                std::memcpy(&pl[i * 4], &p[i * 10 + 6], 4 * sizeof(double));
            }
        } else {
            std::memcpy(po.data(), p.data(), n * 6 * sizeof(double));
        }
    };
    std::vector<double> pts2, poses2, planes2, fs, bs, f_order, screen_sums, expected_sums, next_sums;
    int64_t n_stencil_screens = 0;
    const int n_clouds = ctx->sets[which].n_clouds;
    for (;;) {
        // A lone descent whose next request is "the undecided centres, the expected trial's gradient points
        // and the next main stencil" sends it in two parts: centres and next stencil first, gradient points
        // second.  With the first part's values a copy of the machine looks ahead (curvature test taken to
        // pass) and the next iteration's screening request goes onto the device while the second part is
        // still being computed; it runs beside that part's finishing pass.  The machine itself then gets
        // the whole request as before and the look-ahead counts only if it arrives where the copy is.
        Machine& m0 = ms[0];
        if (n_starts == 1 && ctx->gd_pipelining && m0.screening && m0.phase == Machine::TRIAL_FG && m0.ride_along) {
            const int D = m0.D, nc = m0.fg_trials, n = m0.request_points();     // nc centres, 2 D gradient points, 1 + 2 D stencil
            pts.resize((size_t)n * dim);
            m0.emit(pts.data());
            // request order: centres, next main stencil, gradient points
            pts2.resize(pts.size());
            std::memcpy(pts2.data(), pts.data(), (size_t)nc * dim * sizeof(double));
            std::memcpy(pts2.data() + (size_t)nc * dim, pts.data() + (size_t)(nc + 2 * D) * dim, (size_t)(1 + 2 * D) * dim * sizeof(double));
            std::memcpy(pts2.data() + (size_t)(nc + 1 + 2 * D) * dim, pts.data() + (size_t)nc * dim, (size_t)2 * D * dim * sizeof(double));
            to_poses(pts2, (size_t)n, poses, planes);
            const int n_first = nc + 1 + 2 * D;
            f.resize(n);
            f_order.assign(n, 0.0);
            double t0 = now();
            const double* expected = nullptr;
            if (m0.screen_first >= 0 && (size_t)(m0.screen_first + nc) * n_clouds <= screen_sums.size()) {
                expected_sums.resize((size_t)n * n_clouds);
                for (int i = 0; i < n; ++i) {
                    const int from = m0.screen_first + (i < nc ? i : nc - 1);
                    std::memcpy(&expected_sums[(size_t)i * n_clouds], &screen_sums[(size_t)from * n_clouds], n_clouds * sizeof(double));
                }
                expected = expected_sums.data();
            }
            if (int rc = ccal::exact_begin(ctx, which, n, poses.data(), dim == 10 ? planes.data() : nullptr, n_first, false, true, expected))
                return rc;
            bool two_parts = ctx->exact_job.n_first == n_first, looked_ahead = false;
            Machine ahead = m0;
            if (two_parts) {
                if (int rc = ccal::exact_wait_first(ctx, f.data())) return rc;
                t_exact += now() - t0;
                // the machine's order: centres, gradient points (unknown yet: zeros), stencil
                std::memcpy(f_order.data(), f.data(), (size_t)nc * sizeof(double));
                std::memcpy(f_order.data() + nc + 2 * D, f.data() + nc, (size_t)(1 + 2 * D) * sizeof(double));
                ahead.assume_curvature_passes = true;
                ahead.feed(f_order.data());
                ahead.assume_curvature_passes = false;
                if (ahead.phase == Machine::SCREEN && ahead.it == m0.it + 1) {
                    const int ns = ahead.request_points();
                    pts.resize((size_t)ns * dim);
                    ahead.emit(pts.data());
                    to_poses(pts, (size_t)ns, poses2, planes2);
                    t0 = now();
                    if (int rc = ccal::screen_begin(ctx, which, ns, poses2.data(), dim == 10 ? planes2.data() : nullptr)) return rc;
                    t_screen += now() - t0;
                    looked_ahead = true;
                    ++n_lookahead;
                }
                t0 = now();
            }
            if (int rc = ccal::exact_wait_rest(ctx, f.data() + (two_parts ? n_first : 0), nullptr, nullptr)) return rc;
            t_exact += now() - t0;
            std::memcpy(f_order.data(), f.data(), (size_t)nc * sizeof(double));
            std::memcpy(f_order.data() + nc, f.data() + n_first, (size_t)2 * D * sizeof(double));
            std::memcpy(f_order.data() + nc + 2 * D, f.data() + nc, (size_t)(1 + 2 * D) * sizeof(double));
            m0.feed(f_order.data());
            if (looked_ahead) {
                const int ns = ahead.request_points();
                fs.resize(ns);
                bs.resize(ns);
                t0 = now();
                screen_sums.resize((size_t)ns * n_clouds);
                if (int rc = ccal::screen_wait(ctx, fs.data(), bs.data(), screen_sums.data())) return rc;
                t_screen += now() - t0;
                if (m0.same_state(ahead)) {
                    ++n_lookahead_kept;
                    m0.feed_screen(fs.data(), bs.data());
                }
            }
            continue;
        }
        bool any = false;
        if (const size_t n = gather(true)) {
            any = true;
            bound.resize(n);
            const double t0 = now();
            // a lone descent keeps the pass's per-cloud sums: the exact request that settles the line search
            // evaluates poses at or next to these centres, and says so (see expected_sums of exact_begin)
            if (n_starts == 1) screen_sums.resize(n * (size_t)n_clouds);
            int rc = ccal::eval_screen(ctx, which, (int)n, poses.data(), dim == 10 ? planes.data() : nullptr, f.data(),
                                       bound.data(), n_starts == 1 ? screen_sums.data() : nullptr);
            t_screen += now() - t0;
            if (rc != CCAL_OK) return rc;
            for (size_t a = 0; a < active.size(); ++a) ms[active[a]].feed_screen(f.data() + offs[a], bound.data() + offs[a]);
            // a lone descent's settling request goes out in two parts (top of the loop)
            if (n_starts == 1 && ctx->gd_pipelining && ms[0].phase == Machine::TRIAL_FG && ms[0].ride_along) continue;
        }
        if (const size_t n = gather(false)) {
            any = true;
            const double* expected = nullptr;
            const Machine& m = ms[0];
            if (n_starts == 1 && m.screening && m.phase == Machine::MAIN && n == (size_t)m.stencil()) {
                // a main stencil that did not ride along with a gradient (the first of a descent, as a rule): the
                // running sums on the device belong to some other pose, so the centre goes through the screening
                // pass first and its sums stand in for the whole stencil
                double f1, b1;
                next_sums.resize(n_clouds);
                if (int rc = ccal::eval_screen(ctx, which, 1, poses.data(), dim == 10 ? planes.data() : nullptr, &f1, &b1, next_sums.data()))
                    return rc;
                ++n_stencil_screens;
                expected_sums.resize(n * (size_t)n_clouds);
                for (size_t i = 0; i < n; ++i) std::memcpy(&expected_sums[i * n_clouds], next_sums.data(), n_clouds * sizeof(double));
                expected = expected_sums.data();
            }
            if (n_starts == 1 && m.screen_first >= 0 && (m.phase == Machine::TRIAL_FG || m.phase == Machine::TRIAL_G)) {
                // centres: the screening pass's sums of the same pose; the gradient points and the next main
                // stencil: those of the trial they surround or follow
                const int nc = m.phase == Machine::TRIAL_FG ? m.fg_trials : 0;
                const int last = m.screen_first + (nc > 0 ? nc - 1 : 0);
                if ((size_t)(last + 1) * n_clouds <= screen_sums.size()) {
                    expected_sums.resize(n * (size_t)n_clouds);
                    for (size_t i = 0; i < n; ++i) {
                        const int from = (int)i < nc ? m.screen_first + (int)i : last;
                        std::memcpy(&expected_sums[i * n_clouds], &screen_sums[(size_t)from * n_clouds], n_clouds * sizeof(double));
                    }
                    expected = expected_sums.data();
                    // The next main stencil lies a momentum term away from the trial.  While the descent still moves
                    // far in one iteration (the totals have changed by more than 3 % since the last request) that is
                    // too far for the trial's sums to stand in: one more pose through the screening pass (about 40 us)
                    // saves the exact request several hundred.
                    const std::vector<double>& ref = ctx->sets[which].seg_ref_totals;
                    const int D = m.D;
                    if (m.ride_along && ref.size() == (size_t)n_clouds && n >= (size_t)(1 + 2 * D)) {
                        double moved = 0.0;
                        for (int j = 0; j < n_clouds; ++j)
                            if (ref[j] > 0.0) moved = std::max(moved, std::fabs(screen_sums[(size_t)last * n_clouds + j] / ref[j] - 1.0));
                        if (moved > 0.03) {
                            const double* centre = pts.data() + (n - (size_t)(1 + 2 * D)) * dim;    // the stencil's first point
                            double pose1[6], plane1[4], f1, b1;
                            std::memcpy(pose1, centre, sizeof pose1);
                            if (dim == 10) std::memcpy(plane1, centre + 6, sizeof plane1);
                            next_sums.resize(n_clouds);
                            if (int rc = ccal::eval_screen(ctx, which, 1, pose1, dim == 10 ? plane1 : nullptr, &f1, &b1, next_sums.data()))
                                return rc;
                            ++n_stencil_screens;
                            for (size_t i = n - (size_t)(1 + 2 * D); i < n; ++i)
                                std::memcpy(&expected_sums[i * n_clouds], next_sums.data(), n_clouds * sizeof(double));
                        }
                    }
                }
            }
            const double t0 = now();
            int rc = ccal::eval_exact(ctx, which, (int)n, poses.data(), dim == 10 ? planes.data() : nullptr, f.data(),
                                      nullptr, nullptr, true, expected);
            t_exact += now() - t0;
            if (rc != CCAL_OK) return rc;
            for (size_t a = 0; a < active.size(); ++a) ms[active[a]].feed(f.data() + offs[a]);
        }
        if (!any) break;
    }
    if (std::getenv("CCAL_GD_STATS"))
        std::fprintf(stderr, "gd: %.3f s in screening calls, %.3f s in exact calls; split evaluations summed %lld of %lld "
                     "segments twice (context totals)\n", t_screen, t_exact, (long long)ctx->split_resummed,
                     (long long)ctx->split_segments);
    if (std::getenv("CCAL_GD_STATS"))
        std::fprintf(stderr, "gd: %lld screening requests issued ahead of the curvature test, %lld of them kept\n",
                     (long long)n_lookahead, (long long)n_lookahead_kept);
    if (std::getenv("CCAL_GD_STATS"))
        std::fprintf(stderr, "gd: %lld main points went through the screening pass ahead of their exact request\n",
                     (long long)n_stencil_screens);
    if (std::getenv("CCAL_GD_STATS"))
        for (int s = 0; s < n_starts; ++s)
            std::fprintf(stderr, "gd[%d]: %lld evaluations of the sequential algorithm, %lld evaluated in %lld requests; %lld trials, "
                         "%lld failed Armijo, %lld passed Armijo but failed the curvature test; %lld main "
                         "stencils rode along with a gradient; %lld screening requests settled %lld failing and "
                         "%lld passing trials; the expected trial passed %lld times, another %lld times\n", s,
                         (long long)ms[s].n_evals, (long long)ms[s].n_requested, (long long)ms[s].n_requests,
                         (long long)ms[s].n_trials, (long long)ms[s].n_armijo_fail, (long long)ms[s].n_wolfe_only_fail,
                         (long long)ms[s].n_main_hits, (long long)ms[s].n_screen_requests,
                         (long long)ms[s].n_screened_fail, (long long)ms[s].n_screened_pass,
                         (long long)ms[s].n_expected_hits, (long long)ms[s].n_expected_misses);
    for (int s = 0; s < n_starts; ++s) ms[s].result(&results[s]);
    return CCAL_OK;
}

extern "C" int ccal_gd_minimize(ccal_ctx* ctx, int which, int dim, const double* x0,
                                const ccal_gd_opts* opts, ccal_gd_result* result)
{
    return ccal_gd_minimize_multi(ctx, which, dim, 1, x0, opts, result);
}
