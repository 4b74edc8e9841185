// api_search.hip -- the fast FP32 objective and the discrete candidate-grid search: dense sweep on the
// matrix cores or the separable histogram pass, error budget, exact re-score of whatever the budget
// cannot exclude.
#include "ccal_internal.hpp"
#include "kernels_residual.hpp"
#include "kernels_separable.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <chrono>
#include <limits>
#include <new>
#include <thread>

using namespace ccal;

namespace {

FoldConsts fold_consts(const float plane[4])
{
    FoldConsts fc;
    fc.n[0] = plane[0]; fc.n[1] = plane[1]; fc.n[2] = plane[2];
    fc.d = plane[3];
    fc.inv_norm = 1.0 / std::sqrt(fc.n[0] * fc.n[0] + fc.n[1] * fc.n[1] + fc.n[2] * fc.n[2]);
    return fc;
}

} // namespace

// The points as ready-made bf16x3 operands of the wide residual kernel (64 B per point
// instead of 16): built on first use of that kernel only.
int ccal::ensure_split_groups(ccal_ctx* ctx, CloudSet& cs)
{
    if (cs.split_ready) return CCAL_OK;
    CCAL_HIP(ctx, cs.split_groups.ensure((size_t)cs.n_groups * 256 * sizeof(u32x4)));
    hipLaunchKernelGGL(split_groups_kernel, dim3((unsigned)cs.n_groups), dim3(256), 0, ctx->stream,
                       cs.groups.as<f32x4>(), cs.n_groups, cs.split_groups.as<u32x4>());
    CCAL_HIP(ctx, hipGetLastError());
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cs.split_ready = true;
    return CCAL_OK;
}

namespace {

// wgen + residual + split reduction for one chunk of candidates; costs go to
// cost_dst[0..n).  ev0/ev1 (optional) bracket the residual kernel.
template <typename Source>
int run_fast_chunk(ccal_ctx* ctx, const CloudSet& cs, const Source& src, int64_t n,
                   float* cost_dst, hipEvent_t ev0, hipEvent_t ev1)
{
    // the default kernel gives a wave 16 candidate tiles, the others 8
    FastLaunch f = plan_fast(ctx, cs, n, ctx->fast_kernel == CCAL_FAST_BF16X3 ? 2 * kCandPerBlock : kCandPerBlock);
    CCAL_HIP(ctx, ctx->w_buf.ensure((size_t)cs.n_clouds * f.n_ctiles * 64 * sizeof(float)));
    CCAL_HIP(ctx, ctx->partial_buf.ensure((size_t)f.n_splits * f.n_padded * sizeof(double)));
    const FoldConsts fc = fold_consts(ctx->plane);
    hipLaunchKernelGGL(wgen_kernel<Source>, dim3((unsigned)((f.n_padded + 255) / 256)), dim3(256),
                       0, ctx->stream, src, n, f.n_ctiles, cs.n_clouds,
                       cs.base_dev.as<double>(), cs.centres_dev.as<double>(), fc,
                       ctx->w_buf.as<float>());
    if (ev0) CCAL_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    if (ctx->fast_kernel == CCAL_FAST_BF16X3_WIDE)
        hipLaunchKernelGGL(residual_bf16x3_wide_kernel<kNA / 2>, dim3((unsigned)f.cand_blocks, f.n_splits),
                           dim3(256), 0, ctx->stream, ctx->w_buf.as<float>(),
                           cs.split_groups.as<u32x4>(), cs.group_off_dev.as<int>(),
                           cs.scale_dev.as<double>(), ctx->partial_buf.as<double>(), cs.n_clouds,
                           f.n_ctiles, f.n_splits);
    else if (ctx->fast_kernel == CCAL_FAST_F32)
        hipLaunchKernelGGL(residual_mfma_kernel<kNA>, dim3((unsigned)f.cand_blocks, f.n_splits),
                           dim3(256), 0, ctx->stream, ctx->w_buf.as<float>(),
                           cs.groups.as<f32x4>(), cs.group_off_dev.as<int>(),
                           cs.scale_dev.as<double>(), ctx->partial_buf.as<double>(), cs.n_clouds,
                           f.n_ctiles, f.n_splits);
    else if (ctx->fast_kernel == CCAL_FAST_BF16X3_8)
        hipLaunchKernelGGL((residual_bf16x3_kernel<kNA, 4>), dim3((unsigned)f.cand_blocks, f.n_splits),
                           dim3(256), 0, ctx->stream, ctx->w_buf.as<float>(),
                           cs.groups.as<f32x4>(), cs.group_off_dev.as<int>(),
                           cs.scale_dev.as<double>(), ctx->partial_buf.as<double>(), cs.n_clouds,
                           f.n_ctiles, f.n_splits);
    else
        hipLaunchKernelGGL((residual_bf16x3_kernel<2 * kNA, 8>), dim3((unsigned)f.cand_blocks, f.n_splits),
                           dim3(256), 0, ctx->stream, ctx->w_buf.as<float>(),
                           cs.groups.as<f32x4>(), cs.group_off_dev.as<int>(),
                           cs.scale_dev.as<double>(), ctx->partial_buf.as<double>(), cs.n_clouds,
                           f.n_ctiles, f.n_splits);
    if (ev1) CCAL_HIP(ctx, hipEventRecord(ev1, ctx->stream));
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, ctx->partial_buf.as<double>(), f.n_padded, f.n_splits, n,
                       cost_dst);
    CCAL_HIP(ctx, hipGetLastError());
    return CCAL_OK;
}

} // namespace

// ---------------------------------------------------------------------------
// fast objective for explicit pose lists
// ---------------------------------------------------------------------------

extern "C" int ccal_eval_fast(ccal_ctx* ctx, int which, int64_t n_poses, const double* poses6,
                              float* cost)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (n_poses < 0 || (n_poses > 0 && (!poses6 || !cost)))
        return fail(ctx, CCAL_ERR_ARG, "eval_fast: null argument");
    if (!ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "eval_fast: plane not set");
    if (n_poses == 0) return CCAL_OK;
    if (int rc = bind(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    const int64_t chunk = fast_chunk(cs);
    std::vector<double> rec;
    for (int64_t c0 = 0; c0 < n_poses; c0 += chunk) {
        const int64_t n = std::min(chunk, n_poses - c0);
        rec.resize((size_t)n * 9);
        for (int64_t c = 0; c < n; ++c) {
            const double* pose = poses6 + 6 * (size_t)(c0 + c);
            SinCos3 sc = sincos_of_pose(pose);
            double* r = rec.data() + 9 * (size_t)c;
            for (int k = 0; k < 3; ++k) { r[k] = pose[k]; r[3 + k] = sc.s[k]; r[6 + k] = sc.c[k]; }
        }
        CCAL_HIP(ctx, ctx->rec_buf.ensure(rec.size() * sizeof(double)));
        CCAL_HIP(ctx, ctx->cost_buf.ensure((size_t)n * sizeof(float)));
        CCAL_HIP(ctx, hipMemcpyAsync(ctx->rec_buf.p, rec.data(), rec.size() * sizeof(double),
                                     hipMemcpyHostToDevice, ctx->stream));
        PoseList src{ctx->rec_buf.as<double>()};
        if (int rc = run_fast_chunk(ctx, cs, src, n, ctx->cost_buf.as<float>(), nullptr, nullptr))
            return rc;
        CCAL_HIP(ctx, hipMemcpyAsync(cost + c0, ctx->cost_buf.p, (size_t)n * sizeof(float),
                                     hipMemcpyDeviceToHost, ctx->stream));
        CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CCAL_OK;
}

// ---------------------------------------------------------------------------
// grid
// ---------------------------------------------------------------------------

extern "C" void ccal_grid_default(ccal_grid* g)
{
    if (!g) return;
    for (int k = 0; k < 3; ++k) { g->min[k] = -20; g->step[k] = 4; g->count[k] = 11; }
    for (int k = 3; k < 6; ++k) { g->min[k] = -5; g->step[k] = 1; g->count[k] = 11; }
}

extern "C" int64_t ccal_grid_size(const ccal_grid* g)
{
    if (!g) return 0;
    int64_t n = 1;
    for (int k = 0; k < 6; ++k) {
        if (g->count[k] <= 0) return 0;
        n *= g->count[k];
    }
    return n;
}

static inline double grid_axis_value(const ccal_grid* g, const double base[6], int k, int i)
{
    const double v = g->min[k] + i * g->step[k];
    if (k < 3) return base[k] + v / 1000.0;  // :540 transformation[k] + x/1000.0
    return base[k] + v * 3.141592 / 180;     // helpers.hpp:275-278
}

extern "C" void ccal_grid_pose(const ccal_grid* g, const double base_pose[6], int64_t index,
                               double pose[6])
{
    if (!g || !base_pose || !pose || ccal_grid_size(g) <= 0 || index < 0) return;
    int idx[6];
    for (int k = 5; k >= 0; --k) {
        idx[k] = (int)(index % g->count[k]);
        index /= g->count[k];
    }
    for (int k = 0; k < 6; ++k) pose[k] = grid_axis_value(g, base_pose, k, idx[k]);
}

extern "C" int ccal_argmin_merge(int n, const double* costs, const int64_t* indices)
{
    int best = -1;
    if (!costs || !indices) return best;
    for (int i = 0; i < n; ++i) {
        if (indices[i] < 0) continue;
        if (best < 0 || costs[i] < costs[best] ||
            (costs[i] == costs[best] && indices[i] < indices[best]))
            best = i;
    }
    return best;
}

// ---------------------------------------------------------------------------
// separable search: all translations of a rotation from one pass over the points
// ---------------------------------------------------------------------------
// Translations per launch (histogram bins in LDS, 8 bytes each) x rotations per workgroup.  Every launch
// evaluates every (rotation, point) pair anew, so the work of a search is pairs x launches and the widest
// form that fits is the fastest for a long list even though it leaves one workgroup per CU.
struct SepForm { int tc, rb, bpb; };
static SepForm separable_form(int64_t n_translations)
{
    if (n_translations <= 1024) return {1024, 4, 8};    //  53 KB of LDS: three workgroups of 512 threads per CU
    if (n_translations <= 2048) return {2048, 2, 4};    //  57 KB: two
    return {5120, 2, 4};                                // 143 KB: one workgroup of 1024 threads
}

// Scale of the separable pass's fixed point for residuals bounded by vmax: the largest power of two with
// vmax * scale < 2^31 (capped at 2^40; 1 for a non-finite bound, which the pass is never run with).
static double separable_fx_scale(double vmax)
{
    if (!std::isfinite(vmax)) return 1.0;
    int e = 40;
    while (e > -60 && vmax * std::ldexp(1.0, e) >= 2147483000.0) --e;
    return std::ldexp(1.0, e);
}

template <int TC, int RB, int BPB, int NT>
static hipError_t launch_separable(ccal_ctx* ctx, const CloudSet& cs, const f32x4* wr, int nt, int nr, float fx_scale)
{
    // wide accumulators: one slice of RB x (TC + 1) bins per workgroup
    const size_t n_wg = (size_t)((nr + RB - 1) / RB) * cs.n_clouds, bins = (size_t)RB * (TC + 1);
    hipError_t e = ctx->sep_wcnt.ensure(n_wg * bins * sizeof(unsigned int));
    if (e == hipSuccess) e = ctx->sep_wsum.ensure(n_wg * bins * sizeof(long long));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((separable_hist_kernel<TC, RB, BPB, NT>), dim3((nr + RB - 1) / RB, cs.n_clouds),
                       dim3(NT), 0, ctx->stream, cs.groups.as<f32x4>(), cs.group_off_dev.as<int>(),
                       wr, ctx->sep_sb.as<float>(), ctx->sep_perm.as<int>(), ctx->sep_tab.as<unsigned short>(),
                       ctx->sep_meta.as<SepMeta>(), cs.scale_dev.as<double>(), cs.cloud_pt_off_dev.as<int64_t>(),
                       cs.n_clouds, nt, nr, fx_scale, ctx->sep_wcnt.as<unsigned int>(), ctx->sep_wsum.as<long long>(), ctx->partial_buf.as<double>());
    return hipSuccess;
}

// Fast-path costs of translations [t0, t1) x all R rotations of the grid into
// cost[(t - t0) * R + r] (device).  kernel_ms accumulates the histogram kernel's time.
static int run_separable(ccal_ctx* ctx, const CloudSet& cs, const double base[6], const ccal_grid* g,
                         int64_t t0, int64_t t1, float* cost, double* kernel_ms, int* launches)
{
    const int C = cs.n_clouds;
    const int n3 = g->count[3], n4 = g->count[4], n5 = g->count[5];
    const int64_t R = (int64_t)n3 * n4 * n5;
    const FoldConsts fc = fold_consts(ctx->plane);

    // per cloud: g_j = R_j^T n and h_j = n.b_j + d (the pose-independent parts of the fold)
    std::vector<double> gj((size_t)3 * C), hj(C);
    for (int j = 0; j < C; ++j) {
        const double* B = cs.base_T_flange.data() + 16 * (size_t)j;
        for (int k = 0; k < 3; ++k) gj[3 * j + k] = fc.n[0] * B[k] + fc.n[1] * B[4 + k] + fc.n[2] * B[8 + k];
        hj[j] = fc.n[0] * B[3] + fc.n[1] * B[7] + fc.n[2] * B[11] + fc.d;
    }
    auto translation = [&](int64_t t, double tf[3]) {
        const int i2 = (int)(t % g->count[2]); t /= g->count[2];
        const int i1 = (int)(t % g->count[1]); t /= g->count[1];
        tf[0] = grid_axis_value(g, base, 0, (int)t);
        tf[1] = grid_axis_value(g, base, 1, i1);
        tf[2] = grid_axis_value(g, base, 2, i2);
    };
    auto b_of = [&](int j, const double tf[3]) {
        return (gj[3 * j] * tf[0] + gj[3 * j + 1] * tf[1] + gj[3 * j + 2] * tf[2] + hj[j]) * fc.inv_norm;
    };
    // reference constant per cloud: the middle translation of the whole grid (not of [t0, t1): the costs
    // of a translation are then the same bits whichever range of the grid a call covers)
    std::vector<double> bref(C);
    {
        double tf[3];
        translation((int64_t)g->count[0] * g->count[1] * g->count[2] / 2, tf);
        for (int j = 0; j < C; ++j) bref[j] = b_of(j, tf);
    }

    // rotation vectors: w123 = R_F^T g_j / |n|, a = w123.c_j + bref_j
    std::vector<float> wr((size_t)R * C * 4);
    double vmax = 0.0;   // bound on |u| = |w123 . (p - c) + a| over rotations, clouds and points
    {
        std::vector<double> sn[3], cn[3];
        for (int k = 0; k < 3; ++k) {
            sn[k].resize(g->count[3 + k]);
            cn[k].resize(g->count[3 + k]);
            for (int i = 0; i < g->count[3 + k]; ++i) {
                const double a = grid_axis_value(g, base, 3 + k, i);
                sin_cos(a, &sn[k][i], &cn[k][i]);
            }
        }
        for (int64_t r = 0; r < R; ++r) {
            const int i5 = (int)(r % n5), i4 = (int)((r / n5) % n4), i3 = (int)(r / ((int64_t)n5 * n4));
            const double s[3] = {sn[0][i3], sn[1][i4], sn[2][i5]}, c[3] = {cn[0][i3], cn[1][i4], cn[2][i5]};
            double RF[9];
            rotation_zyx(s, c, RF);
            for (int j = 0; j < C; ++j) {
                double w[3];
                for (int k = 0; k < 3; ++k)
                    w[k] = (gj[3 * j] * RF[k] + gj[3 * j + 1] * RF[3 + k] + gj[3 * j + 2] * RF[6 + k]) * fc.inv_norm;
                const double a = w[0] * cs.centres[3 * j] + w[1] * cs.centres[3 * j + 1] +
                                 w[2] * cs.centres[3 * j + 2] + bref[j];
                float* dst = wr.data() + ((size_t)r * C + j) * 4;
                dst[0] = (float)w[0]; dst[1] = (float)w[1]; dst[2] = (float)w[2]; dst[3] = (float)a;
                const double wn = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                vmax = std::max(vmax, (wn * cs.radius[j] + std::fabs(a)) * (1.0 + 1e-5));
            }
        }
    }
    CCAL_HIP(ctx, ctx->sep_wr.ensure(wr.size() * sizeof(float)));
    CCAL_HIP(ctx, hipMemcpyAsync(ctx->sep_wr.p, wr.data(), wr.size() * sizeof(float), hipMemcpyHostToDevice,
                                 ctx->stream));

    // fixed point of the histogram sums: the power of two that keeps |u| * fx_scale below 2^31
    const float fx_scale = (float)separable_fx_scale(vmax);
    const SepForm form = separable_form(t1 - t0);
    const int tc = form.tc, rb = form.rb, bpb = form.bpb;
    const int M = bpb * tc;
    std::vector<float> sb;
    std::vector<int> perm;
    std::vector<unsigned short> tab;
    std::vector<SepMeta> meta(C);
    std::vector<std::pair<float, int>> order;
    Events ev;
    CCAL_HIP(ctx, ev.create(2));
    const hipEvent_t e0 = ev[0], e1 = ev[1];
    for (int64_t ta = t0; ta < t1; ta += tc) {
        const int nt = (int)std::min<int64_t>(tc, t1 - ta);
        sb.assign((size_t)C * nt, 0.f);
        perm.assign((size_t)C * nt, 0);
        tab.assign((size_t)C * (M + 1), 0);
        for (int j = 0; j < C; ++j) {
            order.resize(nt);
            for (int tl = 0; tl < nt; ++tl) {
                double tf[3];
                translation(ta + tl, tf);
                order[tl] = {(float)(b_of(j, tf) - bref[j]), tl};
            }
            std::sort(order.begin(), order.end());
            for (int q = 0; q < nt; ++q) {
                sb[(size_t)j * nt + q] = order[q].first;
                perm[(size_t)j * nt + q] = order[q].second;
            }
            const float bmin = order.front().first, bmax = order.back().first;
            const double range = (double)bmax - (double)bmin;
            const double delta = range > 1e-30 ? range / M : 1.0;
            meta[j].inv_delta = (float)(1.0 / delta);
            meta[j].offset = (float)(-(double)bmin / delta);
            // entry m: four times the number of thresholds in buckets below m (the rank a key of bucket m
            // starts from), flagged when bucket m holds two or more itself; [M] = 4 nt
            unsigned short* tj = tab.data() + (size_t)j * (M + 1);
            int q = 0;
            for (int m = 0; m < M; ++m) {
                const int lo = q;
                while (q < nt && separable_bucket(order[q].first, meta[j], M) <= m) ++q;
                tj[m] = (unsigned short)(4u * (unsigned)lo | (q - lo >= 2 ? kSepCrowded : 0u));
            }
            tj[M] = (unsigned short)(4 * nt);
        }
        // rotations per launch: bounded by the scratch a rotation takes, per cloud 8 B per translation of
        // the partial buffer and 12 B per bin (tc + 1 of them, however few translations the launch holds)
        // of the wide accumulators
        const int64_t budget = (int64_t)1 << 30;
        int64_t nr_max = budget / ((int64_t)C * (8 * (int64_t)nt + 12 * ((int64_t)tc + 1)));
        nr_max = std::max<int64_t>(rb, (nr_max / rb) * rb);
        auto up = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
            hipError_t e = b.ensure(bytes);
            if (e != hipSuccess) return e;
            return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        };
        CCAL_HIP(ctx, up(ctx->sep_sb, sb.data(), sb.size() * sizeof(float)));
        CCAL_HIP(ctx, up(ctx->sep_perm, perm.data(), perm.size() * sizeof(int)));
        CCAL_HIP(ctx, up(ctx->sep_tab, tab.data(), tab.size() * sizeof(unsigned short)));
        CCAL_HIP(ctx, up(ctx->sep_meta, meta.data(), meta.size() * sizeof(SepMeta)));
        for (int64_t ra = 0; ra < R; ra += nr_max) {
            const int nr = (int)std::min<int64_t>(nr_max, R - ra);
            CCAL_HIP(ctx, ctx->partial_buf.ensure((size_t)C * nt * nr * sizeof(double)));
            CCAL_HIP(ctx, hipEventRecord(e0, ctx->stream));
            const f32x4* wr_dev = ctx->sep_wr.as<f32x4>() + (size_t)ra * C;
            if (tc == 1024) CCAL_HIP(ctx, (launch_separable<1024, 4, 8, 512>(ctx, cs, wr_dev, nt, nr, fx_scale)));
            else if (tc == 2048) CCAL_HIP(ctx, (launch_separable<2048, 2, 4, 512>(ctx, cs, wr_dev, nt, nr, fx_scale)));
            else CCAL_HIP(ctx, (launch_separable<5120, 2, 4, 1024>(ctx, cs, wr_dev, nt, nr, fx_scale)));
            CCAL_HIP(ctx, hipEventRecord(e1, ctx->stream));
            hipLaunchKernelGGL(separable_reduce_kernel, dim3((unsigned)(((int64_t)nt * nr + 255) / 256)), dim3(256),
                               0, ctx->stream, ctx->partial_buf.as<double>(), C, nt, nr, cost, R, ta - t0, ra);
            CCAL_HIP(ctx, hipGetLastError());
            // the host tables of this range are reused by the next launch only after this one is done
            CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && kernel_ms) *kernel_ms += ms;
            if (launches) ++*launches;
        }
    }
    return CCAL_OK;
}

// How far a fast-path cost of the grid can lie from the reference-arithmetic cost of the same
// candidate: |fast(c) - exact(c)| <= abs_err + rel_err * fast(c) for EVERY candidate of the grid
// box (not only those of one shard), so that the set {c : fast(c)(1 - rel) <= min fast (1 + rel) +
// 2 abs} is certain to hold the reference-arithmetic arg-min and all its ties.  Per cloud, with
// rho = mean distance of the points from the centre c_j, rn >= |R_j|_2, u = 2^-24, and over the box
//   D    = |c_j| * (sum of the largest angle offsets) + |largest translation offset|
//          >= |F(cand)(c_j,1) - F(base)(c_j,1)|   (|Rz'Ry'Rx' - RzRyRx| <= sum |angle differences|)
//   q_r  <= |q_r(base)| + |row_r(R_j)| D           (centre carried to the base frame, coordinate r)
//   w4   <= |w4(base)| + sum_r |n_r|/|n| |row_r(R_j)| D    (signed distance of the carried centre)
// the budgets are
//   reference side: every coordinate of the transformed point rounded to float (src/helpers.hpp:151):
//                u * sum_r |n_r|/|n| (q_r + rn rho)
//   dense kernels:  w and p - c rounded to float (2u), then the contraction.  FP32-input matrix
//                instruction: an fmaf chain, 4 roundings.  bf16 forms: products of bf16 pairs are exact;
//                the 32 (+2 chained) FP32 accumulation steps inside the matrix pipe are budgeted
//                at one ulp (2u) each, truncation included, and the dropped low*low term at u.
//                All on magnitudes <= rn |p - c| + w4.
//   accumulation:   4 kFlushGroups = 256 FP32 adds per lane partial + 4 in the promotion, of non-negative
//                terms: relative 260u of the sum; one more u for the float the cost is stored as.
//   separable pass: w, p - c, the rotation constant a and the translation constant b rounded to float,
//                an FP32 chain of three fused steps (budgeted as four roundings) on magnitudes <= rn |p - c| + |a|, fixed point
//                (truncation: below one unit; the sign taken from the unrounded value: one more; budgeted as three):
//                u (6 rn rho + 5 A + B) + 3 units
//                of the fixed point (a unit = 2^-30 of the largest possible |residual|, rounded up to
//                a power of two)
//                with A >= |a| (the carried centre's distance at the middle translation: <= w4 above)
//                and B = 2 rn |largest translation offset| >= |b - b_ref|; integer sums add nothing.
// The FP64 arithmetic on both sides is below 1e-12 of these magnitudes; kScreenSlack covers it.
// separable_ok: false when no finite bound on the residuals exists (a non-finite point, a
// non-finite robot matrix): the histogram pass needs one for its fixed point.
struct GridBudget {
    double abs_err = 0.0, rel_err = 0.0;
    bool separable_ok = true;
};

static GridBudget grid_budget(const ccal_ctx* ctx, const CloudSet& cs, const double base_pose[6],
                              const ccal_grid* g, bool separable)
{
    GridBudget gb;
    const int C = cs.n_clouds;
    const double u24 = 0x1p-24;
    const double inf = std::numeric_limits<double>::infinity();
    double theta = 0.0, t2 = 0.0;
    for (int k = 0; k < 6; ++k) {
        const double lo = g->min[k], hi = g->min[k] + (g->count[k] - 1) * g->step[k];
        const double off = std::max(std::fabs(lo), std::fabs(hi));
        if (k < 3) t2 += (off / 1000.0) * (off / 1000.0);
        else theta += off * 3.141592 / 180;
    }
    const double T = std::sqrt(t2) * (1.0 + 1e-9);
    theta *= 1.0 + 1e-9;
    std::vector<double> mats((size_t)C * 12);
    pose_matrices(cs, base_pose, mats.data());
    double pl[4];
    for (int e = 0; e < 4; ++e) pl[e] = (double)ctx->plane[e];
    const double inv_norm = 1.0 / std::sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
    const double k_dense = ctx->fast_kernel == CCAL_FAST_F32 ? 2.0 + 4.0 : 2.0 + 2.0 * 34.0 + 1.0;
    // per cloud: the reference side's term, the bound on |w4| over the box, and the largest |residual| the
    // histogram pass can meet (rotation part + constant, translation excluded)
    std::vector<double> ex_j(C), w4max_j(C), vmax_j(C);
    for (int j = 0; j < C; ++j) {
        const double* M = mats.data() + 12 * (size_t)j;
        const double* B = cs.base_T_flange.data() + 16 * (size_t)j;
        const double* c = cs.centres.data() + 3 * (size_t)j;
        const double rho = cs.radius_mean[j], rn = cs.rot_norm[j] * (1.0 + 1e-9);
        const double cn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        const double D = cn * theta + T;
        double ex = 0.0, w4 = pl[3] * inv_norm, w4_slack = 0.0;
        for (int r = 0; r < 3; ++r) {
            const double q = M[r * 4] * c[0] + M[r * 4 + 1] * c[1] + M[r * 4 + 2] * c[2] + M[r * 4 + 3];
            const double row = std::sqrt(B[r * 4] * B[r * 4] + B[r * 4 + 1] * B[r * 4 + 1] + B[r * 4 + 2] * B[r * 4 + 2]);
            const double wgt = std::fabs(pl[r]) * inv_norm;
            ex += wgt * (std::fabs(q) + row * D + rn * rho);
            w4 += pl[r] * inv_norm * q;
            w4_slack += wgt * row * D;
        }
        ex_j[j] = ex;
        w4max_j[j] = std::fabs(w4) + w4_slack;
        vmax_j[j] = (rn * cs.radius[j] + w4max_j[j]) * (1.0 + 1e-5);
    }
    // The histogram pass lays out ONE fixed point for all clouds, from the largest |residual| over
    // rotations, clouds and points it computes itself (run_separable: wn radius_j + |a|, with wn <= rn and
    // |a| <= the bound on |w4| here).  The largest of the per-cloud bounds is at least that, so the unit
    // charged here -- the same for every cloud -- is at least the unit the pass works in.
    double vmax_all = 0.0;
    for (int j = 0; j < C; ++j)
        if (!(vmax_j[j] <= vmax_all)) vmax_all = vmax_j[j];          // a NaN sticks
    if (separable && !(vmax_all < 1e12)) gb.separable_ok = false;
    const double fx_unit = 1.0 / separable_fx_scale(vmax_all);
    double total = 0.0;
    for (int j = 0; j < C; ++j) {
        const double rho = cs.radius_mean[j], rn = cs.rot_norm[j] * (1.0 + 1e-9);
        double budget;
        if (separable) {
            const double span = 2.0 * rn * T;
            // the fixed point rounds by half a unit and takes the sign from the unrounded value (one more unit)
            budget = u24 * ex_j[j] + u24 * (6.0 * rn * rho + 5.0 * w4max_j[j] + span) + 3.0 * fx_unit;
        } else {
            budget = u24 * ex_j[j] + k_dense * u24 * (rn * rho + w4max_j[j]);
        }
        total += budget;
    }
    gb.abs_err = std::isfinite(total) ? kScreenSlack * total / (double)C : inf;
    gb.rel_err = separable ? 2.0 * u24 : kScreenSlack * (4.0 * kFlushGroups + 5.0) * u24;
    return gb;
}

// Fast-path costs of grid candidates [begin, end) into device memory (*cost_local points at
// candidate `begin`), by the dense kernels or the separable pass as the context says.
static int grid_fast_pass(ccal_ctx* ctx, const CloudSet& cs, const double base_pose[6], const ccal_grid* g,
                          int64_t begin, int64_t end, float** cost_local, double* kernel_ms, int* launches,
                          GridBudget* budget)
{
    const int64_t n_local = end - begin;
    *kernel_ms = 0.0;
    *launches = 0;
    // the separable pass sums 32-bit fixed-point residuals in 64 bits and counts in 32: clouds of 2^25
    // points or more, and scenes without a finite bound on the residuals over the grid box (its fixed
    // point is laid out from that bound), take the dense sweep
    bool separable = ctx->search_mode == CCAL_SEARCH_SEPARABLE;
    for (size_t n : cs.n_pts) separable = separable && n < ((size_t)1 << 25);
    if (separable) {
        *budget = grid_budget(ctx, cs, base_pose, g, true);
        separable = budget->separable_ok;
    }
    if (!separable) *budget = grid_budget(ctx, cs, base_pose, g, false);
    if (separable) {
        const int64_t rot_count = (int64_t)g->count[3] * g->count[4] * g->count[5];
        const int64_t t0 = begin / rot_count, t1 = (end + rot_count - 1) / rot_count;
        // the separable pass fills whole translation rows
        CCAL_HIP(ctx, ctx->cost_buf.ensure((size_t)((t1 - t0) * rot_count) * sizeof(float)));
        *cost_local = ctx->cost_buf.as<float>() + (begin - t0 * rot_count);
        return run_separable(ctx, cs, base_pose, g, t0, t1, ctx->cost_buf.as<float>(), kernel_ms, launches);
    }

    // per-axis tables (host FP64, libm sin/cos)
    int tot_entries = 0;
    for (int k = 0; k < 6; ++k) tot_entries += g->count[k];
    std::vector<double> tbl((size_t)tot_entries * 2, 0.0);
    CCAL_HIP(ctx, ctx->tbl_buf.ensure(tbl.size() * sizeof(double)));
    GridTables gt;
    {
        size_t o = 0;
        double* dbase = ctx->tbl_buf.as<double>();
        for (int k = 0; k < 3; ++k) {
            gt.t[k] = dbase + o;
            for (int i = 0; i < g->count[k]; ++i) tbl[o + i] = grid_axis_value(g, base_pose, k, i);
            o += g->count[k];
        }
        for (int k = 0; k < 3; ++k) {
            gt.sn[k] = dbase + o;
            gt.cs[k] = dbase + o + g->count[3 + k];
            for (int i = 0; i < g->count[3 + k]; ++i) {
                const double a = grid_axis_value(g, base_pose, 3 + k, i);
                sin_cos(a, &tbl[o + i], &tbl[o + g->count[3 + k] + i]);
            }
            o += 2 * (size_t)g->count[3 + k];
        }
        for (int k = 0; k < 6; ++k) gt.count[k] = g->count[k];
    }
    CCAL_HIP(ctx, hipMemcpyAsync(ctx->tbl_buf.p, tbl.data(), tbl.size() * sizeof(double),
                                 hipMemcpyHostToDevice, ctx->stream));
    CCAL_HIP(ctx, ctx->cost_buf.ensure((size_t)n_local * sizeof(float)));
    *cost_local = ctx->cost_buf.as<float>();

    const int64_t chunk = fast_chunk(cs);
    const int n_chunks = (int)((n_local + chunk - 1) / chunk);
    Events evs;
    CCAL_HIP(ctx, evs.create((size_t)n_chunks * 2));
    for (int ci = 0; ci < n_chunks; ++ci) {
        const int64_t c0 = (int64_t)ci * chunk;
        const int64_t n = std::min(chunk, n_local - c0);
        gt.first = begin + c0;
        if (int rc = run_fast_chunk(ctx, cs, gt, n, ctx->cost_buf.as<float>() + c0, evs[2 * ci],
                                    evs[2 * ci + 1]))
            return rc;
    }
    // the table upload above reads `tbl`, and the event times are wanted now
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int ci = 0; ci < n_chunks; ++ci) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, evs[2 * ci], evs[2 * ci + 1]) == hipSuccess) *kernel_ms += ms;
    }
    *launches = n_chunks;
    return CCAL_OK;
}

static int check_grid_call(ccal_ctx* ctx, int which, const double base_pose[6], const ccal_grid* g,
                           int64_t begin, int64_t end, const char* who)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (!base_pose || !g) return fail(ctx, CCAL_ERR_ARG, std::string(who) + ": null argument");
    if (!ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, std::string(who) + ": plane not set");
    const int64_t total = ccal_grid_size(g);
    if (total <= 0) return fail(ctx, CCAL_ERR_ARG, std::string(who) + ": empty grid axis");
    if (begin < 0 || end > total || begin > end)
        return fail(ctx, CCAL_ERR_ARG, std::string(who) + ": candidate range outside the grid");
    return CCAL_OK;
}

extern "C" int ccal_grid_costs(ccal_ctx* ctx, int which, const double base_pose[6], const ccal_grid* g,
                               int64_t begin, int64_t end, float* cost_out)
{
    if (int rc = check_grid_call(ctx, which, base_pose, g, begin, end, "grid_costs")) return rc;
    if (begin == end) return CCAL_OK;
    if (!cost_out) return fail(ctx, CCAL_ERR_ARG, "grid_costs: null argument");
    if (int rc = bind(ctx)) return rc;
    float* cost_local = nullptr;
    double kernel_ms = 0.0;
    int launches = 0;
    GridBudget budget;
    if (int rc = grid_fast_pass(ctx, ctx->sets[which], base_pose, g, begin, end, &cost_local, &kernel_ms,
                                &launches, &budget))
        return rc;
    CCAL_HIP(ctx, download_array(ctx, cost_out, cost_local, (size_t)(end - begin) * sizeof(float)));
    return CCAL_OK;
}

extern "C" int ccal_grid_error_bound(ccal_ctx* ctx, int which, const double base_pose[6], const ccal_grid* g,
                                     double* abs_out, double* rel_out)
{
    if (int rc = check_grid_call(ctx, which, base_pose, g, 0, 0, "grid_error_bound")) return rc;
    if (!abs_out || !rel_out) return fail(ctx, CCAL_ERR_ARG, "grid_error_bound: null argument");
    const CloudSet& cs = ctx->sets[which];
    bool separable = ctx->search_mode == CCAL_SEARCH_SEPARABLE;
    for (size_t n : cs.n_pts) separable = separable && n < ((size_t)1 << 25);
    GridBudget b;
    if (separable) {
        b = grid_budget(ctx, cs, base_pose, g, true);
        separable = b.separable_ok;
    }
    if (!separable) b = grid_budget(ctx, cs, base_pose, g, false);
    *abs_out = b.abs_err;
    *rel_out = b.rel_err;
    return CCAL_OK;
}

extern "C" int ccal_grid_search(ccal_ctx* ctx, int which, const double base_pose[6],
                                const ccal_grid* g, int64_t begin, int64_t end,
                                int64_t* best_index, double* best_cost, double best_pose[6],
                                ccal_grid_stats* stats)
{
    if (int rc = check_grid_call(ctx, which, base_pose, g, begin, end, "grid_search")) return rc;
    if (!best_index || !best_cost || !best_pose)
        return fail(ctx, CCAL_ERR_ARG, "grid_search: null argument");
    if (stats) std::memset(stats, 0, sizeof *stats);
    *best_index = -1;
    *best_cost = 1e9;  // :531
    std::memcpy(best_pose, base_pose, 6 * sizeof(double));
    const int64_t n_local = end - begin;
    if (n_local == 0) return CCAL_OK;
    if (int rc = bind(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];

    Events ev;
    CCAL_HIP(ctx, ev.create(2));
    const hipEvent_t ev_begin = ev[0], ev_end = ev[1];     // ev_end: after the last re-score pass
    CCAL_HIP(ctx, hipEventRecord(ev_begin, ctx->stream));
    float* cost_local = nullptr;
    double kernel_ms = 0.0;
    int launches = 0;
    GridBudget budget;
    if (int rc = grid_fast_pass(ctx, cs, base_pose, g, begin, end, &cost_local, &kernel_ms, &launches, &budget))
        return rc;

    // arg-min of the fast costs, then every candidate the error budget cannot exclude:
    //   fast(c) (1 - rel) <= fast_min (1 + rel) + 2 abs
    const int n_stage1 = (int)std::min<int64_t>(1024, (n_local + 255) / 256);
    CCAL_HIP(ctx, ctx->minrec_buf.ensure((size_t)(n_stage1 + 1) * sizeof(MinRec)));
    CCAL_HIP(ctx, ctx->list_buf.ensure((size_t)kBandCap * sizeof(int64_t)));
    CCAL_HIP(ctx, ctx->count_buf.ensure(sizeof(unsigned int)));
    MinRec* recs = ctx->minrec_buf.as<MinRec>();
    const double band_rel = budget.rel_err, band_abs = 2.0 * budget.abs_err;
    auto select = [&](int64_t w0, int64_t wn, unsigned int* n_out) -> int {
        CCAL_HIP(ctx, hipMemsetAsync(ctx->count_buf.p, 0, sizeof(unsigned int), ctx->stream));
        const int blocks = (int)std::min<int64_t>(1024, (wn + 255) / 256);
        hipLaunchKernelGGL(select_band_kernel, dim3(blocks), dim3(256), 0, ctx->stream, cost_local + w0, wn, w0,
                           recs + n_stage1, band_rel, band_abs, ctx->list_buf.as<int64_t>(), kBandCap,
                           ctx->count_buf.as<unsigned int>());
        CCAL_HIP(ctx, hipGetLastError());
        CCAL_HIP(ctx, hipMemcpyAsync(n_out, ctx->count_buf.p, sizeof *n_out, hipMemcpyDeviceToHost, ctx->stream));
        return CCAL_OK;
    };
    hipLaunchKernelGGL(argmin_stage1_kernel, dim3(n_stage1), dim3(256), 0, ctx->stream,
                       cost_local, n_local, recs);
    hipLaunchKernelGGL(argmin_stage2_kernel, dim3(1), dim3(256), 0, ctx->stream, recs, n_stage1,
                       recs + n_stage1);
    unsigned int n_band = 0;
    if (int rc = select(0, n_local, &n_band)) return rc;
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));

    // (Non-finite fast costs are kept by the selection -- the comparison that drops a candidate is
    // false for them -- and the reference arithmetic decides: a NaN or infinite cost never passes
    // the strict '<' against 1e9, so a set that yields nothing else returns index -1.)
    // re-score with the reference arithmetic; strict '<' in index order (:542)
    double error_min = 1e9;
    int64_t best = -1, n_rescored = 0;
    std::vector<int64_t> list;
    std::vector<double> poses, exact;
    auto rescore = [&](unsigned int n) -> int {
        list.resize(n);
        CCAL_HIP(ctx, hipMemcpy(list.data(), ctx->list_buf.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
        std::sort(list.begin(), list.end());
        poses.resize((size_t)n * 6);
        exact.resize(n);
        for (unsigned i = 0; i < n; ++i) ccal_grid_pose(g, base_pose, begin + list[i], poses.data() + 6 * (size_t)i);
        if (int rc = ccal::eval_exact(ctx, which, (int)n, poses.data(), nullptr, exact.data(), nullptr, nullptr))
            return rc;
        for (unsigned i = 0; i < n; ++i)
            if (exact[i] < error_min) {
                error_min = exact[i];
                best = list[i];
                std::memcpy(best_pose, poses.data() + 6 * (size_t)i, 6 * sizeof(double));
            }
        n_rescored += n;
        return CCAL_OK;
    };
    if (n_band <= kBandCap) {
        if (int rc = rescore(n_band)) return rc;
    } else {
        // more near-minimum candidates than one list holds (flat or repeated grids, an infinite
        // budget): the range is walked in windows no longer than the list, in index order
        for (int64_t w0 = 0; w0 < n_local; w0 += kBandCap) {
            const int64_t wn = std::min<int64_t>(kBandCap, n_local - w0);
            unsigned int n_win = 0;
            if (int rc = select(w0, wn, &n_win)) return rc;
            CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (n_win == 0) continue;
            if (int rc = rescore(n_win)) return rc;
        }
    }
    if (best >= 0) {
        *best_index = begin + best;
        *best_cost = error_min;
    } else {
        std::memcpy(best_pose, base_pose, 6 * sizeof(double));
    }
    // device time of the whole call: fast pass, selection (every window of it) and the exact re-scores
    float total_ms = 0.f;
    CCAL_HIP(ctx, hipEventRecord(ev_end, ctx->stream));
    CCAL_HIP(ctx, hipEventSynchronize(ev_end));
    (void)hipEventElapsedTime(&total_ms, ev_begin, ev_end);
    if (stats) {
        stats->n_candidates = n_local;
        stats->n_rescored = n_rescored;
        stats->residuals = (double)n_local * (double)cs.total_pts;
        stats->kernel_ms = kernel_ms;
        stats->kernel_launches = launches;
        stats->total_ms = total_ms;
        stats->band_abs = budget.abs_err;
        stats->band_rel = budget.rel_err;
    }
    return CCAL_OK;
}

