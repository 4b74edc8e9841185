// ransac_impl.hpp -- batched RANSAC plane fit (included at the end of api_fit.hip).
//
// Parameters follow the one RANSAC call in the reference tree, getPlane at
// src/cloud_analyzer.cpp:103-121 (SACMODEL_PLANE, SAC_RANSAC, distance threshold
// 0.01, setOptimizeCoefficients(true)); PCL itself is not part of that tree, so
// the sampling scheme below is this library's own definition:
//   1. n_hyp three-point hypotheses from a counter-based generator (kernel),
//   2. every hypothesis scored against every point in ONE pass of the fast
//      matrix-core contraction, counting the points that are inliers for certain
//      and those that may be (the FP32 residual within a proven error bound of
//      the threshold),
//   3. every hypothesis that those two counts cannot exclude re-counted with the
//      reference-arithmetic distance (strict <); most inliers wins, lowest
//      hypothesis index on ties -- the winner of a sequential scan in that
//      arithmetic, whatever the data (points exactly at the threshold included),
//   4. winner refitted on its inliers: centroid + smallest-eigenvalue
//      eigenvector of their covariance (what PCL's coefficient optimisation does).
#pragma once

namespace {

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (FP64).
void jacobi3(double A[3][3], double V[3][3])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

template <int PB>
void launch_exact_count(ccal_ctx* ctx, const CloudSet& cs, int n_planes, double thr)
{
    const int grid = std::min(cs.n_blocks, 8 * ctx->n_cu);
    hipLaunchKernelGGL(exact_count_kernel<PB>, dim3(grid), dim3(256), 0, ctx->stream,
                       cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.blocks.as<BlockDesc>(), cs.n_blocks, ctx->mats_buf.as<double>(),
                       ctx->planes_buf.as<double>(), n_planes, thr,
                       ctx->count_buf.as<unsigned long long>());
}

// Hypotheses re-counted exactly per pass over the points.  The kernel keeps its planes in registers: with 4
// it holds four waves per SIMD and a pass over 10^8 points takes 0.52 ms (0.13 ms per hypothesis); with 8, 1.18 ms;
// with 16 it is down to one wave per SIMD and 3.9 ms.  The first pass only has to set a high mark.
constexpr int kRecount = 4;

} // namespace

extern "C" int ccal_ransac_plane(ccal_ctx* ctx, int which, const double pose[6], int n_hyp,
                                 float threshold, uint64_t seed, float plane_out[4],
                                 int64_t* inliers_out, int* best_hyp_out)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (!pose || !plane_out || n_hyp <= 0 || !(threshold > 0.f))
        return fail(ctx, CCAL_ERR_ARG, "ransac_plane: bad argument");
    if (int rc = bind(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    const int C = cs.n_clouds;
    if (cs.total_pts < 3) return fail(ctx, CCAL_ERR_ARG, "ransac_plane: fewer than three points");

    std::vector<double> mats((size_t)C * 12);
    pose_matrices(cs, pose, mats.data());
    CCAL_HIP(ctx, ctx->mats_buf.ensure(mats.size() * sizeof(double)));
    CCAL_HIP(ctx, hipMemcpyAsync(ctx->mats_buf.p, mats.data(), mats.size() * sizeof(double),
                                 hipMemcpyHostToDevice, ctx->stream));

    // 1. hypotheses
    CCAL_HIP(ctx, ctx->rec_buf.ensure((size_t)n_hyp * kHypStride * sizeof(double)));
    hipLaunchKernelGGL(ransac_hyp_kernel, dim3((n_hyp + 255) / 256), dim3(256), 0, ctx->stream,
                       cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.cloud_pt_off_dev.as<int64_t>(), C, (int64_t)cs.total_pts,
                       ctx->mats_buf.as<double>(), seed, n_hyp, ctx->rec_buf.as<double>());

    // 2. fast scoring of every (hypothesis, point) pair, with a bound on what FP32 can misplace.
    // The reference-arithmetic distance of a point (exact_distance) and its FP32 residual both
    // approximate r = (n.(M p) + d) / |n|.  Per cloud j, with u = 2^-24, c the cloud's centre,
    // rho the largest |p - c|, rn >= |R|_2 (R the 3x3 part of M), m = |M (c,1)|, Q = m + rn rho >= |M p|:
    //   reference:  every coordinate of M p rounded to float, |n_r|/|n| <= 1:           u Q
    //   fast path:  w and p - c rounded to float (2u); the contraction on the bf16 matrix pipe: products
    //               of bf16 pairs are exact, its 32 (+2) FP32 accumulation steps are budgeted at one ulp
    //               (2u) each, truncation included, the dropped low*low term at u -- all on magnitudes
    //               of at most |w123| |p - c| + |w4|, |w123| <= rn, |w4| = |r(c)| <= m + |d|/|n|:
    //                                                                  71 u (rn rho + m + |d|/|n|)
    // A hypothesis passes through a transformed point of the set, so |d|/|n| <= max_j Q_j.  The
    // FP64 arithmetic on either side stays below 2^-45 of these magnitudes; the slack covers it.
    // The matrix pipe may flush denormal operands and results: an absolute 1e-30 covers that.
    constexpr double kSlack = 1.25;
    const double u24 = 0x1p-24;
    std::vector<double> Q((size_t)C), mj((size_t)C);
    double Qmax = 0.0;
    for (int j = 0; j < C; ++j) {
        const double* M = mats.data() + 12 * (size_t)j;
        const double* c = cs.centres.data() + 3 * (size_t)j;
        double m2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            const double q = M[r * 4] * c[0] + M[r * 4 + 1] * c[1] + M[r * 4 + 2] * c[2] + M[r * 4 + 3];
            m2 += q * q;
        }
        mj[j] = std::sqrt(m2);
        Q[j] = mj[j] + cs.rot_norm[j] * (1.0 + 1e-9) * cs.radius[j];
        if (!(Q[j] <= Qmax)) Qmax = Q[j];      // a NaN or infinity sticks
    }
    std::vector<f32x2> thr_tab((size_t)C);
    for (int j = 0; j < C; ++j) {
        const double rn = cs.rot_norm[j] * (1.0 + 1e-9);
        double e = kSlack * u24 * (Q[j] + 71.0 * (rn * cs.radius[j] + mj[j] + Qmax)) + 1e-30;
        if (!(e >= 0.0) || !std::isfinite(e)) e = std::numeric_limits<double>::infinity();
        // outward rounding to float, one more step for the conversion itself
        float lo = (float)((double)threshold - e), hi = (float)((double)threshold + e);
        lo = std::nextafterf(lo, -std::numeric_limits<float>::infinity());
        hi = std::nextafterf(hi, std::numeric_limits<float>::infinity());
        if (!((double)lo <= (double)threshold - e)) lo = -std::numeric_limits<float>::infinity();
        if (!((double)hi >= (double)threshold + e)) hi = std::numeric_limits<float>::infinity();
        thr_tab[j] = f32x2{lo, hi};
    }
    FastLaunch f = plan_fast(ctx, cs, n_hyp, kCandPerBlock);
    const size_t part_words = (size_t)f.n_splits * f.n_padded;
    CCAL_HIP(ctx, ctx->w_buf.ensure((size_t)C * f.n_ctiles * 64 * sizeof(float)));
    CCAL_HIP(ctx, ctx->partial_buf.ensure(part_words * sizeof(unsigned long long)));
    CCAL_HIP(ctx, ctx->cost_buf.ensure((size_t)n_hyp * sizeof(unsigned long long)));
    CCAL_HIP(ctx, ctx->tbl_buf.ensure(thr_tab.size() * sizeof(f32x2)));
    CCAL_HIP(ctx, hipMemcpyAsync(ctx->tbl_buf.p, thr_tab.data(), thr_tab.size() * sizeof(f32x2),
                                 hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(wgen_planes_kernel, dim3((unsigned)((f.n_padded + 255) / 256)), dim3(256), 0,
                       ctx->stream, ctx->rec_buf.as<double>(), n_hyp, f.n_ctiles, C,
                       ctx->mats_buf.as<double>(), cs.centres_dev.as<double>(),
                       ctx->w_buf.as<float>());
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[0], ctx->stream));
    hipLaunchKernelGGL(inlier_mfma_kernel<kNA>, dim3((unsigned)f.cand_blocks, f.n_splits),
                       dim3(256), 0, ctx->stream, ctx->w_buf.as<float>(), cs.groups.as<f32x4>(),
                       cs.group_off_dev.as<int>(), ctx->partial_buf.as<unsigned long long>(),
                       ctx->tbl_buf.as<f32x2>(), C, f.n_ctiles, f.n_splits);
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[1], ctx->stream));
    hipLaunchKernelGGL(reduce_splits_u64_kernel, dim3((n_hyp + 255) / 256), dim3(256), 0,
                       ctx->stream, ctx->partial_buf.as<unsigned long long>(), f.n_padded,
                       f.n_splits, (int64_t)n_hyp, ctx->cost_buf.as<unsigned long long>());
    CCAL_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> fast((size_t)n_hyp);
    std::vector<double> planes((size_t)n_hyp * kHypStride);
    CCAL_HIP(ctx, hipMemcpyAsync(fast.data(), ctx->cost_buf.p, fast.size() * sizeof(unsigned long long),
                                 hipMemcpyDeviceToHost, ctx->stream));
    CCAL_HIP(ctx, hipMemcpyAsync(planes.data(), ctx->rec_buf.p, planes.size() * sizeof(double),
                                 hipMemcpyDeviceToHost, ctx->stream));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->prof_ev[0], ctx->prof_ev[1]) == hipSuccess) {
            ctx->prof.ransac_score_ms += ms;
            ctx->prof.ransac_tests += (double)f.n_padded * (double)cs.n_groups * kGroupPoints;
            ++ctx->prof.ransac_calls;
        }
    }

    // (reference-arithmetic inlier count of h) <= possible[h].  Invalid (collinear) hypotheses have
    // w = 0 and count everything: drop them.
    unsigned long long* possible = fast.data();
    std::vector<int> order;
    for (int h = 0; h < n_hyp; ++h) {
        if (planes[(size_t)h * kHypStride + 5] == 0.0) continue;
        possible[h] -= (unsigned long long)cs.n_pad_points;
        order.push_back(h);
    }
    if (order.empty()) return fail(ctx, CCAL_ERR_ARG, "ransac_plane: every sample was collinear");

    // 3. exact recount, kRecount hypotheses per pass over the points, with the reference arithmetic.
    // First the ones with the highest bounds (lowest index first among equals), which sets a high mark;
    // then -- in index order -- whatever that mark (most inliers so far, lowest index on ties) still cannot
    // exclude: a hypothesis whose bound is below the mark, or equal to it at a higher index, cannot win.
    std::vector<int> leaders = order;
    std::stable_sort(leaders.begin(), leaders.end(),
                     [&](int a, int b) { return possible[a] > possible[b]; });
    std::vector<int> rest;
    if (leaders.size() > (size_t)kRecount) {
        rest.assign(leaders.begin() + kRecount, leaders.end());
        leaders.resize(kRecount);
        std::sort(rest.begin(), rest.end());
    }
    int best_h = -1;
    unsigned long long best_cnt = 0;
    std::vector<double> pl((size_t)kRecount * kPlaneStride);
    std::vector<unsigned long long> cnt(kRecount);
    CCAL_HIP(ctx, ctx->planes_buf.ensure(pl.size() * sizeof(double)));
    CCAL_HIP(ctx, ctx->count_buf.ensure(kRecount * sizeof(unsigned long long)));
    size_t next = 0;
    std::vector<int> batch;
    for (;;) {
        batch.clear();
        if (best_h < 0)
            batch = leaders;
        else
            for (; next < rest.size() && batch.size() < (size_t)kRecount; ++next) {
                const int h = rest[next];
                if (possible[h] > best_cnt || (possible[h] == best_cnt && h < best_h)) batch.push_back(h);
            }
        if (batch.empty()) break;
        const int nb = (int)batch.size();
        for (int k = 0; k < kRecount; ++k) {
            const int h = batch[k < nb ? k : 0];
            double* row = pl.data() + (size_t)k * kPlaneStride;
            for (int e = 0; e < 5; ++e) row[e] = planes[(size_t)h * kHypStride + e];
            row[5] = 1.0 / row[4];                                        // rounded reciprocal of the norm
            row[6] = (row[4] > 0x1p-300 && row[4] < 0x1p300) ? 1.0 : 0.0; // the range its corrections are exact in
            row[7] = 0.0;
        }
        CCAL_HIP(ctx, hipMemcpyAsync(ctx->planes_buf.p, pl.data(), pl.size() * sizeof(double),
                                     hipMemcpyHostToDevice, ctx->stream));
        CCAL_HIP(ctx, hipMemsetAsync(ctx->count_buf.p, 0, kRecount * sizeof(unsigned long long), ctx->stream));
        if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[0], ctx->stream));
        launch_exact_count<kRecount>(ctx, cs, nb, (double)threshold);
        if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[1], ctx->stream));
        CCAL_HIP(ctx, hipGetLastError());
        CCAL_HIP(ctx, hipMemcpyAsync(cnt.data(), ctx->count_buf.p, kRecount * sizeof(unsigned long long),
                                     hipMemcpyDeviceToHost, ctx->stream));
        CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->profiling) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->prof_ev[0], ctx->prof_ev[1]) == hipSuccess) {
                ctx->prof.ransac_recount_ms += ms;
                ctx->prof.ransac_recounted += nb;
            }
        }
        for (int k = 0; k < nb; ++k) {
            const int h = batch[k];
            if (best_h < 0 || cnt[k] > best_cnt || (cnt[k] == best_cnt && h < best_h)) {
                best_cnt = cnt[k];
                best_h = h;
            }
        }
    }

    // 4. refit on the winner's inliers
    const double* bp = &planes[(size_t)best_h * kHypStride];
    InlierMask mask;
    for (int e = 0; e < 5; ++e) mask.pl[e] = bp[e];
    mask.thr = (double)threshold;
    double org[3];
    for (int r = 0; r < 3; ++r)
        org[r] = mats[r * 4 + 0] * cs.centres[0] + mats[r * 4 + 1] * cs.centres[1] +
                 mats[r * 4 + 2] * cs.centres[2] + mats[r * 4 + 3];
    CCAL_HIP(ctx, ctx->psum_buf.ensure((size_t)cs.n_blocks * kMoments * sizeof(double)));
    CCAL_HIP(ctx, ctx->osum_buf.ensure(kMoments * sizeof(double)));
    hipLaunchKernelGGL(moments_kernel, dim3(moment_rows(cs.n_blocks)), dim3(256), 0, ctx->stream,
                       cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.blocks.as<BlockDesc>(), cs.n_blocks, ctx->mats_buf.as<double>(), org[0], org[1],
                       org[2], mask, ctx->psum_buf.as<double>());
    hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                       ctx->psum_buf.as<double>(), moment_rows(cs.n_blocks), ctx->osum_buf.as<double>());
    CCAL_HIP(ctx, hipGetLastError());
    double m[kMoments];
    CCAL_HIP(ctx, hipMemcpyAsync(m, ctx->osum_buf.p, sizeof m, hipMemcpyDeviceToHost, ctx->stream));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double n[3] = {bp[0], bp[1], bp[2]}, d = bp[3];
    if (m[0] >= 3.0) {
        const double N = m[0];
        const double cx = m[1] / N, cy = m[2] / N, cz = m[3] / N;
        double A[3][3] = {{m[4] / N - cx * cx, m[5] / N - cx * cy, m[7] / N - cx * cz},
                          {m[5] / N - cx * cy, m[6] / N - cy * cy, m[8] / N - cy * cz},
                          {m[7] / N - cx * cz, m[8] / N - cy * cz, m[9] / N - cz * cz}};
        double V[3][3];
        jacobi3(A, V);
        int k = 0;
        if (A[1][1] < A[k][k]) k = 1;
        if (A[2][2] < A[k][k]) k = 2;
        double e[3] = {V[0][k], V[1][k], V[2][k]};
        const double len = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        if (len > 0) {
            const double sgn = e[2] < 0 ? -1.0 : 1.0;
            for (int i = 0; i < 3; ++i) n[i] = sgn * e[i] / len;
            d = -(n[0] * (cx + org[0]) + n[1] * (cy + org[1]) + n[2] * (cz + org[2]));
        }
    }
    plane_out[0] = (float)n[0];
    plane_out[1] = (float)n[1];
    plane_out[2] = (float)n[2];
    plane_out[3] = (float)d;
    if (inliers_out) *inliers_out = (int64_t)best_cnt;
    if (best_hyp_out) *best_hyp_out = best_h;
    return CCAL_OK;
}
