// api_exact.hip -- the objective in the reference's own arithmetic (ccal_eval), the per-point report,
// the FP32 screening pass with its error bound, and the profiling counters of these kernels.
#include "ccal_internal.hpp"
#include "kernels_exact.hpp"

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

// ---------------------------------------------------------------------------
// exact objective
// ---------------------------------------------------------------------------

struct ExactTables {
    const double* mats;     // [pose][cloud][12]
    const double* planes;   // [pose][kPlaneStride]
    double* sum;            // [pose][cloud]
    double* max;            // [pose][cloud] or null
};

template <int W, bool kWantMax>
static void launch_exact_chains_w(ccal_ctx* ctx, const CloudSet& cs, int n_poses, const ExactTables& t)
{
    hipLaunchKernelGGL((exact_chain_kernel<W, kWantMax>), dim3(n_poses * cs.n_clouds), dim3(64 * W), 0,
                       ctx->stream, cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.cloud_pt_off_dev.as<int64_t>(), t.mats, t.planes, n_poses, cs.n_clouds,
                       (int64_t)cs.total_pts, t.sum, kWantMax ? t.max : nullptr, cs.cloud_seg_off_dev.as<int64_t>(),
                       cs.seg_sums.as<double>());
}

// Workgroup width by chain count (measured on 20 x 300k points: 16 waves win below ~128
// chains, 8 up to ~512, 4 beyond).
static void launch_exact_chains(ccal_ctx* ctx, const CloudSet& cs, int n_poses, bool want_max,
                                const ExactTables& t)
{
    const int64_t chains = (int64_t)n_poses * cs.n_clouds;
    const int64_t per_cu = chains / std::max(1, ctx->n_cu);
    if (want_max) {
        if (per_cu < 1) launch_exact_chains_w<16, true>(ctx, cs, n_poses, t);
        else if (per_cu < 2) launch_exact_chains_w<8, true>(ctx, cs, n_poses, t);
        else launch_exact_chains_w<4, true>(ctx, cs, n_poses, t);
    } else {
        if (per_cu < 1) launch_exact_chains_w<16, false>(ctx, cs, n_poses, t);
        else if (per_cu < 2) launch_exact_chains_w<8, false>(ctx, cs, n_poses, t);
        else launch_exact_chains_w<4, false>(ctx, cs, n_poses, t);
    }
}

// Results of the small requests arrive in pinned host memory, one 8-byte word per (pose, cloud) written
// by the last kernel of the request.  The host marks every word before the launch and then watches the
// words themselves: a word that has changed is a result (8-byte device stores to host memory are not torn),
// and the wait ends a few microseconds before the stream's completion event would be seen (measured on a
// descent of 919 iterations: 362 -> 353 us per iteration).  The mark is a NaN whose payload no arithmetic
// produces.  Every kSpinsPerQuery looks the event is asked as well, so a failed launch ends the wait with
// its error instead of hanging, and a finished one with marks left over (none should be) ends it too.
constexpr unsigned long long kUnwritten = 0x7ff8c0de5ca1ab1eull;
constexpr int kSpinsPerQuery = 1 << 16;

static void mark_unwritten(double* words, size_t n)
{
    for (size_t i = 0; i < n; ++i) std::memcpy(&words[i], &kUnwritten, sizeof(double));
}

static hipError_t wait_for_words(const double* words, size_t n, hipEvent_t done)
{
    const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(words);
    int spins = 0;
    for (size_t i = n; i-- > 0;) {           // the last words are written last, more often than not
        while (w[i] == kUnwritten) {
            __builtin_ia32_pause();
            if (++spins >= kSpinsPerQuery) {
                spins = 0;
                const hipError_t e = hipEventQuery(done);
                if (e != hipErrorNotReady) return e == hipSuccess ? hipEventSynchronize(done) : e;
            }
        }
    }
    return hipSuccess;
}

// The auxiliary streams and events of the requests in flight; made at first use, kept with the context.
static int ensure_job_objects(ccal_ctx* ctx)
{
    // the finishing passes' streams take precedence when workgroup slots free up
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    for (hipStream_t& s : ctx->aux_stream)
        if (!s) CCAL_HIP(ctx, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_high));
    for (hipEvent_t& e : ctx->job_ev)
        if (!e) CCAL_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!ctx->screen_ev) CCAL_HIP(ctx, hipEventCreateWithFlags(&ctx->screen_ev, hipEventDisableTiming));
    for (hipEvent_t& e : ctx->exact_prof_ev)
        if (!e) CCAL_HIP(ctx, hipEventCreate(&e));
    for (hipEvent_t& e : ctx->screen_prof_ev)
        if (!e) CCAL_HIP(ctx, hipEventCreate(&e));
    return CCAL_OK;
}

// The buffers, streams and events a descent's requests use, made when a cloud set is uploaded rather than
// inside the first requests (pinned and device allocations cost 0.1-0.9 ms each, about 5 ms of a first
// descent): sized for requests of up to kPresizePoses poses, grown later like any other when that is not
// enough.  The segment records are left to the first request when they would take more than 256 MiB.
constexpr int kPresizePoses = 48;

int ccal::presize_request_buffers(ccal_ctx* ctx, const CloudSet& cs)
{
    if (int rc = ensure_job_objects(ctx)) return rc;
    const size_t nb = kPresizePoses, C = (size_t)cs.n_clouds;
    const size_t n_mats = nb * C * 12, n_recs = nb * kPoseRec, n_planes = nb * kPlaneStride, n_out = nb * C;
    CCAL_HIP(ctx, ctx->exact_in.ensure((n_recs + n_planes + n_out) * sizeof(double)));
    CCAL_HIP(ctx, ctx->exact_out.ensure(2 * n_out * sizeof(double)));
    CCAL_HIP(ctx, ctx->exact_mats.ensure((n_mats + n_planes + 2 * n_out) * sizeof(double)));
    CCAL_HIP(ctx, ctx->split_first.ensure(n_out * sizeof(double)));
    const size_t map_bytes = nb * (size_t)cs.total_segs * sizeof(SegMap);
    if (map_bytes <= ((size_t)256 << 20)) CCAL_HIP(ctx, ctx->split_maps.ensure(map_bytes));
    CCAL_HIP(ctx, ctx->screen_in.ensure(nb * kScreenRec * sizeof(double)));
    CCAL_HIP(ctx, ctx->screen_out.ensure(2 * n_out * sizeof(double)));
    CCAL_HIP(ctx, ctx->screen_partial.ensure((size_t)cs.n_blocks * kScreenParts * nb * sizeof(double)));
    CCAL_HIP(ctx, ctx->screen_w.ensure(n_out * 4 * sizeof(float)));
    return CCAL_OK;
}

// parts per chain of a split launch: enough workgroups to fill the chip several times over (measured
// on 620 chains: 4 parts 0.578 s, 8 parts 0.562 s, 12 parts 0.560 s per descent)
static int split_parts(const ccal_ctx* ctx, int64_t chains)
{
    // at least 16 workgroups per compute unit, then as many more as still fit the last round of what the
    // chip holds at once (five workgroups of the part kernel per compute unit), so that no round is half empty
    int64_t K = std::max<int64_t>(8, (16 * (int64_t)ctx->n_cu + chains - 1) / chains);
    const int64_t resident = 5 * (int64_t)ctx->n_cu;
    const int64_t rounds = (chains * K + resident - 1) / resident;
    K = std::max<int64_t>(K, rounds * resident / chains);
    return (int)std::min<int64_t>(64, K);
}

int ccal::exact_begin(ccal_ctx* ctx, int which, int nb, const double* poses6, const double* planes4, int n_first,
                      bool want_max, bool allow_split, const double* expected_sums)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (nb <= 0 || nb > 4096 || !poses6) return fail(ctx, CCAL_ERR_ARG, "exact_begin: 1 to 4096 poses");
    if (ctx->exact_job.active) return fail(ctx, CCAL_ERR_STATE, "exact_begin: a request is in flight");
    if (!planes4 && !ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "eval: plane not set");
    if (int rc = bind(ctx)) return rc;
    if (int rc = ensure_job_objects(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    const int C = cs.n_clouds;
    ccal_ctx::ExactJob& job = ctx->exact_job;
    job.host0 = ctx->profiling ? host_ms() : 0.0;

    // Pose records, planes and results live in pinned host memory that the kernels address directly: a
    // request is a few launches and a wait, with no copy call on either side.
    const size_t n_mats = (size_t)nb * C * 12, n_recs = (size_t)nb * kPoseRec, n_planes = (size_t)nb * kPlaneStride,
                 n_out = (size_t)nb * C;
    CCAL_HIP(ctx, ctx->exact_in.ensure((n_recs + n_planes + n_out) * sizeof(double)));
    CCAL_HIP(ctx, ctx->exact_out.ensure(2 * n_out * sizeof(double)));
    CCAL_HIP(ctx, ctx->exact_mats.ensure((n_mats + n_planes + 2 * n_out) * sizeof(double)));
    double* recs = ctx->exact_in.as<double>();
    double* planes_host = recs + n_recs;
    double* expected_host = planes_host + n_planes;
    double* mats = ctx->exact_mats.as<double>();
    double* planes = mats + n_mats;          // the device's copy, made by pose_table_kernel
    double* guess_scale = planes + n_planes; // [pose][cloud][2], made by pose_table_kernel
    if (expected_sums) std::memcpy(expected_host, expected_sums, n_out * sizeof(double));
    double* hsum = ctx->exact_out.as<double>();
    double* hmax = hsum + n_out;
    for (int k = 0; k < nb; ++k) {
        // the host takes the sines and cosines (libm's sincos, as the reference's binary does); the
        // matrices of every (pose, cloud) are composed on the device from them (pose_table_kernel)
        const double* pose = poses6 + 6 * (size_t)k;
        const SinCos3 sc = sincos_of_pose(pose);
        double* r = recs + (size_t)k * kPoseRec;
        for (int e = 0; e < 3; ++e) { r[e] = pose[e]; r[3 + e] = sc.s[e]; r[6 + e] = sc.c[e]; }
        double* pl = planes_host + (size_t)k * kPlaneStride;
        if (planes4) {
            for (int e = 0; e < 4; ++e) pl[e] = planes4[4 * (size_t)k + e];
        } else {
            for (int e = 0; e < 4; ++e) pl[e] = (double)ctx->plane[e];
        }
        // sqrt(pow(a,2)+pow(b,2)+pow(c,2)) as at :123 / :128
        pl[4] = std::sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
        pl[5] = 1.0 / pl[4];
        pl[6] = (pl[4] > 0x1p-300 && pl[4] < 0x1p300) ? 1.0 : 0.0;
        pl[7] = 0.0;
    }
    // Requests too small to fill the chip with whole chains cut every chain into K parts (see
    // split_chain_kernel); hmax's slots then carry the number of segments summed twice.
    const int64_t chains = (int64_t)nb * C;
    // (not before some evaluation of this set has left its running sums behind: without them
    // every segment would be summed again by one wave)
    // (a caller that knows what the chains will sum to has poses that belong together -- one descent's -- and
    // then the parts win at any size: 5.4 -> 4.6 ms per iteration of a descent over 100 x 10^6 points; the
    // requests of several descents in lock step are summed faster as whole chains once they fill the chip)
    const bool split = ctx->chain_split && allow_split && !want_max &&
                       (chains < 8 * (int64_t)ctx->n_cu || expected_sums != nullptr) &&
                       cs.seg_sums_valid &&
                       (int64_t)nb * cs.total_segs * (int64_t)sizeof(SegMap) <= ((int64_t)1 << 30);   // records: at most 1 GiB
    const bool prof = ctx->profiling;
    // (with profiling on the waits go through the events, whose times are what is asked for)
    job.watch_words = split && !prof;
    if (job.watch_words) mark_unwritten(hsum, 2 * n_out);
    hipStream_t S = ctx->stream;
    hipLaunchKernelGGL(pose_table_kernel, dim3((unsigned)((chains + 63) / 64)), dim3(64), 0, S, recs,
                       cs.base_dev.as<double>(), nb, C, mats, planes_host, planes,
                       expected_sums && cs.seg_sums_valid ? expected_host : nullptr, cs.seg_sums.as<double>(),
                       cs.cloud_seg_off_dev.as<int64_t>(), guess_scale);
    if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->exact_prof_ev[0], S));
    job.which = which;
    job.nb = nb;
    job.split = split;
    job.want_max = want_max;
    job.n_first = split && n_first > 0 && n_first < nb ? n_first : 0;
    if (split) {
        CCAL_HIP(ctx, ctx->split_maps.ensure((size_t)nb * cs.total_segs * sizeof(SegMap)));
        CCAL_HIP(ctx, ctx->split_first.ensure(n_out * sizeof(double)));
        // poses [p0, p0 + np) of the request: their chains cut into parts, on the context's stream
        auto chains_of = [&](int p0, int np, int K) {
            hipLaunchKernelGGL(split_chain_kernel, dim3((unsigned)((int64_t)np * C * K)), dim3(64 * kSplitWaves), 0, S,
                               cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(), cs.cloud_pt_off_dev.as<int64_t>(),
                               cs.cloud_seg_off_dev.as<int64_t>(), mats + (size_t)p0 * C * 12, planes + (size_t)p0 * kPlaneStride,
                               np, C, (int64_t)cs.total_pts, K, cs.seg_sums.as<double>(), guess_scale + 2 * (size_t)p0 * C,
                               cs.total_segs, ctx->split_maps.as<SegMap>() + (size_t)p0 * cs.total_segs,
                               ctx->split_first.as<double>() + (size_t)p0 * C);
        };
        // ... and the pass that applies the parts to the true sums, one wave per chain, on `stream`
        auto finish_of = [&](hipStream_t stream, int p0, int np, int K, int ref_pose) {
            hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)((int64_t)np * C)), dim3(64), 0, stream, cs.xs.as<float>(),
                               cs.ys.as<float>(), cs.zs.as<float>(), cs.cloud_pt_off_dev.as<int64_t>(),
                               cs.cloud_seg_off_dev.as<int64_t>(), mats + (size_t)p0 * C * 12, planes + (size_t)p0 * kPlaneStride,
                               np, C, (int64_t)cs.total_pts, K, ctx->split_maps.as<SegMap>() + (size_t)p0 * cs.total_segs,
                               cs.total_segs, ctx->split_first.as<double>() + (size_t)p0 * C, cs.seg_sums.as<double>(),
                               ref_pose, hsum + (size_t)p0 * C, hmax + (size_t)p0 * C);
        };
        if (job.n_first) {
            // Two parts.  Both parts' chains follow each other on the context's stream; the first part's
            // finishing pass (a wave per chain: a few hundred waves, bound by their own latency) runs beside
            // the second part's chains on a stream of its own, and the second part's beside whatever the
            // caller enqueues on the context's stream meanwhile.  The running sums for the next request's
            // guesses are left by the second part's last pose, after every chain of this request has read
            // the old ones.
            const int nA = job.n_first, nB = nb - nA;
            const int KA = split_parts(ctx, (int64_t)nA * C), KB = split_parts(ctx, (int64_t)nB * C);
            chains_of(0, nA, KA);
            CCAL_HIP(ctx, hipEventRecord(ctx->job_ev[0], S));
            chains_of(nA, nB, KB);
            CCAL_HIP(ctx, hipEventRecord(ctx->job_ev[1], S));
            if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->exact_prof_ev[1], S));
            CCAL_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream[0], ctx->job_ev[0], 0));
            finish_of(ctx->aux_stream[0], 0, nA, KA, -1);
            CCAL_HIP(ctx, hipEventRecord(ctx->job_ev[2], ctx->aux_stream[0]));
            CCAL_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream[1], ctx->job_ev[1], 0));
            finish_of(ctx->aux_stream[1], nA, nB, KB, nB - 1);
            if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->exact_prof_ev[2], ctx->aux_stream[1]));
            CCAL_HIP(ctx, hipEventRecord(ctx->job_ev[3], ctx->aux_stream[1]));
        } else {
            const int K = split_parts(ctx, chains);
            chains_of(0, nb, K);
            if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->exact_prof_ev[1], S));
            finish_of(S, 0, nb, K, nb - 1);
            if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->exact_prof_ev[2], S));
            CCAL_HIP(ctx, hipEventRecord(ctx->job_ev[3], S));
        }
    } else {
        launch_exact_chains(ctx, cs, nb, want_max, ExactTables{mats, planes, hsum, hmax});
        if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->exact_prof_ev[2], S));
        CCAL_HIP(ctx, hipEventRecord(ctx->job_ev[3], S));
    }
    CCAL_HIP(ctx, hipGetLastError());
    job.active = true;
    job.host_ms = prof ? host_ms() - job.host0 : 0.0;
    job.wait_ms = 0.0;
    return CCAL_OK;
}

// :309 and :312 for poses [p0, p1) of the request in flight
static void exact_collect(ccal_ctx* ctx, int p0, int p1, double* cost, double* per_cloud_mean, double* per_cloud_max)
{
    const ccal_ctx::ExactJob& job = ctx->exact_job;
    const CloudSet& cs = ctx->sets[job.which];
    const int C = cs.n_clouds;
    const double* hsum = ctx->exact_out.as<double>();
    const double* hmax = hsum + (size_t)job.nb * C;
    for (int k = p0; k < p1; ++k) {
        double error = 0.0;
        for (int j = 0; j < C; ++j) {
            const double mean = hsum[(size_t)k * C + j] / (double)cs.n_pts[j];  // :309
            error += mean;
            if (per_cloud_mean) per_cloud_mean[(size_t)(k - p0) * C + j] = mean;
            if (per_cloud_max) per_cloud_max[(size_t)(k - p0) * C + j] = hmax[(size_t)k * C + j];
        }
        cost[k - p0] = error / (double)C;  // :312
    }
}

int ccal::exact_wait_first(ccal_ctx* ctx, double* cost)
{
    ccal_ctx::ExactJob& job = ctx->exact_job;
    if (!job.active || !job.n_first || !cost) return fail(ctx, CCAL_ERR_STATE, "exact_wait_first: no two-part request in flight");
    const double t0 = ctx->profiling ? host_ms() : 0.0;
    if (job.watch_words) {
        // the first part's words of both arrays (sums, then the counts of segments summed twice)
        const size_t n_all = (size_t)job.nb * ctx->sets[job.which].n_clouds, n_a = (size_t)job.n_first * ctx->sets[job.which].n_clouds;
        CCAL_HIP(ctx, wait_for_words(ctx->exact_out.as<double>(), n_a, ctx->job_ev[2]));
        CCAL_HIP(ctx, wait_for_words(ctx->exact_out.as<double>() + n_all, n_a, ctx->job_ev[2]));
    } else {
        CCAL_HIP(ctx, hipEventSynchronize(ctx->job_ev[2]));
    }
    const double t1 = ctx->profiling ? host_ms() : 0.0;
    exact_collect(ctx, 0, job.n_first, cost, nullptr, nullptr);
    if (ctx->profiling) {
        job.wait_ms += t1 - t0;
        job.host_ms += host_ms() - t1;
    }
    return CCAL_OK;
}

int ccal::exact_wait_rest(ccal_ctx* ctx, double* cost, double* per_cloud_mean, double* per_cloud_max)
{
    ccal_ctx::ExactJob& job = ctx->exact_job;
    if (!job.active || !cost) return fail(ctx, CCAL_ERR_STATE, "exact_wait_rest: no request in flight");
    const bool prof = ctx->profiling;
    const double t0 = prof ? host_ms() : 0.0;
    hipError_t e = job.watch_words
                       ? wait_for_words(ctx->exact_out.as<double>(), 2 * (size_t)job.nb * ctx->sets[job.which].n_clouds, ctx->job_ev[3])
                       : hipEventSynchronize(ctx->job_ev[3]);
    if (e == hipSuccess && job.n_first && !job.watch_words) e = hipEventSynchronize(ctx->job_ev[2]);
    job.active = false;
    CCAL_HIP(ctx, e);
    const double t1 = prof ? host_ms() : 0.0;
    CloudSet& cs = ctx->sets[job.which];
    const int C = cs.n_clouds;
    if (prof) {
        float ms = 0.f, ms2 = 0.f;
        const double pp = (double)job.nb * (double)cs.total_pts;
        if (job.split) {
            if (hipEventElapsedTime(&ms, ctx->exact_prof_ev[0], ctx->exact_prof_ev[1]) == hipSuccess &&
                hipEventElapsedTime(&ms2, ctx->exact_prof_ev[1], ctx->exact_prof_ev[2]) == hipSuccess) {
                ctx->prof.split_chain_ms += ms;
                ctx->prof.split_finish_ms += ms2;     // of a two-part request: the part not hidden behind chains
                ctx->prof.split_pose_points += pp;
                ++ctx->prof.split_requests;
            }
        } else if (hipEventElapsedTime(&ms, ctx->exact_prof_ev[0], ctx->exact_prof_ev[2]) == hipSuccess) {
            ctx->prof.whole_chain_ms += ms;
            ctx->prof.whole_pose_points += pp;
            ++ctx->prof.whole_requests;
        }
    }
    cs.seg_sums_valid = true;
    {
        // the request's last pose left its running sums behind (both kinds of kernel do that); its totals, for
        // callers that want to say how far the next request's chains are from them
        const double* hs = ctx->exact_out.as<double>() + (size_t)(job.nb - 1) * C;
        cs.seg_ref_totals.assign(hs, hs + C);
    }
    if (job.split) {
        const double* hmax = ctx->exact_out.as<double>() + (size_t)job.nb * C;
        for (size_t i = 0; i < (size_t)job.nb * C; ++i) ctx->split_resummed += (int64_t)hmax[i];
        ctx->split_segments += (int64_t)job.nb * (cs.total_segs - C);
    }
    exact_collect(ctx, job.n_first, job.nb, cost, per_cloud_mean, per_cloud_max);
    if (prof) {
        ctx->prof.exact_host_ms += job.host_ms + (host_ms() - t1);
        ctx->prof.exact_wait_ms += job.wait_ms + (t1 - t0);
    }
    return CCAL_OK;
}

int ccal::eval_exact(ccal_ctx* ctx, int which, int n_poses, const double* poses6,
                     const double* planes4, double* cost, double* per_cloud_mean,
                     double* per_cloud_max, bool allow_split, const double* expected_sums)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (n_poses < 0 || (n_poses > 0 && (!poses6 || !cost)))
        return fail(ctx, CCAL_ERR_ARG, "eval: null argument");
    if (!planes4 && !ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "eval: plane not set");
    const int C = ctx->sets[which].n_clouds;
    // one workgroup per (cloud, pose) chain (or several, see exact_begin); the batch bound only keeps the
    // grid and the constant tables modest
    const int max_batch = 4096;
    for (int k0 = 0; k0 < n_poses; k0 += max_batch) {
        const int nb = std::min(max_batch, n_poses - k0);
        if (int rc = exact_begin(ctx, which, nb, poses6 + 6 * (size_t)k0, planes4 ? planes4 + 4 * (size_t)k0 : nullptr, 0,
                                 per_cloud_max != nullptr, allow_split, expected_sums ? expected_sums + (size_t)k0 * C : nullptr))
            return rc;
        if (int rc = exact_wait_rest(ctx, cost + k0, per_cloud_mean ? per_cloud_mean + (size_t)k0 * C : nullptr,
                                     per_cloud_max ? per_cloud_max + (size_t)k0 * C : nullptr))
            return rc;
    }
    return CCAL_OK;
}

extern "C" int ccal_eval(ccal_ctx* ctx, int which, int n_poses, const double* poses6,
                         const double* planes4, double* cost, double* per_cloud_mean,
                         double* per_cloud_max)
{
    // Callers that drive their own optimiser through this entry point (the reference's functor calls
    // getError pose by pose, :419-426) send runs of nearby poses, which is what the split evaluation
    // wants; a caller that jumps around makes it sum most segments twice.  So: split, look at how
    // much was summed twice, and after a bad call stay with whole chains for a while (1, 2, 4, ...
    // calls) before trying again.
    if (!ctx) return CCAL_ERR_ARG;
    bool allow = ctx->chain_split && !per_cloud_max;
    if (allow && ctx->split_skip > 0) {
        --ctx->split_skip;
        allow = false;
    }
    const int64_t resummed0 = ctx->split_resummed, segments0 = ctx->split_segments;
    const int rc = ccal::eval_exact(ctx, which, n_poses, poses6, planes4, cost, per_cloud_mean, per_cloud_max, allow);
    const int64_t segments = ctx->split_segments - segments0;
    if (rc == CCAL_OK && segments > 0) {
        if ((ctx->split_resummed - resummed0) * 10 > segments) {
            ctx->split_skip = ctx->split_penalty;
            ctx->split_penalty = std::min(256, 2 * ctx->split_penalty);
        } else {
            ctx->split_penalty = 1;
        }
    }
    return rc;
}

// Error budget of the screening pass against the exact objective, per cloud and pose, with
// rho_a = MEAN of |p_a - c_a| over the cloud's points along its own axis a (the budgets below are per point
// and linear in the |p_a - c_a|; the objective is a mean over the points), c the cloud's centre, R the 3x3
// part of M and w4 = the folded constant (the signed distance of the centre):
//   exact path:  rounds every transformed coordinate q_r to float (src/helpers.hpp:151), and the distance weighs
//                coordinate r with |n_r|/|n|: the residual moves by at most
//                2^-24 sum_r |n_r|/|n| |q_r| <= 2^-24 sum_r |n_r|/|n| (|(M (c,1))_r| + sum_a |R_ra| |p_a - c_a|)
//   fast path:   w and p - c rounded to float (one rounding each), then three fused multiply-adds (the
//                constant is the innermost addend), each on a magnitude of at most
//                sum_a |w_a| |p_a - c_a| + |w4|:  5 * 2^-24 (sum_a |w_a| rho_a + |w4|)
//   sums:        8 + 6 FP32 adds per partial (eight points per lane, six levels of the wave reduction):
//                kScreenLaneAdds * 2^-24 of the cloud's mean residual
// The FP64 arithmetic on either side is below 1e-12 of these magnitudes.  kScreenSlack rounds all of it up.

int ccal::screen_begin(ccal_ctx* ctx, int which, int nb, const double* poses6, const double* planes4)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (nb <= 0 || nb > 1024 || !poses6) return fail(ctx, CCAL_ERR_ARG, "screen_begin: 1 to 1024 poses");
    if (ctx->screen_job.active) return fail(ctx, CCAL_ERR_STATE, "screen_begin: a request is in flight");
    if (!planes4 && !ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "eval_screen: plane not set");
    if (int rc = bind(ctx)) return rc;
    if (int rc = ensure_job_objects(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    const int C = cs.n_clouds;
    ccal_ctx::ScreenJob& job = ctx->screen_job;
    job.host0 = ctx->profiling ? host_ms() : 0.0;
    // pinned: one record per pose in, per-(pose, cloud) sums and budgets out; the device builds the
    // four-vectors and the budgets from the records (screen_prep_kernel), the host only takes the
    // sines and cosines (libm, as for every pose matrix of the library)
    CCAL_HIP(ctx, ctx->screen_in.ensure((size_t)nb * kScreenRec * sizeof(double)));
    CCAL_HIP(ctx, ctx->screen_out.ensure(2 * (size_t)nb * C * sizeof(double)));
    CCAL_HIP(ctx, ctx->screen_partial.ensure((size_t)cs.n_blocks * kScreenParts * nb * sizeof(double)));
    CCAL_HIP(ctx, ctx->screen_w.ensure((size_t)nb * C * 4 * sizeof(float)));
    double* recs = ctx->screen_in.as<double>();
    double* sums = ctx->screen_out.as<double>();
    double* budget = sums + (size_t)nb * C;
    for (int k = 0; k < nb; ++k) {
        const double* pose = poses6 + 6 * (size_t)k;
        const SinCos3 sc = sincos_of_pose(pose);
        double* r = recs + (size_t)k * kScreenRec;
        for (int e = 0; e < 3; ++e) { r[e] = pose[e]; r[3 + e] = sc.s[e]; r[6 + e] = sc.c[e]; }
        for (int e = 0; e < 4; ++e) r[9 + e] = planes4 ? planes4[4 * (size_t)k + e] : (double)ctx->plane[e];
        r[13] = 1.0 / std::sqrt(r[9] * r[9] + r[10] * r[10] + r[11] * r[11]);
        r[14] = r[15] = 0.0;
    }
    job.watch_words = !ctx->profiling;
    if (job.watch_words) mark_unwritten(sums, 2 * (size_t)nb * C);
    hipStream_t S = ctx->stream;
    hipLaunchKernelGGL(screen_prep_kernel, dim3((nb * C + 63) / 64), dim3(64), 0, S, recs,
                       cs.base_dev.as<double>(), cs.centres_dev.as<double>(), cs.bound_dev.as<double>(), nb, C,
                       ctx->screen_w.as<f32x4>(), budget);
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->screen_prof_ev[0], S));
    hipLaunchKernelGGL(screen_cost_kernel, dim3(cs.n_blocks * kScreenParts), dim3(256), 0, S, cs.groups.as<f32x4>(),
                       cs.group_off_dev.as<int>(), cs.cloud_pt_off_dev.as<int64_t>(), cs.blocks.as<BlockDesc>(),
                       ctx->screen_w.as<f32x4>(), nb, C, ctx->screen_partial.as<double>());
    hipLaunchKernelGGL(screen_reduce_kernel, dim3((nb * C + 3) / 4), dim3(256), 0, S,
                       ctx->screen_partial.as<double>(), cs.cloud_block_off_dev.as<int>(), nb, C, sums);
    CCAL_HIP(ctx, hipGetLastError());
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->screen_prof_ev[1], S));
    CCAL_HIP(ctx, hipEventRecord(ctx->screen_ev, S));
    job.which = which;
    job.nb = nb;
    job.active = true;
    job.host_ms = ctx->profiling ? host_ms() - job.host0 : 0.0;
    return CCAL_OK;
}

int ccal::screen_wait(ccal_ctx* ctx, double* cost, double* bound, double* per_cloud_sums)
{
    ccal_ctx::ScreenJob& job = ctx->screen_job;
    if (!job.active || !cost || !bound) return fail(ctx, CCAL_ERR_STATE, "screen_wait: no request in flight");
    const double t0 = ctx->profiling ? host_ms() : 0.0;
    const hipError_t e = job.watch_words
                             ? wait_for_words(ctx->screen_out.as<double>(), 2 * (size_t)job.nb * ctx->sets[job.which].n_clouds, ctx->screen_ev)
                             : hipEventSynchronize(ctx->screen_ev);
    job.active = false;
    CCAL_HIP(ctx, e);
    const double t1 = ctx->profiling ? host_ms() : 0.0;
    const CloudSet& cs = ctx->sets[job.which];
    const int C = cs.n_clouds, nb = job.nb;
    const double u24 = 0x1p-24;
    if (ctx->profiling) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->screen_prof_ev[0], ctx->screen_prof_ev[1]) == hipSuccess) {
            ctx->prof.screen_ms += ms;
            ctx->prof.screen_pose_points += (double)nb * (double)cs.total_pts;
            ++ctx->prof.screen_requests;
        }
    }
    const double* sums = ctx->screen_out.as<double>();
    const double* budget = sums + (size_t)nb * C;
    for (int k = 0; k < nb; ++k) {
        double f = 0.0, b = 0.0;
        for (int j = 0; j < C; ++j) {
            const double mean = sums[(size_t)k * C + j] / (double)cs.n_pts[j];
            f += mean;
            b += budget[(size_t)k * C + j] + (double)kScreenLaneAdds * u24 * mean;
        }
        if (per_cloud_sums) std::memcpy(per_cloud_sums + (size_t)k * C, sums + (size_t)k * C, (size_t)C * sizeof(double));
        cost[k] = f / (double)C;
        bound[k] = std::isfinite(b) ? kScreenSlack * b / (double)C : std::numeric_limits<double>::infinity();
    }
    if (ctx->profiling) {
        ctx->prof.screen_host_ms += job.host_ms + (host_ms() - t1);
        ctx->prof.screen_wait_ms += t1 - t0;
    }
    return CCAL_OK;
}

int ccal::eval_screen(ccal_ctx* ctx, int which, int n_poses, const double* poses6, const double* planes4,
                      double* cost, double* bound, double* per_cloud_sums)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (n_poses < 0 || (n_poses > 0 && (!poses6 || !cost || !bound)))
        return fail(ctx, CCAL_ERR_ARG, "eval_screen: null argument");
    if (!planes4 && !ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "eval_screen: plane not set");
    const int max_batch = 1024;
    for (int k0 = 0; k0 < n_poses; k0 += max_batch) {
        const int nb = std::min(max_batch, n_poses - k0);
        if (int rc = screen_begin(ctx, which, nb, poses6 + 6 * (size_t)k0, planes4 ? planes4 + 4 * (size_t)k0 : nullptr)) return rc;
        if (int rc = screen_wait(ctx, cost + k0, bound + k0,
                                 per_cloud_sums ? per_cloud_sums + (size_t)k0 * ctx->sets[which].n_clouds : nullptr))
            return rc;
    }
    return CCAL_OK;
}

extern "C" int ccal_eval_bounded(ccal_ctx* ctx, int which, int n_poses, const double* poses6,
                                 const double* planes4, double* cost, double* bound)
{
    return ccal::eval_screen(ctx, which, n_poses, poses6, planes4, cost, bound);
}

extern "C" int ccal_set_profiling(ccal_ctx* ctx, int on)
{
    if (!ctx) return CCAL_ERR_ARG;
    if (on) {
        if (int rc = bind(ctx)) return rc;
        for (hipEvent_t& e : ctx->prof_ev)
            if (!e) CCAL_HIP(ctx, hipEventCreate(&e));
    }
    ctx->profiling = on != 0;
    return CCAL_OK;
}

extern "C" int ccal_get_profile(ccal_ctx* ctx, ccal_profile* out, int reset)
{
    if (!ctx || !out) return CCAL_ERR_ARG;
    *out = ctx->prof;
    if (reset) ctx->prof = ccal_profile{};
    return CCAL_OK;
}

extern "C" int ccal_set_chain_split(ccal_ctx* ctx, int on)
{
    if (!ctx) return CCAL_ERR_ARG;
    ctx->chain_split = on != 0;
    return CCAL_OK;
}

extern "C" int ccal_set_gd_pipelining(ccal_ctx* ctx, int on)
{
    if (!ctx) return CCAL_ERR_ARG;
    ctx->gd_pipelining = on != 0;
    return CCAL_OK;
}

extern "C" int ccal_set_gd_screening(ccal_ctx* ctx, int on)
{
    if (!ctx) return CCAL_ERR_ARG;
    ctx->gd_screening = on != 0;
    return CCAL_OK;
}

extern "C" int ccal_point_errors(ccal_ctx* ctx, int which, const double pose[6], const double* plane4,
                                 int arithmetic, double* dist_out, unsigned char* class_out)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (!pose || (!dist_out && !class_out)) return fail(ctx, CCAL_ERR_ARG, "point_errors: null argument");
    if (arithmetic != CCAL_POINT_OPTIMISER && arithmetic != CCAL_POINT_VIEWER)
        return fail(ctx, CCAL_ERR_ARG, "point_errors: unknown arithmetic selector");
    if (!plane4 && !ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "point_errors: plane not set");
    if (int rc = bind(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    std::vector<double> mats((size_t)cs.n_clouds * 12);
    pose_matrices(cs, pose, mats.data());
    double pl[kPlaneStride] = {0};
    for (int e = 0; e < 4; ++e) pl[e] = plane4 ? plane4[e] : (double)ctx->plane[e];
    pl[4] = std::sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
    CCAL_HIP(ctx, ctx->mats_buf.ensure(mats.size() * sizeof(double)));
    CCAL_HIP(ctx, ctx->planes_buf.ensure(sizeof pl));
    CCAL_HIP(ctx, ctx->dist_buf.ensure(cs.total_pts * sizeof(double)));
    CCAL_HIP(ctx, ctx->list_buf.ensure(cs.total_pts));
    CCAL_HIP(ctx, hipMemcpyAsync(ctx->mats_buf.p, mats.data(), mats.size() * sizeof(double),
                                 hipMemcpyHostToDevice, ctx->stream));
    CCAL_HIP(ctx, hipMemcpyAsync(ctx->planes_buf.p, pl, sizeof pl, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[0], ctx->stream));
    hipLaunchKernelGGL(point_error_kernel, dim3(cs.n_blocks), dim3(256), 0, ctx->stream,
                       cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.blocks.as<BlockDesc>(), ctx->mats_buf.as<double>(),
                       ctx->planes_buf.as<double>(), arithmetic == CCAL_POINT_VIEWER ? 1 : 0,
                       ctx->dist_buf.as<double>(), ctx->list_buf.as<unsigned char>());
    CCAL_HIP(ctx, hipGetLastError());
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[1], ctx->stream));
    if (dist_out) CCAL_HIP(ctx, download_array(ctx, dist_out, ctx->dist_buf.p, cs.total_pts * sizeof(double)));
    if (class_out) CCAL_HIP(ctx, download_array(ctx, class_out, ctx->list_buf.p, cs.total_pts));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->prof_ev[0], ctx->prof_ev[1]) == hipSuccess) {
            ctx->prof.point_error_ms += ms;
            ctx->prof.point_error_points += (double)cs.total_pts;
        }
    }
    return CCAL_OK;
}

// The accumulation primitive of the exact path on a caller-supplied array (see ccal.h).
extern "C" int ccal_ordered_sum(ccal_ctx* ctx, const double* values, size_t n, double* out)
{
    if (!ctx) return CCAL_ERR_ARG;
    if (!out || (n && !values)) return fail(ctx, CCAL_ERR_ARG, "ordered_sum: null argument");
    if (int rc = bind(ctx)) return rc;
    CCAL_HIP(ctx, ctx->dist_buf.ensure(std::max<size_t>(n, 1) * sizeof(double)));
    CCAL_HIP(ctx, ctx->osum_buf.ensure(sizeof(double)));
    CCAL_HIP(ctx, ctx->omax_buf.ensure(sizeof(double)));
    if (n)
        CCAL_HIP(ctx, hipMemcpyAsync(ctx->dist_buf.p, values, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(64 * kChainWaves), 0, ctx->stream, ctx->dist_buf.as<double>(),
                       (int64_t)n, ctx->osum_buf.as<double>(), ctx->omax_buf.as<double>());
    CCAL_HIP(ctx, hipGetLastError());
    CCAL_HIP(ctx, hipMemcpyAsync(out, ctx->osum_buf.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CCAL_OK;
}

