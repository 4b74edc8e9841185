// api_fit.hip -- plane fits: least squares over the moments of the transformed points (ccal_fit_plane_lsq and
// its variant that allocates nothing) and batched RANSAC (ransac_impl.hpp).
#include "ccal_internal.hpp"
#include "kernels_fit.hpp"

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

// Scratch of ccal_fit_plane_lsq_static, the one entry point that allocates nothing: the per-cloud
// matrices of the pose ([kStaticMaxClouds][12]) followed by the partial moments of the moment pass's
// workgroups ([kMomentGrid][kMoments]).  Every GPU holds its own copy of this array at its own address;
// host code only ever uses the address hipGetSymbolAddress returned for the context's device
// (ccal_ctx::static_scratch), never the array's name.  All contexts of one GPU share the copy:
// calls that overlap on one GPU race.  Nothing in it has to start at zero (a call overwrites
// every row it reads).
static_assert(kStaticScratchDoubles == (size_t)kStaticMaxClouds * 12 + (size_t)kMomentGrid * kMoments, "scratch layout");
static __device__ __align__(256) double scratch[kStaticScratchDoubles];   // 256 KiB per GPU

// The address of the CURRENT device's copy of the array (ccal_create calls this once per context,
// with the context's device current, and keeps the pointer).
hipError_t ccal::static_scratch_address(double** out)
{
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
    *out = e == hipSuccess ? static_cast<double*>(sym) : nullptr;
    return e;
}

// ---------------------------------------------------------------------------
// least-squares plane
// ---------------------------------------------------------------------------

static bool solve3(long double A[3][4], long double x[3])
{
    for (int c = 0; c < 3; ++c) {
        int piv = c;
        for (int r = c + 1; r < 3; ++r)
            if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0L) return false;
        if (piv != c)
            for (int k = 0; k < 4; ++k) std::swap(A[c][k], A[piv][k]);
        for (int r = c + 1; r < 3; ++r) {
            long double f = A[r][c] / A[c][c];
            for (int k = c; k < 4; ++k) A[r][k] -= f * A[c][k];
        }
    }
    for (int r = 2; r >= 0; --r) {
        long double s = A[r][3];
        for (int k = r + 1; k < 3; ++k) s -= A[r][k] * x[k];
        x[r] = s / A[r][r];
    }
    return true;
}

// Normal equations of [x y 1](a b c')^T = -z about the origin `org`, from the ten moments.
static int plane_from_moments(ccal_ctx* ctx, const double* m, const double org[3], float plane_out[4], const char* who)
{
    long double A[3][4] = {{m[4], m[5], m[1], -m[7]},
                           {m[5], m[6], m[2], -m[8]},
                           {m[1], m[2], m[0], -m[3]}};
    long double p[3];
    if (!solve3(A, p)) return fail(ctx, CCAL_ERR_ARG, std::string(who) + ": degenerate point set");
    const double a = (double)p[0], b = (double)p[1];
    const double c = (double)(p[2] - p[0] * org[0] - p[1] * org[1] - (long double)org[2]);
    const double len = std::sqrt(a * a + b * b + 1.0);
    plane_out[0] = (float)(a / len);
    plane_out[1] = (float)(b / len);
    plane_out[2] = (float)(1.0 / len);
    plane_out[3] = (float)(c / len);
    return CCAL_OK;
}

extern "C" int ccal_fit_plane_lsq(ccal_ctx* ctx, int which, const double pose[6],
                                  float plane_out[4])
{
    if (int rc = check_set(ctx, which)) return rc;
    if (!pose || !plane_out) return fail(ctx, CCAL_ERR_ARG, "fit_plane_lsq: null argument");
    if (int rc = bind(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    const int C = cs.n_clouds;
    std::vector<double> mats((size_t)C * 12);
    pose_matrices(cs, pose, mats.data());
    // origin for the moments: the first cloud's centre carried to the base frame
    double org[3];
    for (int r = 0; r < 3; ++r)
        org[r] = mats[r * 4 + 0] * cs.centres[0] + mats[r * 4 + 1] * cs.centres[1] +
                 mats[r * 4 + 2] * cs.centres[2] + mats[r * 4 + 3];
    CCAL_HIP(ctx, ctx->mats_buf.ensure(mats.size() * sizeof(double)));
    CCAL_HIP(ctx, ctx->psum_buf.ensure((size_t)cs.n_blocks * kMoments * sizeof(double)));
    CCAL_HIP(ctx, ctx->osum_buf.ensure(kMoments * sizeof(double)));
    CCAL_HIP(ctx, hipMemcpyAsync(ctx->mats_buf.p, mats.data(), mats.size() * sizeof(double),
                                 hipMemcpyHostToDevice, ctx->stream));
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[0], ctx->stream));
    hipLaunchKernelGGL(moments_kernel, dim3(moment_rows(cs.n_blocks)), dim3(256), 0, ctx->stream,
                       cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.blocks.as<BlockDesc>(), cs.n_blocks, ctx->mats_buf.as<double>(), org[0], org[1],
                       org[2], InlierMask{{0, 0, 0, 0, 1}, -1.0}, ctx->psum_buf.as<double>());
    if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[1], ctx->stream));
    hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                       ctx->psum_buf.as<double>(), moment_rows(cs.n_blocks), ctx->osum_buf.as<double>());
    CCAL_HIP(ctx, hipGetLastError());
    double m[kMoments];
    CCAL_HIP(ctx, hipMemcpyAsync(m, ctx->osum_buf.p, sizeof m, hipMemcpyDeviceToHost, ctx->stream));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->prof_ev[0], ctx->prof_ev[1]) == hipSuccess) {
            ctx->prof.moments_ms += ms;
            ctx->prof.moments_points += (double)cs.total_pts;
            ++ctx->prof.moments_calls;
        }
    }
    return plane_from_moments(ctx, m, org, plane_out, "fit_plane_lsq");
}

// Writes rows 0..2 of base_T_flange_j * F for every cloud from the pose's translation and the sines and
// cosines the host took (same expressions as the host's pose_matrices: bit-identical matrices).
__global__ __launch_bounds__(256) void pose_rows_kernel(const double* __restrict__ base, int n_clouds,
                                                        SinCos3 sc, double tx, double ty, double tz,
                                                        double* __restrict__ mats)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_clouds) return;
    double R[9];
    rotation_zyx(sc.s, sc.c, R);
    const double t[3] = {tx, ty, tz};
    double M[12];
    compose_rows(base + 16 * (int64_t)j, R, t, M);
    for (int e = 0; e < 12; ++e) mats[(int64_t)j * 12 + e] = M[e];
}

extern "C" int ccal_fit_plane_lsq_static(ccal_ctx* ctx, int which, const double pose[6], float plane_out[4])
{
    if (int rc = check_set(ctx, which)) return rc;
    if (!pose || !plane_out) return fail(ctx, CCAL_ERR_ARG, "fit_plane_lsq_static: null argument");
    const CloudSet& cs = ctx->sets[which];
    const int C = cs.n_clouds;
    if (C > kStaticMaxClouds)
        return fail(ctx, CCAL_ERR_CAPACITY,
                    "fit_plane_lsq_static: the set exceeds the static scratch (at most 1024 clouds); use "
                    "ccal_fit_plane_lsq");
    if (int rc = bind(ctx)) return rc;
    // origin for the moments: the first cloud's centre carried to the base frame (as ccal_fit_plane_lsq)
    const SinCos3 sc = sincos_of_pose(pose);
    double R[9], M0[12];
    rotation_zyx(sc.s, sc.c, R);
    compose_rows(cs.base_T_flange.data(), R, pose, M0);
    double org[3];
    for (int r = 0; r < 3; ++r)
        org[r] = M0[r * 4 + 0] * cs.centres[0] + M0[r * 4 + 1] * cs.centres[1] + M0[r * 4 + 2] * cs.centres[2] +
                 M0[r * 4 + 3];
    double* mats = ctx->static_scratch;                               // this device's copy, never the symbol
    double* part = ctx->static_scratch + (size_t)kStaticMaxClouds * 12;
    double* m = ctx->static_out.as<double>();                         // pinned: the last kernel writes it
    hipLaunchKernelGGL(pose_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, ctx->stream, cs.base_dev.as<double>(),
                       C, sc, pose[0], pose[1], pose[2], mats);
    hipLaunchKernelGGL(moments_kernel, dim3(moment_rows(cs.n_blocks)), dim3(256), 0, ctx->stream, cs.xs.as<float>(),
                       cs.ys.as<float>(), cs.zs.as<float>(), cs.blocks.as<BlockDesc>(), cs.n_blocks, mats, org[0], org[1], org[2],
                       InlierMask{{0, 0, 0, 0, 1}, -1.0}, part);
    hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(1024), 0, ctx->stream, part, moment_rows(cs.n_blocks), m);
    CCAL_HIP(ctx, hipGetLastError());
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return plane_from_moments(ctx, m, org, plane_out, "fit_plane_lsq_static");
}

#include "ransac_impl.hpp"
