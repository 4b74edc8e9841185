// voxel_impl.hpp -- device voxel-grid downsample (the body of api_voxel.hip).
//
// Stands in for pcl::VoxelGrid<PointXYZ> at src/camera_calibration_optimization.cpp:196-205
// (leaf 0.02f) with the same definition as the host filter (csrc/host/cloud_io.cpp): PCL's
// index arithmetic in float, voxels emitted in ascending linear index, the points of a voxel
// accumulated in input order in float.  All clouds of the set go through ONE pipeline:
// per-cloud bounds (one launch) -> keys (cloud number above the cell index, so that one sort orders
// the whole set cloud by cloud, cell by cell) -> one stable radix sort of (key, point index) over the
// significant key bits -> run-length encode -> coordinates gathered in sorted order -> one thread per
// voxel summing its run in order.  Three host round trips in all (bounds, run count, run keys); a sort
// per cloud -- a few hundred voxels of a few thousand points each -- spent most of its time in launch
// overheads and in the merge sort the library picks for small inputs.
#pragma once

#include <rocprim/rocprim.hpp>

namespace ccal {

// grid = (blocks per cloud, clouds); out[cloud][block][6] = min xyz, max xyz over the block's share
__global__ __launch_bounds__(256) void voxel_bounds_kernel(const float* __restrict__ xs,
                                                           const float* __restrict__ ys,
                                                           const float* __restrict__ zs,
                                                           const int64_t* __restrict__ cloud_pt_off,
                                                           float* __restrict__ out)
{
    __shared__ float red[4][6];
    const int64_t begin = cloud_pt_off[blockIdx.y], end = cloud_pt_off[blockIdx.y + 1];
    float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    float mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float v[3] = {xs[i], ys[i], zs[i]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!__builtin_isfinite(v[k])) continue;
            mn[k] = v[k] < mn[k] ? v[k] : mn[k];
            mx[k] = v[k] > mx[k] ? v[k] : mx[k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            const float a = __shfl_xor(mn[k], off, 64), b = __shfl_xor(mx[k], off, 64);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
        if (lane == 0) { red[wave][k] = mn[k]; red[wave][3 + k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[0][k];
        for (int w = 1; w < 4; ++w) v = k < 3 ? (red[w][k] < v ? red[w][k] : v) : (red[w][k] > v ? red[w][k] : v);
        out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 6 + k] = v;
    }
}

struct VoxelDims {
    float inv;
    float minb[3];
    long long div0, div01;
    unsigned long long nonfinite_key;   // one past the last cell: sorts last within the cloud
};

// One thread per point; the table of 4096-point blocks says which cloud a point belongs to.
// key = cloud << cell_bits | cell; idx = the point's index in the whole set.
// KeyT: 32 bits wide when cloud number and cell index fit (the usual case: the sort then moves 8 bytes
// per point and pass instead of 12), else 64.
template <typename KeyT>
__global__ __launch_bounds__(256) void voxel_keys_kernel(const float* __restrict__ xs,
                                                         const float* __restrict__ ys,
                                                         const float* __restrict__ zs,
                                                         const BlockDesc* __restrict__ blocks,
                                                         const VoxelDims* __restrict__ dims, unsigned cell_bits,
                                                         KeyT* __restrict__ keys,
                                                         unsigned int* __restrict__ idx)
{
    const BlockDesc bd = blocks[blockIdx.x];
    const VoxelDims d = dims[bd.cloud];
    for (int64_t i = bd.begin + threadIdx.x; i < bd.end; i += blockDim.x) {
        const float x = xs[i], y = ys[i], z = zs[i];
        unsigned long long key = d.nonfinite_key;   // non-finite points sort last in their cloud and are dropped
        if (__builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z)) {
            const long long i0 = (long long)(floorf(x * d.inv) - d.minb[0]);
            const long long i1 = (long long)(floorf(y * d.inv) - d.minb[1]);
            const long long i2 = (long long)(floorf(z * d.inv) - d.minb[2]);
            key = (unsigned long long)(i0 + i1 * d.div0 + i2 * d.div01);
        }
        keys[i] = (KeyT)(((unsigned long long)bd.cloud << cell_bits) | key);
        idx[i] = (unsigned int)i;
    }
}

// Coordinates in sorted order (cloud by cloud, voxel by voxel, input order inside a voxel: the sort is
// stable), so that the per-voxel sums below read contiguous runs.
__global__ __launch_bounds__(256) void voxel_gather_kernel(const float* __restrict__ xs,
                                                           const float* __restrict__ ys,
                                                           const float* __restrict__ zs,
                                                           const unsigned int* __restrict__ sorted_idx, int64_t n,
                                                           float* __restrict__ gx, float* __restrict__ gy,
                                                           float* __restrict__ gz)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t i = sorted_idx[t];
    gx[t] = xs[i];
    gy[t] = ys[i];
    gz[t] = zs[i];
}

// One thread per run: float sums of its points in order (one add at a time, as a serial filter
// does); eight elements are in flight per trip so the loads do not wait for the adds.
__global__ __launch_bounds__(64) void voxel_centroid_kernel(
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const unsigned int* __restrict__ counts, const unsigned int* __restrict__ offsets, unsigned int n_runs,
    float* __restrict__ out /* AoS xyz */)
{
    const unsigned int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_runs) return;
    const size_t o = offsets[v];
    const unsigned int c = counts[v];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    unsigned int t = 0;
    for (; t + 8 <= c; t += 8) {
        float x[8], y[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = gx[o + t + u]; y[u] = gy[o + t + u]; z[u] = gz[o + t + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { sx += x[u]; sy += y[u]; sz += z[u]; }
    }
    for (; t < c; ++t) { sx += gx[o + t]; sy += gy[o + t]; sz += gz[o + t]; }
    const float cnt = (float)c;
    out[3 * (size_t)v + 0] = sx / cnt;
    out[3 * (size_t)v + 1] = sy / cnt;
    out[3 * (size_t)v + 2] = sz / cnt;
}

// The same sums for long runs: one wave per run.  The wave reads the run 512 elements at a time with
// coalesced loads (the next piece is in flight while the present one is added) and hands it through LDS
// to three of its lanes, one per coordinate, which add their elements one at a time in order.
constexpr int kCentroidPiece = 512;     // elements per piece and coordinate (8 per lane)
constexpr int kCentroidWaves = 4;       // waves per workgroup (each works on runs of its own)

__global__ __launch_bounds__(64 * kCentroidWaves) void voxel_centroid_wave_kernel(
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const unsigned int* __restrict__ counts, const unsigned int* __restrict__ offsets, unsigned int n_runs,
    float* __restrict__ out /* AoS xyz */)
{
    __shared__ __attribute__((aligned(16))) float buf[kCentroidWaves][3][kCentroidPiece];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int n_waves = gridDim.x * kCentroidWaves;
    constexpr int kPerLane = kCentroidPiece / 64;
    const float* g[3] = {gx, gy, gz};
    for (unsigned int v = blockIdx.x * kCentroidWaves + wave; v < n_runs; v += n_waves) {
        const size_t o = offsets[v];
        const unsigned int c = counts[v];
        float s = 0.f;                               // lane k < 3: the sum of coordinate k
        float r[3][kPerLane];
        auto fetch = [&](unsigned int t0) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u = 0; u < kPerLane; ++u) {
                    const unsigned int i = t0 + (unsigned)(u * 64 + lane);
                    r[k][u] = i < c ? g[k][o + i] : 0.f;
                }
        };
        fetch(0);
        for (unsigned int t0 = 0; t0 < c; t0 += kCentroidPiece) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u = 0; u < kPerLane; ++u) buf[wave][k][u * 64 + lane] = r[k][u];
            if (t0 + kCentroidPiece < c) fetch(t0 + kCentroidPiece);
            // LDS operations of one wave complete in order; the fences keep the compiler from moving
            // the reads of other lanes' elements above the writes (and the next writes above the reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 3) {
                const unsigned int n = c - t0 < (unsigned)kCentroidPiece ? c - t0 : (unsigned)kCentroidPiece;
                const float* b = buf[wave][lane];
                unsigned int i = 0;
                for (; i + 4 <= n; i += 4) {
                    const f32x4 q = *(const f32x4*)(b + i);
                    s += q[0]; s += q[1]; s += q[2]; s += q[3];
                }
                for (; i < n; ++i) s += b[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane < 3) out[3 * (size_t)v + lane] = s / (float)c;
    }
}

} // namespace ccal

extern "C" int ccal_downsample_raw(ccal_ctx* ctx, float leaf, size_t* n_pts_out)
{
    if (int rc = check_set(ctx, CCAL_RAW)) return rc;
    if (!(leaf > 0.f)) return fail(ctx, CCAL_ERR_ARG, "downsample_raw: leaf must be positive");
    if (int rc = bind(ctx)) return rc;
    const CloudSet& raw = ctx->sets[CCAL_RAW];
    const int C = raw.n_clouds;
    const size_t N = raw.total_pts;
    if (N >= ((size_t)1 << 32)) return fail(ctx, CCAL_ERR_CAPACITY, "downsample_raw: more than 2^32 points");
    hipStream_t st = ctx->stream;
    const float* xs = raw.xs.as<float>();
    const float* ys = raw.ys.as<float>();
    const float* zs = raw.zs.as<float>();

    // ---- bounds of every cloud ---------------------------------------------------------------
    const int nb = 64;
    CCAL_HIP(ctx, ctx->psum_buf.ensure((size_t)C * nb * 6 * sizeof(float)));
    hipLaunchKernelGGL(voxel_bounds_kernel, dim3(nb, C), dim3(256), 0, st, xs, ys, zs,
                       raw.cloud_pt_off_dev.as<int64_t>(), ctx->psum_buf.as<float>());
    std::vector<float> hb((size_t)C * nb * 6);
    CCAL_HIP(ctx, hipMemcpyAsync(hb.data(), ctx->psum_buf.p, hb.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    CCAL_HIP(ctx, hipStreamSynchronize(st));
    std::vector<VoxelDims> dims(C);
    unsigned cell_bits = 1;
    for (int j = 0; j < C; ++j) {
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], hb[((size_t)j * nb + b) * 6 + k]);
                mx[k] = std::max(mx[k], hb[((size_t)j * nb + b) * 6 + 3 + k]);
            }
        if (!(mn[0] <= mx[0])) return fail(ctx, CCAL_ERR_ARG, "downsample_raw: a cloud has no finite point");
        VoxelDims& d = dims[j];
        d.inv = 1.0f / leaf;
        long long minb[3], div[3];
        for (int k = 0; k < 3; ++k) {
            minb[k] = (long long)std::floor(mn[k] * d.inv);
            div[k] = (long long)std::floor(mx[k] * d.inv) - minb[k] + 1;
            d.minb[k] = (float)minb[k];
        }
        d.div0 = div[0];
        d.div01 = div[0] * div[1];
        const double cells = (double)div[0] * (double)div[1] * (double)div[2];
        if (!(cells < 4.0e18)) return fail(ctx, CCAL_ERR_ARG, "voxel grid: leaf too small for the cloud extent");   // keys stay below 2^62
        d.nonfinite_key = (unsigned long long)div[0] * (unsigned long long)div[1] * (unsigned long long)div[2];
        while (cell_bits < 64 && (1ull << cell_bits) <= d.nonfinite_key) ++cell_bits;
    }
    unsigned cloud_bits = 0;
    while ((1u << cloud_bits) < (unsigned)C) ++cloud_bits;
    if (cell_bits + cloud_bits > 64)
        return fail(ctx, CCAL_ERR_ARG, "voxel grid: leaf too small for the cloud extent");

    // ---- keys, one stable sort over the significant bits, run-length encode ------------------------
    // (scratch lives in the context and only grows)
    DevBuf& keys = ctx->vox[0];
    DevBuf& keys2 = ctx->vox[1];
    DevBuf& idx = ctx->vox[2];
    DevBuf& idx2 = ctx->vox[3];
    DevBuf& uniq = ctx->vox[4];
    DevBuf& counts = ctx->vox[5];
    DevBuf& offsets = ctx->vox[6];
    DevBuf& nruns = ctx->vox[7];
    DevBuf& tmp = ctx->vox[8];
    DevBuf& cent = ctx->vox[9];
    DevBuf& gathered = ctx->vox[10];      // x | y | z of the whole set in sorted order
    DevBuf& dims_dev = ctx->vox[11];
    CCAL_HIP(ctx, keys.ensure(N * 8)); CCAL_HIP(ctx, keys2.ensure(N * 8));
    CCAL_HIP(ctx, idx.ensure(N * 4)); CCAL_HIP(ctx, idx2.ensure(N * 4));
    CCAL_HIP(ctx, nruns.ensure(4));
    CCAL_HIP(ctx, gathered.ensure(3 * N * sizeof(float)));
    CCAL_HIP(ctx, dims_dev.ensure((size_t)C * sizeof(VoxelDims)));
    CCAL_HIP(ctx, hipMemcpyAsync(dims_dev.p, dims.data(), (size_t)C * sizeof(VoxelDims), hipMemcpyHostToDevice, st));
    auto sort_and_encode = [&](auto key_tag) -> int {
        using KeyT = decltype(key_tag);
        hipLaunchKernelGGL(voxel_keys_kernel<KeyT>, dim3(raw.n_blocks), dim3(256), 0, st, xs, ys, zs, raw.blocks.as<BlockDesc>(),
                           dims_dev.as<VoxelDims>(), cell_bits, keys.as<KeyT>(), idx.as<unsigned int>());
        size_t tmp_bytes = 0;
        CCAL_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.as<KeyT>(), keys2.as<KeyT>(), idx.as<unsigned int>(),
                                                idx2.as<unsigned int>(), N, 0, cell_bits + cloud_bits, st));
        CCAL_HIP(ctx, tmp.ensure(tmp_bytes));
        CCAL_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.as<KeyT>(), keys2.as<KeyT>(), idx.as<unsigned int>(),
                                                idx2.as<unsigned int>(), N, 0, cell_bits + cloud_bits, st));
        // the sorted keys are no longer needed once encoded: the run tables reuse the unsorted key array
        // (runs <= points)
        size_t tmp2 = 0;
        CCAL_HIP(ctx, rocprim::run_length_encode(nullptr, tmp2, keys2.as<KeyT>(), (unsigned int)N, uniq.as<KeyT>(),
                                                 counts.as<unsigned int>(), nruns.as<unsigned int>(), st));
        CCAL_HIP(ctx, tmp.ensure(tmp2));
        CCAL_HIP(ctx, rocprim::run_length_encode(tmp.p, tmp2, keys2.as<KeyT>(), (unsigned int)N, uniq.as<KeyT>(),
                                                 counts.as<unsigned int>(), nruns.as<unsigned int>(), st));
        return CCAL_OK;
    };
    const bool narrow = cell_bits + cloud_bits <= 32;
    CCAL_HIP(ctx, uniq.ensure(N * 8)); CCAL_HIP(ctx, counts.ensure(N * 4)); CCAL_HIP(ctx, offsets.ensure(N * 4));
    if (int rc = narrow ? sort_and_encode((unsigned int)0) : sort_and_encode((unsigned long long)0)) return rc;
    hipLaunchKernelGGL(voxel_gather_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, xs, ys, zs,
                       idx2.as<unsigned int>(), (int64_t)N, gathered.as<float>(), gathered.as<float>() + N,
                       gathered.as<float>() + 2 * N);
    unsigned int n_runs = 0;
    CCAL_HIP(ctx, hipMemcpyAsync(&n_runs, nruns.p, 4, hipMemcpyDeviceToHost, st));
    CCAL_HIP(ctx, hipStreamSynchronize(st));
    size_t tmp3 = 0;
    CCAL_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp3, counts.as<unsigned int>(), offsets.as<unsigned int>(), 0u,
                                          n_runs, rocprim::plus<unsigned int>(), st));
    CCAL_HIP(ctx, tmp.ensure(tmp3));
    CCAL_HIP(ctx, rocprim::exclusive_scan(tmp.p, tmp3, counts.as<unsigned int>(), offsets.as<unsigned int>(), 0u,
                                          n_runs, rocprim::plus<unsigned int>(), st));
    // ---- centroids of all runs; the host drops each cloud's run of non-finite points (its last) -------
    CCAL_HIP(ctx, cent.ensure((size_t)n_runs * 3 * sizeof(float)));
    // long runs (coarse leaves): a wave per run; short ones: a lane per run
    if (n_runs > 0 && N / n_runs >= 64) {
        const unsigned int groups = std::min<unsigned int>((n_runs + kCentroidWaves - 1) / kCentroidWaves, 256u * 8u);
        hipLaunchKernelGGL(voxel_centroid_wave_kernel, dim3(groups), dim3(64 * kCentroidWaves), 0, st,
                           gathered.as<float>(), gathered.as<float>() + N, gathered.as<float>() + 2 * N,
                           counts.as<unsigned int>(), offsets.as<unsigned int>(), n_runs, cent.as<float>());
    } else {
        hipLaunchKernelGGL(voxel_centroid_kernel, dim3((n_runs + 63) / 64), dim3(64), 0, st, gathered.as<float>(),
                           gathered.as<float>() + N, gathered.as<float>() + 2 * N, counts.as<unsigned int>(),
                           offsets.as<unsigned int>(), n_runs, cent.as<float>());
    }
    CCAL_HIP(ctx, hipGetLastError());
    std::vector<float> all((size_t)n_runs * 3);
    std::vector<unsigned long long> run_key(n_runs);
    std::vector<unsigned int> run_key32(narrow ? n_runs : 0);
    CCAL_HIP(ctx, hipMemcpyAsync(all.data(), cent.p, all.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    if (narrow) CCAL_HIP(ctx, hipMemcpyAsync(run_key32.data(), uniq.p, (size_t)n_runs * 4, hipMemcpyDeviceToHost, st));
    else CCAL_HIP(ctx, hipMemcpyAsync(run_key.data(), uniq.p, (size_t)n_runs * 8, hipMemcpyDeviceToHost, st));
    CCAL_HIP(ctx, hipStreamSynchronize(st));
    if (narrow) std::copy(run_key32.begin(), run_key32.end(), run_key.begin());
    std::vector<float> down;
    down.reserve(all.size());
    std::vector<size_t> sizes(C, 0);
    const unsigned long long cell_mask = cell_bits >= 64 ? ~0ull : ((1ull << cell_bits) - 1ull);
    for (unsigned int r = 0; r < n_runs; ++r) {
        const int j = cell_bits >= 64 ? 0 : (int)(run_key[r] >> cell_bits);
        if ((run_key[r] & cell_mask) == dims[j].nonfinite_key) continue;
        down.insert(down.end(), all.begin() + 3 * (size_t)r, all.begin() + 3 * (size_t)r + 3);
        ++sizes[j];
    }
    std::vector<const float*> ptrs(C);
    size_t at = 0;
    for (int j = 0; j < C; ++j) {
        if (sizes[j] == 0) return fail(ctx, CCAL_ERR_ARG, "downsample_raw: a cloud has no finite point");
        ptrs[j] = down.data() + 3 * at;
        at += sizes[j];
        if (n_pts_out) n_pts_out[j] = sizes[j];
    }
    const std::vector<double> B = raw.base_T_flange;   // upload replaces sets[which] only, copy is defensive
    return ccal_upload_clouds(ctx, CCAL_DOWNSAMPLED, C, ptrs.data(), sizes.data(), B.data());
}

extern "C" int ccal_cloud_size(const ccal_ctx* ctx, int which, int cloud, size_t* n_out)
{
    if (!ctx || !n_out || (which != CCAL_DOWNSAMPLED && which != CCAL_RAW)) return CCAL_ERR_ARG;
    const CloudSet& cs = ctx->sets[which];
    if (!cs.ready) return CCAL_ERR_STATE;
    if (cloud < 0 || cloud >= cs.n_clouds) return CCAL_ERR_ARG;
    *n_out = cs.n_pts[cloud];
    return CCAL_OK;
}

extern "C" int ccal_download_cloud(ccal_ctx* ctx, int which, int cloud, float* xyz_out)
{
    if (int rc = check_set(ctx, which)) return rc;
    const CloudSet& cs = ctx->sets[which];
    if (cloud < 0 || cloud >= cs.n_clouds || !xyz_out) return fail(ctx, CCAL_ERR_ARG, "download_cloud: bad argument");
    if (int rc = bind(ctx)) return rc;
    size_t off = 0;
    for (int j = 0; j < cloud; ++j) off += cs.n_pts[j];
    const size_t n = cs.n_pts[cloud];
    std::vector<float> x(n), y(n), z(n);
    CCAL_HIP(ctx, hipMemcpy(x.data(), cs.xs.as<float>() + off, n * sizeof(float), hipMemcpyDeviceToHost));
    CCAL_HIP(ctx, hipMemcpy(y.data(), cs.ys.as<float>() + off, n * sizeof(float), hipMemcpyDeviceToHost));
    CCAL_HIP(ctx, hipMemcpy(z.data(), cs.zs.as<float>() + off, n * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) { xyz_out[3 * i] = x[i]; xyz_out[3 * i + 1] = y[i]; xyz_out[3 * i + 2] = z[i]; }
    return CCAL_OK;
}
