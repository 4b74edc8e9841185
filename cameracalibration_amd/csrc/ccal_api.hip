// ccal_api.hip -- C ABI of the calibration hot path (see include/ccal.h).
//
// Host orchestration only: pose matrices and grid tables are prepared in FP64 on
// the host (sines/cosines from libm, as the reference's Eigen calls do), every
// per-point loop runs in the HIP kernels of kernels.hip.hpp.  There is no CPU
// fallback: a missing or failing device is reported, never worked around.
#include "ccal_internal.hpp"
#include "kernels.hip.hpp"

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
constexpr int kStaticMaxClouds = 1024;
constexpr size_t kStaticScratchDoubles = (size_t)kStaticMaxClouds * 12 + (size_t)kMomentGrid * kMoments;
static __device__ __align__(256) double scratch[kStaticScratchDoubles];   // 256 KiB per GPU

namespace {

int fail(ccal_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

#define CCAL_HIP(ctx, expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return fail((ctx), CCAL_ERR_DEVICE,                                          \
                        std::string(#expr) + ": " + hipGetErrorString(e__));             \
    } while (0)

// HIP events that are destroyed on every way out of a function.
struct Events {
    std::vector<hipEvent_t> e;
    hipError_t create(size_t n)
    {
        e.assign(n, nullptr);
        for (auto& ev : e) {
            hipError_t rc = hipEventCreate(&ev);
            if (rc != hipSuccess) return rc;
        }
        return hipSuccess;
    }
    hipEvent_t operator[](size_t i) const { return e[i]; }
    ~Events() { for (auto ev : e) if (ev) (void)hipEventDestroy(ev); }
    Events() = default;
    Events(const Events&) = delete;
    Events& operator=(const Events&) = delete;
};

// fn(j) for j in [0, n) on a few host threads (clouds are independent units of host work).
template <typename Fn>
void parallel_over(int n, Fn fn)
{
    const int n_threads = (int)std::min<size_t>({(size_t)n, 8, std::max(1u, std::thread::hardware_concurrency())});
    std::atomic<int> next{0};
    auto work = [&]() { for (int j = next++; j < n; j = next++) fn(j); };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

int bind(ccal_ctx* ctx)
{
    CCAL_HIP(ctx, hipSetDevice(ctx->device));
    return CCAL_OK;
}

int check_set(ccal_ctx* ctx, int which)
{
    if (!ctx) return CCAL_ERR_ARG;
    if (which != CCAL_DOWNSAMPLED && which != CCAL_RAW)
        return fail(ctx, CCAL_ERR_ARG, "cloud-set selector must be CCAL_DOWNSAMPLED or CCAL_RAW");
    if (!ctx->sets[which].ready) return fail(ctx, CCAL_ERR_STATE, "cloud set not uploaded");
    return CCAL_OK;
}

constexpr int kPointsPerBlock = 256 * 16;  // exact path: points handled by one block
constexpr int kNA = 8;                      // candidate tiles per wave on the fast path
constexpr int kCandPerBlock = 4 * kNA * 16; // 512
constexpr unsigned kBandCap = 1u << 16;     // near-minimum candidates re-scored per pass

// Rows 0..2 of base_T_flange_j * F(pose) for every cloud, [n_clouds][12].
void pose_matrices(const CloudSet& cs, const double pose[6], double* mats)
{
    SinCos3 sc = sincos_of_pose(pose);
    double R[9];
    rotation_zyx(sc.s, sc.c, R);
    for (int j = 0; j < cs.n_clouds; ++j)
        compose_rows(cs.base_T_flange.data() + 16 * j, R, pose, mats + 12 * j);
}

// Splits of the point stream so that the launch has enough blocks to fill the
// chip several times over, without cutting a cloud into slivers.
int choose_splits(const ccal_ctx* ctx, const CloudSet& cs, int64_t cand_blocks)
{
    const int64_t target = (int64_t)ctx->n_cu * 64;
    int64_t s = (target + cand_blocks - 1) / cand_blocks;
    int64_t s_max = std::max<int64_t>(1, cs.min_groups / 8);
    s = std::min<int64_t>(s, s_max);
    s = std::min<int64_t>(s, 64);
    return (int)std::max<int64_t>(1, s);
}

struct FastLaunch {
    int n_ctiles;
    int64_t n_padded;
    int64_t cand_blocks;
    int n_splits;
};

// per_block: candidates of one workgroup of the kernel the plan is for (kCandPerBlock for kernels that
// give a wave 8 candidate tiles, twice that for 16).
FastLaunch plan_fast(const ccal_ctx* ctx, const CloudSet& cs, int64_t n, int64_t per_block)
{
    FastLaunch f;
    f.cand_blocks = (n + per_block - 1) / per_block;
    f.n_padded = f.cand_blocks * per_block;
    f.n_ctiles = (int)(f.n_padded / 16);
    f.n_splits = choose_splits(ctx, cs, f.cand_blocks);
    return f;
}

// Largest candidate chunk per launch: bounded by the W buffer (16 B per
// candidate and cloud) and by the 2^31 tile index range.
int64_t fast_chunk(const CloudSet& cs)
{
    const int64_t w_budget = (int64_t)4 << 30;  // 4 GiB of W per launch
    int64_t c = w_budget / (16 * (int64_t)cs.n_clouds);
    c = std::min<int64_t>(c, (int64_t)1 << 24);
    c = std::max<int64_t>(c, 2 * kCandPerBlock);
    return (c / (2 * kCandPerBlock)) * (2 * kCandPerBlock);
}

FoldConsts fold_consts(const float plane[4])
{
    FoldConsts fc;
    fc.n[0] = plane[0]; fc.n[1] = plane[1]; fc.n[2] = plane[2];
    fc.d = plane[3];
    fc.inv_norm = 1.0 / std::sqrt(fc.n[0] * fc.n[0] + fc.n[1] * fc.n[1] + fc.n[2] * fc.n[2]);
    return fc;
}

// The points as ready-made bf16x3 operands of the wide residual kernel (64 B per point
// instead of 16): built on first use of that kernel only.
int ensure_split_groups(ccal_ctx* ctx, CloudSet& cs)
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
// lifetime
// ---------------------------------------------------------------------------

extern "C" int ccal_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* ccal_status_string(int status)
{
    switch (status) {
    case CCAL_OK: return "ok";
    case CCAL_ERR_ARG: return "invalid argument";
    case CCAL_ERR_DEVICE: return "device error";
    case CCAL_ERR_STATE: return "invalid state";
    case CCAL_ERR_CAPACITY: return "capacity exceeded";
    case CCAL_ERR_IO: return "i/o error";
    default: return "unknown status";
    }
}

extern "C" int ccal_create(int device_id, ccal_ctx** out)
{
    if (!out) return CCAL_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CCAL_ERR_DEVICE;
    if (device_id < 0 || device_id >= n) return CCAL_ERR_ARG;
    ccal_ctx* ctx = new (std::nothrow) ccal_ctx;
    if (!ctx) return CCAL_ERR_DEVICE;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return CCAL_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cu = prop.multiProcessorCount;
    // this device's copy of the static scratch array, and a pinned result slot its last kernel writes
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)) != hipSuccess || !sym ||
        ctx->static_out.ensure(kMoments * sizeof(double)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        ctx->static_out.release();
        delete ctx;
        return CCAL_ERR_DEVICE;
    }
    ctx->static_scratch = static_cast<double*>(sym);
    *out = ctx;
    return CCAL_OK;
}

extern "C" int ccal_destroy(ccal_ctx* ctx)
{
    if (!ctx) return CCAL_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->sets) s.release();
    DevBuf* bufs[] = {&ctx->w_buf, &ctx->partial_buf, &ctx->cost_buf, &ctx->rec_buf,
                      &ctx->minrec_buf, &ctx->list_buf, &ctx->count_buf, &ctx->mats_buf,
                      &ctx->planes_buf, &ctx->psum_buf, &ctx->osum_buf,
                      &ctx->omax_buf, &ctx->tbl_buf, &ctx->dist_buf, &ctx->sep_wr,
                      &ctx->sep_sb, &ctx->sep_perm, &ctx->sep_tab, &ctx->sep_meta, &ctx->sep_wcnt, &ctx->sep_wsum};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : ctx->vox) b.release();
    ctx->exact_in.release();
    ctx->exact_out.release();
    ctx->screen_in.release();
    ctx->screen_out.release();
    ctx->screen_partial.release();
    ctx->screen_w.release();
    ctx->split_maps.release();
    ctx->split_first.release();
    ctx->static_out.release();
    {
        ccal::Staging& S = ctx->staging;
        if (S.stream) (void)hipStreamSynchronize(S.stream);
        for (ccal::StageLane& L : S.lanes)
            for (int k = 0; k < 2; ++k) {
                L.buf[k].release();
                if (L.ev[k]) (void)hipEventDestroy(L.ev[k]);
            }
        S.lanes.clear();
        for (hipEvent_t& e : S.batch_ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (S.stream) (void)hipStreamDestroy(S.stream);
        S.stream = nullptr;
    }
    for (hipEvent_t& e : ctx->prof_ev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CCAL_OK;
}

extern "C" const char* ccal_last_error(const ccal_ctx* ctx)
{
    return ctx ? ctx->err.c_str() : "null context";
}

extern "C" void ccal_pose_to_matrix(const double pose[6], double T[16])
{
    SinCos3 sc = sincos_of_pose(pose);
    double R[9];
    rotation_zyx(sc.s, sc.c, R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = pose[i];
    }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// ---------------------------------------------------------------------------
// upload
// ---------------------------------------------------------------------------

// The caller's clouds (pageable memory, one array per cloud) into one device AoS array, and from
// there -- batch of clouds by batch -- centres, SoA arrays and centred groups (`tile_batch`).
//   small sets: one copy call per cloud on the context's stream, then all clouds as one batch;
//   large sets: the byte stream of all clouds is cut into chunks.  A few host threads take chunks in
//               turn, copy them into pinned buffers of their own and hand each to ONE copy stream
//               (asynchronous copy + an event that frees the buffer), so the host-side copies of
//               several threads and the DMA of finished chunks overlap -- a copy call on pageable
//               memory does both steps one after the other inside the runtime (20 GB/s).  The calling
//               thread follows the chunks as they are handed over: once all chunks of a batch of clouds
//               are on the copy stream it lets the context's stream wait for them and enqueues that
//               batch's kernels, which then run while the next batch is still arriving.
//               Pinned buffers, the copy stream and the events stay with the context.
constexpr size_t kStageChunk = (size_t)8 << 20;
constexpr size_t kStageThreshold = (size_t)192 << 20;
constexpr int kStageLanes = 8;
constexpr int kStageBatches = 8;

// Stride of the fast path's point order (fast_slot_source): the integer nearest n / golden ratio or the
// next one above it that is coprime to n.
static int64_t fast_slot_stride(int64_t n)
{
    if (n <= 2 || n >= ((int64_t)1 << 32)) return 1;     // slot * stride must stay below 2^64
    auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
    int64_t s = (int64_t)((double)n * 0.6180339887498949);
    if (s < 1) s = 1;
    while (gcd(s, n) != 1) ++s;      // n - 1 is coprime to n: the loop ends below n
    return s;
}

static int staging_lanes(size_t n_chunks)
{
    return (int)std::min<size_t>({(size_t)kStageLanes, n_chunks, std::max(1u, std::thread::hardware_concurrency())});
}

// The copy stream, its events and two pinned chunk buffers per lane; all stay with the context.
static hipError_t ensure_staging(ccal_ctx* ctx, int n_lanes)
{
    Staging& S = ctx->staging;
    if (!S.stream) {
        hipError_t e = hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking);
        for (int b = 0; b < kStageBatches && e == hipSuccess; ++b)
            e = hipEventCreateWithFlags(&S.batch_ev[b], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    if ((int)S.lanes.size() < n_lanes) S.lanes.resize(n_lanes);
    for (int t = 0; t < n_lanes; ++t) {
        StageLane& L = S.lanes[t];
        for (int k = 0; k < 2; ++k) {
            if (L.ev[k]) continue;
            hipError_t e = L.buf[k].ensure(kStageChunk);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&L.ev[k], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

// A result array of the context's stream into the caller's (pageable) memory; returns with the bytes in
// place.  Small arrays: one copy call.  Large ones take the staging lanes the other way round: the copy
// stream brings chunks into pinned buffers, one thread per lane copies each on into the caller's array
// while its other buffer is being filled -- a copy call on pageable memory does the two steps one after
// the other inside the runtime (9 GB/s for 800 MB).
static hipError_t download_array(ccal_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    if (bytes < kStageThreshold) {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
        return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
    }
    const size_t n_chunks = (bytes + kStageChunk - 1) / kStageChunk;
    const int n_lanes = staging_lanes(n_chunks);
    if (hipError_t e = ensure_staging(ctx, n_lanes); e != hipSuccess) return e;
    Staging& S = ctx->staging;
    // the copy stream starts once the context's stream has produced the array
    hipError_t e = hipEventRecord(S.batch_ev[0], ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(S.stream, S.batch_ev[0], 0);
    if (e != hipSuccess) return e;
    std::atomic<size_t> next_chunk{0};
    std::atomic<int> failed{0};
    std::vector<hipError_t> errs(n_lanes, hipSuccess);
    std::mutex order;      // chunks enter the (in-order) copy stream in the order they are taken
    auto work = [&](int t) {
        StageLane& L = S.lanes[t];
        hipError_t le = hipSetDevice(ctx->device);
        auto take = [&](int slot, size_t* c_out) -> bool {      // next chunk into buf[slot]
            std::lock_guard<std::mutex> lock(order);
            const size_t c = next_chunk.load();
            if (c >= n_chunks || le != hipSuccess || failed.load()) return false;
            if (L.pending[slot]) le = hipEventSynchronize(L.ev[slot]);    // an upload's copy out of this buffer
            L.pending[slot] = false;
            const size_t b = c * kStageChunk, len = std::min(kStageChunk, bytes - b);
            if (le == hipSuccess)
                le = hipMemcpyAsync(L.buf[slot].p, static_cast<const char*>(src) + b, len, hipMemcpyDeviceToHost, S.stream);
            if (le == hipSuccess) le = hipEventRecord(L.ev[slot], S.stream);
            if (le != hipSuccess) return false;
            next_chunk.store(c + 1);
            *c_out = c;
            return true;
        };
        size_t cur = 0, nxt = 0;
        int slot = 0;
        bool have = take(slot, &cur);
        while (have) {
            const bool more = take(slot ^ 1, &nxt);
            if (le == hipSuccess) le = hipEventSynchronize(L.ev[slot]);
            if (le != hipSuccess) break;
            const size_t b = cur * kStageChunk, len = std::min(kStageChunk, bytes - b);
            std::memcpy(static_cast<char*>(dst) + b, L.buf[slot].p, len);
            have = more;
            cur = nxt;
            slot ^= 1;
        }
        if (le != hipSuccess) failed.store(1);
        errs[t] = le;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_lanes; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    // whatever a failed lane left on the copy stream must not outlive the caller's buffers
    hipError_t es = hipStreamSynchronize(S.stream);
    for (hipError_t le : errs)
        if (le != hipSuccess) return le;
    return es;
}

template <typename TileBatch>
static hipError_t stage_clouds(ccal_ctx* ctx, int n_clouds, const float* const* xyz, const size_t* n_pts,
                               const std::vector<int64_t>& pt_off, float* stage, TileBatch tile_batch)
{
    const size_t total_bytes = (size_t)pt_off[n_clouds] * 3 * sizeof(float);
    if (total_bytes < kStageThreshold) {
        for (int j = 0; j < n_clouds; ++j) {
            hipError_t e = hipMemcpyAsync(stage + 3 * (size_t)pt_off[j], xyz[j], n_pts[j] * 3 * sizeof(float),
                                          hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) return e;
        }
        tile_batch(0, n_clouds);
        return hipSuccess;
    }
    const size_t n_chunks = (total_bytes + kStageChunk - 1) / kStageChunk;
    const int n_lanes = staging_lanes(n_chunks);
    Staging& S = ctx->staging;
    if (hipError_t e = ensure_staging(ctx, n_lanes); e != hipSuccess) return e;
    // batches: runs of whole clouds of about total / kStageBatches bytes; batch b is complete once every
    // chunk below last_chunk[b] is on the copy stream
    std::vector<size_t> byte_off(n_clouds + 1);
    for (int j = 0; j <= n_clouds; ++j) byte_off[j] = (size_t)pt_off[j] * 3 * sizeof(float);
    std::vector<int> batch_end;          // exclusive cloud index
    std::vector<size_t> batch_chunks;    // chunks that must have been handed over
    {
        const size_t want = std::max<size_t>(4 * kStageChunk, total_bytes / kStageBatches);
        int j0 = 0;
        while (j0 < n_clouds) {
            int j1 = j0 + 1;
            while (j1 < n_clouds && byte_off[j1] - byte_off[j0] < want && (int)batch_end.size() < kStageBatches - 1) ++j1;
            if ((int)batch_end.size() == kStageBatches - 1) j1 = n_clouds;
            batch_end.push_back(j1);
            batch_chunks.push_back((byte_off[j1] + kStageChunk - 1) / kStageChunk);
            j0 = j1;
        }
    }
    std::vector<std::atomic<unsigned char>> handed(n_chunks);
    for (auto& h : handed) h.store(0, std::memory_order_relaxed);
    std::atomic<size_t> next_chunk{0};
    std::atomic<int> failed{0};
    std::vector<hipError_t> errs(n_lanes, hipSuccess);
    auto work = [&](int t) {
        StageLane& L = S.lanes[t];
        hipError_t e = hipSetDevice(ctx->device);
        int slot = 0;
        for (size_t c = next_chunk++; c < n_chunks && e == hipSuccess && !failed.load(); c = next_chunk++) {
            const size_t b = c * kStageChunk, len = std::min(kStageChunk, total_bytes - b);
            if (L.pending[slot]) e = hipEventSynchronize(L.ev[slot]);   // the buffer's previous copy has left it
            if (e != hipSuccess) break;
            L.pending[slot] = false;
            char* dst = L.buf[slot].as<char>();
            int j = (int)(std::upper_bound(byte_off.begin(), byte_off.end(), b) - byte_off.begin()) - 1;
            for (size_t done = 0; done < len; ++j) {
                const size_t in_cloud = b + done - byte_off[j];
                const size_t n = std::min(len - done, byte_off[j + 1] - byte_off[j] - in_cloud);
                std::memcpy(dst + done, reinterpret_cast<const char*>(xyz[j]) + in_cloud, n);
                done += n;
            }
            e = hipMemcpyAsync(reinterpret_cast<char*>(stage) + b, dst, len, hipMemcpyHostToDevice, S.stream);
            if (e == hipSuccess) e = hipEventRecord(L.ev[slot], S.stream);
            L.pending[slot] = e == hipSuccess;
            slot ^= 1;
            handed[c].store(1, std::memory_order_release);
        }
        if (e != hipSuccess) failed.store(1);
        errs[t] = e;
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_lanes; ++t) pool.emplace_back(work, t);
    hipError_t e = hipSuccess;
    size_t seen = 0;
    int j0 = 0;
    for (size_t b = 0; b < batch_end.size() && e == hipSuccess; ++b) {
        while (seen < batch_chunks[b] && !failed.load()) {
            if (handed[seen].load(std::memory_order_acquire)) ++seen;
            else std::this_thread::yield();
        }
        if (failed.load()) break;
        // the copy stream is in order: an event recorded now follows every chunk handed over so far
        e = hipEventRecord(S.batch_ev[b], S.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, S.batch_ev[b], 0);
        if (e == hipSuccess) tile_batch(j0, batch_end[b]);
        j0 = batch_end[b];
    }
    if (e != hipSuccess) failed.store(1);
    for (auto& th : pool) th.join();
    for (hipError_t le : errs)
        if (le != hipSuccess) return le;
    return e;
}

extern "C" int ccal_upload_clouds(ccal_ctx* ctx, int which, int n_clouds,
                                  const float* const* xyz, const size_t* n_pts,
                                  const double* base_T_flange)
{
    if (!ctx) return CCAL_ERR_ARG;
    if (which != CCAL_DOWNSAMPLED && which != CCAL_RAW)
        return fail(ctx, CCAL_ERR_ARG, "cloud-set selector must be CCAL_DOWNSAMPLED or CCAL_RAW");
    if (n_clouds <= 0 || !xyz || !n_pts || !base_T_flange)
        return fail(ctx, CCAL_ERR_ARG, "upload: null or empty argument");
    size_t total = 0;
    for (int j = 0; j < n_clouds; ++j) {
        if (n_pts[j] == 0 || !xyz[j])
            return fail(ctx, CCAL_ERR_ARG, "upload: empty cloud (the objective divides by its size)");
        total += n_pts[j];
    }
    if (total >= ((size_t)1 << 37)) return fail(ctx, CCAL_ERR_ARG, "upload: too many points");
    if (int rc = bind(ctx)) return rc;

    CloudSet& cs = ctx->sets[which];
    cs.ready = false;
    cs.n_clouds = n_clouds;
    cs.n_pts.assign(n_pts, n_pts + n_clouds);
    cs.base_T_flange.assign(base_T_flange, base_T_flange + 16 * (size_t)n_clouds);
    cs.total_pts = total;

    // ---- tables: block table of the exact path, group offsets of the fast path ----------
    std::vector<BlockDesc> blocks;
    cs.cloud_block_off.assign(n_clouds + 1, 0);
    std::vector<int64_t> pt_off(n_clouds + 1, 0);
    for (int j = 0; j < n_clouds; ++j) {
        cs.cloud_block_off[j] = (int)blocks.size();
        for (size_t b = 0; b < n_pts[j]; b += kPointsPerBlock) {
            BlockDesc bd;
            bd.cloud = j;
            bd.first_of_cloud = cs.cloud_block_off[j];
            bd.begin = pt_off[j] + (int64_t)b;
            bd.end = pt_off[j] + (int64_t)std::min(n_pts[j], b + (size_t)kPointsPerBlock);
            blocks.push_back(bd);
        }
        pt_off[j + 1] = pt_off[j] + (int64_t)n_pts[j];
    }
    cs.cloud_block_off[n_clouds] = (int)blocks.size();
    cs.n_blocks = (int)blocks.size();
    // 512-element segments of every chain (counted from the cloud's start rounded down to a run), one
    // slot more per cloud for the sum at the chain's end
    std::vector<int64_t> seg_off(n_clouds + 1, 0);
    for (int j = 0; j < n_clouds; ++j) {
        const int64_t b0 = pt_off[j] & ~(int64_t)(kChainRun - 1);
        seg_off[j + 1] = seg_off[j] + (pt_off[j + 1] - b0 + kChainSeg - 1) / kChainSeg + 1;
    }
    cs.total_segs = seg_off[n_clouds];

    cs.group_off.assign(n_clouds + 1, 0);
    cs.centres.assign(3 * (size_t)n_clouds, 0.0);
    int64_t n_groups = 0;
    cs.min_groups = std::numeric_limits<int>::max();
    for (int j = 0; j < n_clouds; ++j) {
        cs.group_off[j] = (int)n_groups;
        int64_t g = ((int64_t)n_pts[j] + kGroupPoints - 1) / kGroupPoints;
        cs.min_groups = (int)std::min<int64_t>(cs.min_groups, g);
        n_groups += g;
    }
    if (n_groups >= ((int64_t)1 << 31)) return fail(ctx, CCAL_ERR_ARG, "upload: too many points");
    cs.group_off[n_clouds] = (int)n_groups;
    cs.n_groups = n_groups;
    cs.n_pad_points = n_groups * kGroupPoints - (int64_t)total;

    std::vector<double> scale(n_clouds);
    for (int j = 0; j < n_clouds; ++j) scale[j] = 1.0 / ((double)n_clouds * (double)n_pts[j]);

    // ---- device: one copy of the caller's AoS clouds, centres and both layouts made from it ------
    CCAL_HIP(ctx, cs.xs.ensure(total * sizeof(float)));
    CCAL_HIP(ctx, cs.ys.ensure(total * sizeof(float)));
    CCAL_HIP(ctx, cs.zs.ensure(total * sizeof(float)));
    CCAL_HIP(ctx, cs.groups.ensure((size_t)n_groups * 256 * sizeof(float)));
    CCAL_HIP(ctx, hipMemsetAsync(cs.groups.p, 0, (size_t)n_groups * 256 * sizeof(float), ctx->stream));
    CCAL_HIP(ctx, cs.radius_dev.ensure(n_clouds * sizeof(unsigned int)));
    CCAL_HIP(ctx, hipMemsetAsync(cs.radius_dev.p, 0, n_clouds * sizeof(unsigned int), ctx->stream));
    CCAL_HIP(ctx, cs.cloud_pt_off_dev.ensure((n_clouds + 1) * sizeof(int64_t)));
    CCAL_HIP(ctx, hipMemcpyAsync(cs.cloud_pt_off_dev.p, pt_off.data(), (n_clouds + 1) * sizeof(int64_t),
                                 hipMemcpyHostToDevice, ctx->stream));
    CCAL_HIP(ctx, cs.centres_dev.ensure(3 * n_clouds * sizeof(double)));
    std::vector<unsigned int> radius2_bits(n_clouds, 0u);
    std::vector<size_t> rblock_off(n_clouds + 1, 0);
    for (int j = 0; j < n_clouds; ++j) rblock_off[j + 1] = rblock_off[j] + (n_pts[j] + 255) / 256;
    std::vector<float> radius_sums(rblock_off[n_clouds], 0.f);
    DevBuf radius_sums_dev;   // per tiling block: sum of its points' distances from the centre
    const bool trace_upload = std::getenv("CCAL_TRACE_UPLOAD") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tr0 = now();
    double tr1 = 0, tr2 = 0, tr3 = 0;
    {
        DevBuf stage, centre_part;   // the AoS copy lives only for this call
        hipError_t e = stage.ensure(total * 3 * sizeof(float));
        if (e == hipSuccess) e = radius_sums_dev.ensure(radius_sums.size() * sizeof(float));
        if (e == hipSuccess) e = centre_part.ensure((size_t)n_clouds * kCentreParts * 3 * sizeof(double));
        tr1 = now();
        auto tile_batch = [&](int j0, int j1) {
            hipLaunchKernelGGL(centre_partial_kernel, dim3(kCentreParts, j1 - j0), dim3(256), 0, ctx->stream,
                               stage.as<float>(), cs.cloud_pt_off_dev.as<int64_t>(), j0, centre_part.as<double>());
            hipLaunchKernelGGL(centre_final_kernel, dim3((3 * (j1 - j0) + 255) / 256), dim3(256), 0, ctx->stream,
                               centre_part.as<double>(), cs.cloud_pt_off_dev.as<int64_t>(), j0, j1,
                               cs.centres_dev.as<double>());
            for (int j = j0; j < j1; ++j) {
                const float* src = stage.as<float>() + 3 * (size_t)pt_off[j];
                hipLaunchKernelGGL(tile_cloud_kernel, dim3((unsigned)((n_pts[j] + 255) / 256)), dim3(256), 0, ctx->stream,
                                   src, (int64_t)n_pts[j], fast_slot_stride((int64_t)n_pts[j]), pt_off[j], (int64_t)cs.group_off[j],
                                   cs.centres_dev.as<double>() + 3 * (size_t)j, cs.xs.as<float>(), cs.ys.as<float>(),
                                   cs.zs.as<float>(), cs.groups.as<float>(), cs.radius_dev.as<unsigned int>() + j,
                                   radius_sums_dev.as<float>() + rblock_off[j]);
            }
        };
        if (e == hipSuccess) e = stage_clouds(ctx, n_clouds, xyz, n_pts, pt_off, stage.as<float>(), tile_batch);
        tr2 = now();
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(cs.centres.data(), cs.centres_dev.p, 3 * (size_t)n_clouds * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(radius2_bits.data(), cs.radius_dev.p, n_clouds * sizeof(unsigned int),
                               hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(radius_sums.data(), radius_sums_dev.p, radius_sums.size() * sizeof(float),
                               hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        tr3 = now();
        stage.release();
        centre_part.release();
        radius_sums_dev.release();
        if (trace_upload)
            std::fprintf(stderr, "upload: alloc %.2f ms, staging %.2f ms, drain %.2f ms, free %.2f ms\n", tr1 - tr0, tr2 - tr1,
                         tr3 - tr2, now() - tr3);
        CCAL_HIP(ctx, e);
    }
    CCAL_HIP(ctx, cs.blocks.ensure(blocks.size() * sizeof(BlockDesc)));
    CCAL_HIP(ctx, cs.cloud_block_off_dev.ensure((n_clouds + 1) * sizeof(int)));
    CCAL_HIP(ctx, cs.group_off_dev.ensure((n_clouds + 1) * sizeof(int)));
    CCAL_HIP(ctx, cs.scale_dev.ensure(n_clouds * sizeof(double)));
    CCAL_HIP(ctx, cs.base_dev.ensure(16 * (size_t)n_clouds * sizeof(double)));
    CCAL_HIP(ctx, hipMemcpy(cs.blocks.p, blocks.data(), blocks.size() * sizeof(BlockDesc),
                            hipMemcpyHostToDevice));
    CCAL_HIP(ctx, hipMemcpy(cs.cloud_block_off_dev.p, cs.cloud_block_off.data(),
                            (n_clouds + 1) * sizeof(int), hipMemcpyHostToDevice));
    CCAL_HIP(ctx, cs.cloud_seg_off_dev.ensure((n_clouds + 1) * sizeof(int64_t)));
    CCAL_HIP(ctx, hipMemcpy(cs.cloud_seg_off_dev.p, seg_off.data(), (n_clouds + 1) * sizeof(int64_t),
                            hipMemcpyHostToDevice));
    CCAL_HIP(ctx, cs.seg_sums.ensure((size_t)cs.total_segs * sizeof(double)));
    CCAL_HIP(ctx, hipMemset(cs.seg_sums.p, 0, (size_t)cs.total_segs * sizeof(double)));
    cs.seg_sums_valid = false;
    cs.radius.assign(n_clouds, 0.0);
    for (int j = 0; j < n_clouds; ++j) {
        float r2;
        std::memcpy(&r2, &radius2_bits[j], sizeof r2);
        // FP32 evaluation of the squared length: a few ulps, covered by the factor
        cs.radius[j] = std::isfinite(r2) ? std::sqrt((double)r2) * (1.0 + 1e-6) : std::numeric_limits<double>::infinity();
    }
    // mean distance from the centre (float square roots and block sums: the factor covers their roundings)
    // and a bound on the 2-norm of the robot pose's 3x3 part: 1 for a rotation, else its Frobenius norm
    cs.radius_mean.assign(n_clouds, 0.0);
    cs.rot_norm.assign(n_clouds, 0.0);
    for (int j = 0; j < n_clouds; ++j) {
        double sum = 0.0;
        for (size_t b = rblock_off[j]; b < rblock_off[j + 1]; ++b) sum += (double)radius_sums[b];
        cs.radius_mean[j] = std::isfinite(cs.radius[j]) && std::isfinite(sum)
                                ? std::min(cs.radius[j], sum / (double)n_pts[j] * (1.0 + 1e-5))
                                : std::numeric_limits<double>::infinity();
        const double* B = base_T_flange + 16 * (size_t)j;
        double off = 0.0, fro = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double g = 0.0;
                for (int k = 0; k < 3; ++k) g += B[k * 4 + a] * B[k * 4 + b];
                off = std::max(off, std::fabs(g - (a == b ? 1.0 : 0.0)));
                fro += B[a * 4 + b] * B[a * 4 + b];
            }
        cs.rot_norm[j] = off <= 1e-6 ? 1.0 + 2e-6 : std::sqrt(fro);
        if (!std::isfinite(cs.rot_norm[j])) cs.rot_norm[j] = std::numeric_limits<double>::infinity();
    }
    cs.split_ready = false;
    if (ctx->fast_kernel == CCAL_FAST_BF16X3_WIDE)
        if (int rc = ensure_split_groups(ctx, cs)) return rc;
    CCAL_HIP(ctx, hipMemcpy(cs.group_off_dev.p, cs.group_off.data(), (n_clouds + 1) * sizeof(int),
                            hipMemcpyHostToDevice));
    CCAL_HIP(ctx, hipMemcpy(cs.scale_dev.p, scale.data(), n_clouds * sizeof(double),
                            hipMemcpyHostToDevice));
    CCAL_HIP(ctx, hipMemcpy(cs.base_dev.p, cs.base_T_flange.data(),
                            16 * (size_t)n_clouds * sizeof(double), hipMemcpyHostToDevice));
    cs.ready = true;
    return CCAL_OK;
}

extern "C" int ccal_set_fast_kernel(ccal_ctx* ctx, int kind)
{
    if (!ctx) return CCAL_ERR_ARG;
    if (kind != CCAL_FAST_BF16X3 && kind != CCAL_FAST_F32 && kind != CCAL_FAST_BF16X3_WIDE && kind != CCAL_FAST_BF16X3_8)
        return fail(ctx, CCAL_ERR_ARG,
                    "set_fast_kernel: kind must be CCAL_FAST_BF16X3, CCAL_FAST_F32, CCAL_FAST_BF16X3_WIDE or CCAL_FAST_BF16X3_8");
    ctx->fast_kernel = kind;
    if (kind == CCAL_FAST_BF16X3_WIDE) {
        if (int rc = bind(ctx)) return rc;
        for (CloudSet& cs : ctx->sets)
            if (cs.ready)
                if (int rc = ensure_split_groups(ctx, cs)) return rc;
    }
    return CCAL_OK;
}

extern "C" int ccal_set_search_mode(ccal_ctx* ctx, int mode)
{
    if (!ctx) return CCAL_ERR_ARG;
    if (mode != CCAL_SEARCH_DENSE && mode != CCAL_SEARCH_SEPARABLE)
        return fail(ctx, CCAL_ERR_ARG, "set_search_mode: mode must be CCAL_SEARCH_DENSE or CCAL_SEARCH_SEPARABLE");
    ctx->search_mode = mode;
    return CCAL_OK;
}

extern "C" int ccal_set_plane(ccal_ctx* ctx, const float plane[4])
{
    if (!ctx || !plane) return CCAL_ERR_ARG;
    double n2 = (double)plane[0] * plane[0] + (double)plane[1] * plane[1] + (double)plane[2] * plane[2];
    if (!(n2 > 0.0) || !std::isfinite(n2) || !std::isfinite(plane[3]))
        return fail(ctx, CCAL_ERR_ARG, "plane normal must be finite and non-zero");
    std::memcpy(ctx->plane, plane, 4 * sizeof(float));
    ctx->plane_set = true;
    return CCAL_OK;
}

extern "C" int ccal_get_plane(const ccal_ctx* ctx, float plane[4])
{
    if (!ctx || !plane) return CCAL_ERR_ARG;
    if (!ctx->plane_set) return CCAL_ERR_STATE;
    std::memcpy(plane, ctx->plane, 4 * sizeof(float));
    return CCAL_OK;
}

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

int ccal::eval_exact(ccal_ctx* ctx, int which, int n_poses, const double* poses6,
                     const double* planes4, double* cost, double* per_cloud_mean,
                     double* per_cloud_max, bool allow_split)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (n_poses < 0 || (n_poses > 0 && (!poses6 || !cost)))
        return fail(ctx, CCAL_ERR_ARG, "eval: null argument");
    if (!planes4 && !ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "eval: plane not set");
    if (n_poses == 0) return CCAL_OK;
    if (int rc = bind(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    const int C = cs.n_clouds;

    // one workgroup per (cloud, pose) chain; the batch bound only keeps the grid and the
    // constant tables modest
    // Pose tables and results live in pinned host memory that the kernel addresses directly: a
    // request is one launch and one stream wait, with no copy call on either side.
    const int max_batch = 4096;
    for (int k0 = 0; k0 < n_poses; k0 += max_batch) {
        const int nb = std::min(max_batch, n_poses - k0);
        const size_t n_mats = (size_t)nb * C * 12, n_planes = (size_t)nb * kPlaneStride, n_out = (size_t)nb * C;
        CCAL_HIP(ctx, ctx->exact_in.ensure((n_mats + n_planes) * sizeof(double)));
        CCAL_HIP(ctx, ctx->exact_out.ensure(2 * n_out * sizeof(double)));
        double* mats = ctx->exact_in.as<double>();
        double* planes = mats + n_mats;
        double* hsum = ctx->exact_out.as<double>();
        double* hmax = hsum + n_out;
        for (int k = 0; k < nb; ++k) {
            const int src = k0 + k;
            pose_matrices(cs, poses6 + 6 * (size_t)src, mats + (size_t)k * C * 12);
            double* pl = planes + (size_t)k * kPlaneStride;
            if (planes4) {
                for (int e = 0; e < 4; ++e) pl[e] = planes4[4 * (size_t)src + e];
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
        const bool split = ctx->chain_split && allow_split && !per_cloud_max && chains < 8 * (int64_t)ctx->n_cu &&
                           cs.seg_sums_valid &&
                           (int64_t)nb * cs.total_segs * (int64_t)sizeof(SegMap) <= ((int64_t)1 << 30);   // records: at most 1 GiB
        const bool prof = ctx->profiling;
        if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[0], ctx->stream));
        if (split) {
            // parts per chain: enough workgroups to fill the chip several times over (measured on 620
            // chains: 4 parts 0.578 s, 8 parts 0.562 s, 12 parts 0.560 s per descent)
            const int K = (int)std::min<int64_t>(64, std::max<int64_t>(8, (16 * (int64_t)ctx->n_cu + chains - 1) / chains));
            CCAL_HIP(ctx, ctx->split_maps.ensure((size_t)nb * cs.total_segs * sizeof(SegMap)));
            CCAL_HIP(ctx, ctx->split_first.ensure(n_out * sizeof(double)));
            hipLaunchKernelGGL(split_chain_kernel, dim3((unsigned)(chains * K)), dim3(64 * kSplitWaves), 0, ctx->stream,
                               cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(), cs.cloud_pt_off_dev.as<int64_t>(),
                               cs.cloud_seg_off_dev.as<int64_t>(), mats, planes, nb, C, (int64_t)cs.total_pts, K,
                               cs.seg_sums.as<double>(), cs.total_segs, ctx->split_maps.as<SegMap>(),
                               ctx->split_first.as<double>());
            if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[1], ctx->stream));
            hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)chains), dim3(64), 0, ctx->stream, cs.xs.as<float>(),
                               cs.ys.as<float>(), cs.zs.as<float>(), cs.cloud_pt_off_dev.as<int64_t>(),
                               cs.cloud_seg_off_dev.as<int64_t>(), mats, planes, nb, C, (int64_t)cs.total_pts, K,
                               ctx->split_maps.as<SegMap>(), cs.total_segs, ctx->split_first.as<double>(),
                               cs.seg_sums.as<double>(), hsum, hmax);
        } else {
            launch_exact_chains(ctx, cs, nb, per_cloud_max != nullptr, ExactTables{mats, planes, hsum, hmax});
        }
        CCAL_HIP(ctx, hipGetLastError());
        if (prof) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[2], ctx->stream));
        CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (prof) {
            float ms = 0.f, ms2 = 0.f;
            const double pp = (double)nb * (double)cs.total_pts;
            if (split) {
                if (hipEventElapsedTime(&ms, ctx->prof_ev[0], ctx->prof_ev[1]) == hipSuccess &&
                    hipEventElapsedTime(&ms2, ctx->prof_ev[1], ctx->prof_ev[2]) == hipSuccess) {
                    ctx->prof.split_chain_ms += ms;
                    ctx->prof.split_finish_ms += ms2;
                    ctx->prof.split_pose_points += pp;
                    ++ctx->prof.split_requests;
                }
            } else if (hipEventElapsedTime(&ms, ctx->prof_ev[0], ctx->prof_ev[2]) == hipSuccess) {
                ctx->prof.whole_chain_ms += ms;
                ctx->prof.whole_pose_points += pp;
                ++ctx->prof.whole_requests;
            }
        }
        ctx->sets[which].seg_sums_valid = true;
        if (split) {
            for (size_t i = 0; i < n_out; ++i) ctx->split_resummed += (int64_t)hmax[i];
            ctx->split_segments += (int64_t)nb * (cs.total_segs - C);
        }
        for (int k = 0; k < nb; ++k) {
            double error = 0.0;
            for (int j = 0; j < C; ++j) {
                const double mean = hsum[(size_t)k * C + j] / (double)cs.n_pts[j];  // :309
                error += mean;
                if (per_cloud_mean) per_cloud_mean[(size_t)(k0 + k) * C + j] = mean;
                if (per_cloud_max) per_cloud_max[(size_t)(k0 + k) * C + j] = hmax[(size_t)k * C + j];
            }
            cost[k0 + k] = error / (double)C;  // :312
        }
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
// rho = MEAN distance of the cloud's points from its centre c (the budgets below are per point and linear
// in |p - c|; the objective is a mean over the points), m = |M (c,1)| (the centre carried to the base
// frame) and w4 = the folded constant (the signed distance of the centre):
//   exact path:  rounds every transformed coordinate q_r to float (src/helpers.hpp:151), and the distance weighs
//                coordinate r with |n_r|/|n|: the residual moves by at most
//                2^-24 sum_r |n_r|/|n| |q_r| <= 2^-24 sum_r |n_r|/|n| (|(M (c,1))_r| + |R|_2 |p - c|)
//                                                                       (R = the 3x3 part of M)
//   fast path:   w and p - c rounded to float (one rounding each), then three fused multiply-adds (the
//                innermost product is by exactly 1 and does not round), each on a magnitude of at most
//                |w123||p - c| + |w4|:  5 * 2^-24 (|w123| rho + |w4|)
//   sums:        4 + 6 FP32 adds per partial (four points per lane, six levels of the wave reduction):
//                10 * 2^-24 of the cloud's mean residual
// The FP64 arithmetic on either side is below 1e-12 of these magnitudes.  kScreenSlack rounds all of it up.
constexpr double kScreenSlack = 1.25;

int ccal::eval_screen(ccal_ctx* ctx, int which, int n_poses, const double* poses6, const double* planes4,
                      double* cost, double* bound)
{
    if (int rc = check_set(ctx, which)) return rc;
    if (n_poses < 0 || (n_poses > 0 && (!poses6 || !cost || !bound)))
        return fail(ctx, CCAL_ERR_ARG, "eval_screen: null argument");
    if (!planes4 && !ctx->plane_set) return fail(ctx, CCAL_ERR_STATE, "eval_screen: plane not set");
    if (n_poses == 0) return CCAL_OK;
    if (int rc = bind(ctx)) return rc;
    const CloudSet& cs = ctx->sets[which];
    const int C = cs.n_clouds;
    const double u24 = 0x1p-24;
    std::vector<double> mats((size_t)C * 12), fixed((size_t)C);
    const int max_batch = 1024;
    for (int k0 = 0; k0 < n_poses; k0 += max_batch) {
        const int nb = std::min(max_batch, n_poses - k0);
        CCAL_HIP(ctx, ctx->screen_in.ensure((size_t)nb * C * 4 * sizeof(float)));
        CCAL_HIP(ctx, ctx->screen_out.ensure((size_t)nb * C * sizeof(double)));
        CCAL_HIP(ctx, ctx->screen_partial.ensure((size_t)cs.n_blocks * kScreenQuarters * nb * sizeof(double)));
        float* w = ctx->screen_in.as<float>();
        double* sums = ctx->screen_out.as<double>();
        std::vector<double> budget((size_t)nb * C);
        for (int k = 0; k < nb; ++k) {
            const int src = k0 + k;
            pose_matrices(cs, poses6 + 6 * (size_t)src, mats.data());
            double pl[4];
            for (int e = 0; e < 4; ++e) pl[e] = planes4 ? planes4[4 * (size_t)src + e] : (double)ctx->plane[e];
            const double inv_norm = 1.0 / std::sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
            for (int j = 0; j < C; ++j) {
                const double* M = mats.data() + 12 * (size_t)j;
                const double* c = cs.centres.data() + 3 * (size_t)j;
                double wv[4];
                fold_plane(M, pl, pl[3], inv_norm, c, wv);
                for (int e = 0; e < 4; ++e) w[((size_t)k * C + j) * 4 + e] = (float)wv[e];
                // |q_i| <= |R|_2 |p_i - c| + |M (c,1)| for a transformed point and |w123 . (p_i - c)| <=
                // |w123| |p_i - c|: both budgets are linear in |p_i - c|, and the objective is a mean over the
                // points, so the MEAN distance from the centre enters.  R = (robot pose) * (a rotation): its
                // 2-norm is at most that of the robot pose's 3x3 part, whatever the caller uploaded.
                // The float rounding of the exact path acts on each coordinate of q, and the distance weighs
                // coordinate r with |n_r| / |n|: sum_r |n_r|/|n| (|(M (c,1))_r| + |R|_2 rho) -- for a plane
                // that is roughly level this is far below |q|, which is mostly horizontal offset.
                const double rho = cs.radius_mean[j];
                const double rn = cs.rot_norm[j] * (1.0 + 1e-9);
                double ex = 0.0;
                for (int r = 0; r < 3; ++r) {
                    const double q = M[r * 4] * c[0] + M[r * 4 + 1] * c[1] + M[r * 4 + 2] * c[2] + M[r * 4 + 3];
                    ex += std::fabs(pl[r]) * inv_norm * (std::fabs(q) + rn * rho);
                }
                const double wn = std::sqrt(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]);
                budget[(size_t)k * C + j] = u24 * ex + 5.0 * u24 * (wn * rho + std::fabs(wv[3]));
            }
        }
        // every block reads the whole table: it goes to device memory first (one small copy from pinned memory)
        CCAL_HIP(ctx, ctx->screen_w.ensure((size_t)nb * C * 4 * sizeof(float)));
        CCAL_HIP(ctx, hipMemcpyAsync(ctx->screen_w.p, w, (size_t)nb * C * 4 * sizeof(float), hipMemcpyHostToDevice,
                                     ctx->stream));
        if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[0], ctx->stream));
        hipLaunchKernelGGL(screen_cost_kernel, dim3(cs.n_blocks * kScreenQuarters), dim3(256), 0, ctx->stream, cs.groups.as<f32x4>(),
                           cs.group_off_dev.as<int>(), cs.cloud_pt_off_dev.as<int64_t>(), cs.blocks.as<BlockDesc>(),
                           ctx->screen_w.as<f32x4>(), nb, C, ctx->screen_partial.as<double>());
        hipLaunchKernelGGL(screen_reduce_kernel, dim3((nb * C + 3) / 4), dim3(256), 0, ctx->stream,
                           ctx->screen_partial.as<double>(), cs.cloud_block_off_dev.as<int>(), nb, C, sums);
        CCAL_HIP(ctx, hipGetLastError());
        if (ctx->profiling) CCAL_HIP(ctx, hipEventRecord(ctx->prof_ev[2], ctx->stream));
        CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->profiling) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->prof_ev[0], ctx->prof_ev[2]) == hipSuccess) {
                ctx->prof.screen_ms += ms;
                ctx->prof.screen_pose_points += (double)nb * (double)cs.total_pts;
                ++ctx->prof.screen_requests;
            }
        }
        for (int k = 0; k < nb; ++k) {
            double f = 0.0, b = 0.0;
            for (int j = 0; j < C; ++j) {
                const double mean = sums[(size_t)k * C + j] / (double)cs.n_pts[j];
                f += mean;
                b += budget[(size_t)k * C + j] + 10.0 * u24 * mean;
            }
            cost[k0 + k] = f / (double)C;
            bound[k0 + k] = std::isfinite(b) ? kScreenSlack * b / (double)C : std::numeric_limits<double>::infinity();
        }
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
    hipLaunchKernelGGL(point_error_kernel, dim3(cs.n_blocks), dim3(256), 0, ctx->stream,
                       cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.blocks.as<BlockDesc>(), ctx->mats_buf.as<double>(),
                       ctx->planes_buf.as<double>(), arithmetic == CCAL_POINT_VIEWER ? 1 : 0,
                       ctx->dist_buf.as<double>(), ctx->list_buf.as<unsigned char>());
    CCAL_HIP(ctx, hipGetLastError());
    if (dist_out) CCAL_HIP(ctx, download_array(ctx, dist_out, ctx->dist_buf.p, cs.total_pts * sizeof(double)));
    if (class_out) CCAL_HIP(ctx, download_array(ctx, class_out, ctx->list_buf.p, cs.total_pts));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
        // rotations per launch: bounded by the partial buffer (8 B per cloud, translation, rotation)
        const int64_t budget = (int64_t)1 << 30;
        int64_t nr_max = budget / (8 * (int64_t)C * nt);
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
    double total = 0.0;
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
        const double w4max = std::fabs(w4) + w4_slack;
        double budget;
        if (separable) {
            const double span = 2.0 * rn * T;
            // largest |residual| the histogram pass can meet: rotation part + constant, translation excluded;
            // its fixed point rounds by half a unit and takes the sign from the unrounded value (one more unit)
            const double vmax = (rn * cs.radius[j] + w4max) * (1.0 + 1e-5);
            if (!(vmax < 1e12)) gb.separable_ok = false;
            budget = u24 * ex + u24 * (6.0 * rn * rho + 5.0 * w4max + span) + 3.0 / separable_fx_scale(vmax);
        } else {
            budget = u24 * ex + k_dense * u24 * (rn * rho + w4max);
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
    const hipEvent_t ev_begin = ev[0], ev_end = ev[1];
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
    CCAL_HIP(ctx, hipEventRecord(ev_end, ctx->stream));

    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float total_ms = 0.f;
    (void)hipEventElapsedTime(&total_ms, ev_begin, ev_end);

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
    hipLaunchKernelGGL(moments_kernel, dim3(moment_rows(cs.n_blocks)), dim3(256), 0, ctx->stream,
                       cs.xs.as<float>(), cs.ys.as<float>(), cs.zs.as<float>(),
                       cs.blocks.as<BlockDesc>(), cs.n_blocks, ctx->mats_buf.as<double>(), org[0], org[1],
                       org[2], InlierMask{{0, 0, 0, 0, 1}, -1.0}, ctx->psum_buf.as<double>());
    hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                       ctx->psum_buf.as<double>(), moment_rows(cs.n_blocks), ctx->osum_buf.as<double>());
    CCAL_HIP(ctx, hipGetLastError());
    double m[kMoments];
    CCAL_HIP(ctx, hipMemcpyAsync(m, ctx->osum_buf.p, sizeof m, hipMemcpyDeviceToHost, ctx->stream));
    CCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
#include "voxel_impl.hpp"
