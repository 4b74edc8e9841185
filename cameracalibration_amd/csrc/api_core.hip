// api_core.hip -- C ABI of the calibration hot path (see include/ccal.h): contexts, upload, settings,
// and the host helpers the other translation units of libccal share (ccal_internal.hpp).
//
// Host orchestration only: pose matrices and grid tables are prepared in FP64 on
// the host (sines/cosines from libm, as the reference's Eigen calls do), every
// per-point loop runs in HIP kernels.  There is no CPU fallback: a missing or
// failing device is reported, never worked around.
#include "ccal_internal.hpp"
#include "kernels_ingest.hpp"

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

namespace ccal {

int fail(ccal_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return code;
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

} // namespace ccal

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
    double* sym = nullptr;
    if (static_scratch_address(&sym) != hipSuccess || !sym ||
        ctx->static_out.ensure(kMoments * sizeof(double)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        ctx->static_out.release();
        delete ctx;
        return CCAL_ERR_DEVICE;
    }
    ctx->static_scratch = sym;
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
    ctx->exact_mats.release();
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
    for (hipStream_t& st : ctx->aux_stream)
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
    for (hipEvent_t& e : ctx->job_ev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t& e : ctx->exact_prof_ev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t& e : ctx->screen_prof_ev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (ctx->screen_ev) { (void)hipEventDestroy(ctx->screen_ev); ctx->screen_ev = nullptr; }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CCAL_OK;
}

extern "C" int ccal_device_memory(ccal_ctx* ctx, size_t* free_bytes, size_t* total_bytes)
{
    if (!ctx) return CCAL_ERR_ARG;
    if (!free_bytes || !total_bytes) return fail(ctx, CCAL_ERR_ARG, "device_memory: null argument");
    if (int rc = bind(ctx)) return rc;
    CCAL_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
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
hipError_t ccal::download_array(ccal_ctx* ctx, void* dst, const void* src, size_t bytes)
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
    std::vector<float> radius_sums(4 * rblock_off[n_clouds], 0.f);
    DevBuf radius_sums_dev;   // per tiling block: sums of its points' distances from the centre and of |dx|, |dy|, |dz|
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
                                   radius_sums_dev.as<float>() + 4 * rblock_off[j]);
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
    cs.radius_axis.assign(3 * (size_t)n_clouds, 0.0);
    cs.rot_norm.assign(n_clouds, 0.0);
    for (int j = 0; j < n_clouds; ++j) {
        double sum[4] = {0.0, 0.0, 0.0, 0.0};
        for (size_t b = rblock_off[j]; b < rblock_off[j + 1]; ++b)
            for (int k = 0; k < 4; ++k) sum[k] += (double)radius_sums[4 * b + k];
        auto mean_of = [&](double v) {
            return std::isfinite(cs.radius[j]) && std::isfinite(v) ? std::min(cs.radius[j], v / (double)n_pts[j] * (1.0 + 1e-5))
                                                                    : std::numeric_limits<double>::infinity();
        };
        cs.radius_mean[j] = mean_of(sum[0]);
        for (int k = 0; k < 3; ++k) cs.radius_axis[3 * (size_t)j + k] = mean_of(sum[1 + k]);
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
    {
        std::vector<double> cb(kCloudBound * (size_t)n_clouds);
        for (int j = 0; j < n_clouds; ++j) {
            double* b = &cb[kCloudBound * (size_t)j];
            b[0] = cs.radius_mean[j];
            b[1] = cs.rot_norm[j] * (1.0 + 1e-9);
            for (int k = 0; k < 3; ++k) b[2 + k] = cs.radius_axis[3 * (size_t)j + k];
        }
        CCAL_HIP(ctx, cs.bound_dev.ensure(cb.size() * sizeof(double)));
        CCAL_HIP(ctx, hipMemcpy(cs.bound_dev.p, cb.data(), cb.size() * sizeof(double), hipMemcpyHostToDevice));
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
    if (int rc = presize_request_buffers(ctx, cs)) return rc;
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

