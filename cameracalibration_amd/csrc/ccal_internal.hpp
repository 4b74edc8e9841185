// ccal_internal.hpp -- context layout shared by the translation units of libccal.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ccal.h"

namespace ccal {

struct BlockDesc;

// Grow-only device allocation.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        // round up so repeated slightly larger requests do not re-allocate
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Grow-only pinned host allocation; the pointer is valid in kernels too (small tables the
// device reads once and result words it writes once, without a copy call on either side).
struct HostBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Large uploads: every copying host thread owns a lane (a pinned double buffer and the events that say
// when a buffer's copy has left it); all lanes feed one copy stream; one event per batch of clouds
// tells the context's stream that the batch has arrived.
struct StageLane {
    HostBuf buf[2];
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool pending[2] = {false, false};   // a copy out of buf[k] was enqueued and not yet waited for
};

struct Staging {
    hipStream_t stream = nullptr;
    hipEvent_t batch_ev[8] = {};
    std::vector<StageLane> lanes;
};

struct CloudSet {
    bool ready = false;
    int n_clouds = 0;
    std::vector<size_t> n_pts;
    std::vector<double> base_T_flange;  // n_clouds * 16
    size_t total_pts = 0;

    // exact path: SoA coordinates of all clouds back to back + block table
    DevBuf xs, ys, zs, blocks, cloud_block_off_dev, cloud_pt_off_dev;
    // chains cut into parts: per cloud the offset of its 512-element segments in the record arrays, and
    // the running sums at the segment starts left behind by the last split evaluation (zero: none yet)
    DevBuf cloud_seg_off_dev, seg_sums;
    int64_t total_segs = 0;
    bool seg_sums_valid = false;        // some evaluation since the upload has written seg_sums
    std::vector<double> seg_ref_totals; // per cloud: the total of the chain that wrote seg_sums (host copy)
    std::vector<int> cloud_block_off;  // n_clouds + 1
    int n_blocks = 0;

    // fast path: centred 64-point groups
    DevBuf groups, group_off_dev, scale_dev, centres_dev, base_dev;
    DevBuf split_groups;               // bf16x3 K-slot form of `groups` for the wide kernel
    bool split_ready = false;          // built on first selection of that kernel
    std::vector<int> group_off;        // n_clouds + 1
    std::vector<double> centres;       // n_clouds * 3
    DevBuf bound_dev;                  // per cloud, kCloudBound doubles: radius_mean, rot_norm (1 + 1e-9), radius_axis[3]: the screening budget's inputs
    DevBuf radius_dev;                 // per cloud: bits of the largest squared distance from the centre
    std::vector<double> radius;        // per cloud: that distance (infinity if a point is not finite)
    std::vector<double> radius_mean;   // per cloud: mean distance from the centre (same convention)
    std::vector<double> radius_axis;   // per cloud: mean |x - cx|, |y - cy|, |z - cz| (same convention)
    std::vector<double> rot_norm;      // per cloud: bound on the 2-norm of the robot pose's 3x3 part
    int64_t n_groups = 0;
    int64_t n_pad_points = 0;          // zero points added to fill the last groups
    int min_groups = 0;

    void release()
    {
        xs.release(); ys.release(); zs.release(); blocks.release();
        cloud_block_off_dev.release(); cloud_pt_off_dev.release(); cloud_seg_off_dev.release(); seg_sums.release(); groups.release(); group_off_dev.release();
        scale_dev.release(); centres_dev.release(); base_dev.release(); split_groups.release(); radius_dev.release(); bound_dev.release();
        ready = false;
    }
};

} // namespace ccal

struct ccal_ctx {
    int device = -1;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    ccal::CloudSet sets[2];
    int fast_kernel = 0;  // CCAL_FAST_BF16X3
    int search_mode = 0;  // CCAL_SEARCH_DENSE
    bool plane_set = false;
    float plane[4] = {0, 0, 0, 0};
    std::string err;

    // scratch (grow-only)
    ccal::DevBuf w_buf, partial_buf, cost_buf, rec_buf, minrec_buf, list_buf, count_buf;
    ccal::DevBuf sep_wr, sep_sb, sep_perm, sep_tab, sep_meta, sep_wcnt, sep_wsum;
    ccal::DevBuf vox[13];                // scratch of the device voxel grid
    ccal::HostBuf exact_in, exact_out;   // pose records + planes, and per-chain results of the exact objective
    ccal::DevBuf exact_mats;             // [pose][cloud][12]: the rows of base_T_flange_j * F(pose), built on the device
    ccal::DevBuf split_maps, split_first;  // segment records and part-0 sums of the split evaluation
    // One exact request in flight (ccal::exact_begin ... exact_wait_rest): the poses may be cut into a first
    // and a second part whose chain kernels follow each other on the context's stream while the
    // finishing pass of the first part runs beside the second part's chains on a stream of its own.
    struct ExactJob {
        bool active = false, split = false, want_max = false;
        bool watch_words = false;                  // the wait watches the result words instead of the event
        int which = 0, nb = 0, n_first = 0;        // n_first: poses of the first part (0: one part)
        double host0 = 0.0, host_ms = 0.0, wait_ms = 0.0;
    } exact_job;
    hipStream_t aux_stream[2] = {nullptr, nullptr};       // finishing passes of the two parts
    hipEvent_t job_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // chains of part 1 / part 2 done, results of part 1 / 2 in place
    hipEvent_t exact_prof_ev[3] = {nullptr, nullptr, nullptr};
    struct ScreenJob {
        bool active = false, watch_words = false;
        int which = 0, nb = 0;
        double host0 = 0.0, host_ms = 0.0;
    } screen_job;
    hipEvent_t screen_ev = nullptr;                        // results of the screening request in place
    hipEvent_t screen_prof_ev[2] = {nullptr, nullptr};
    bool gd_pipelining = false;          // a lone descent issues its next screening pass ahead of the curvature test (measured: no gain)
    bool chain_split = true;             // small exact requests cut every chain into parts
    int64_t split_resummed = 0, split_segments = 0;   // statistics of the split evaluations
    int split_skip = 0, split_penalty = 1;            // ccal_eval: calls to go without splitting after a bad one
    ccal::HostBuf screen_in, screen_out; // four-vectors and per-(pose, cloud) sums of the screening pass
    ccal::DevBuf screen_partial, screen_w;
    bool gd_screening = true;            // line-search trials are first screened on the fast path
    ccal::Staging staging;               // pinned staging of large uploads (made at the first one, kept)
    bool profiling = false;              // HIP events around the exact and screening kernels
    hipEvent_t prof_ev[3] = {nullptr, nullptr, nullptr};
    ccal_profile prof = {};
    double* static_scratch = nullptr;    // this device's address of the library's static scratch array
    ccal::HostBuf static_out;            // pinned result slot of ccal_fit_plane_lsq_static (made at create)
    ccal::DevBuf mats_buf, planes_buf, psum_buf, osum_buf, omax_buf, tbl_buf, dist_buf;
};

namespace ccal {

// ---- host helpers shared by the translation units (defined in api_core.hip unless noted) ----

int fail(ccal_ctx* ctx, int code, const std::string& msg);      // records msg as the context's last error

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

// Host wall clock in milliseconds (the host-side counters of ccal_profile).
inline double host_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

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

constexpr int kPointsPerBlock = 256 * 16;  // exact path: points handled by one block
constexpr int kNA = 8;                      // candidate tiles per wave on the fast path
constexpr int kCandPerBlock = 4 * kNA * 16; // 512
constexpr unsigned kBandCap = 1u << 16;     // near-minimum candidates re-scored per pass

// Every rounding budget of the fast paths (screening pass, grid search) is inflated by this factor.
constexpr double kScreenSlack = 1.25;

// ccal_fit_plane_lsq_static: clouds per set it accepts, and the doubles of its static scratch array
// (per-cloud pose rows, then the partial moments of the moment pass: kMomentGrid x kMoments).
constexpr int kStaticMaxClouds = 1024;
constexpr size_t kStaticScratchDoubles = (size_t)kStaticMaxClouds * 12 + (size_t)2048 * 10;

struct FastLaunch {
    int n_ctiles;
    int64_t n_padded;
    int64_t cand_blocks;
    int n_splits;
};


int bind(ccal_ctx* ctx);                        // makes the context's GPU current
int check_set(ccal_ctx* ctx, int which);        // the selector is valid and the set uploaded

// Rows 0..2 of base_T_flange_j * F(pose) for every cloud, [n_clouds][12].
void pose_matrices(const CloudSet& cs, const double pose[6], double* mats);
int choose_splits(const ccal_ctx* ctx, const CloudSet& cs, int64_t cand_blocks);
// per_block: candidates of one workgroup of the kernel the plan is for.
FastLaunch plan_fast(const ccal_ctx* ctx, const CloudSet& cs, int64_t n, int64_t per_block);
int64_t fast_chunk(const CloudSet& cs);
// bf16x3 K-slot form of the point groups for the wide residual kernel (api_search.hip).
int ensure_split_groups(ccal_ctx* ctx, CloudSet& cs);
// This device's address of the static scratch array of ccal_fit_plane_lsq_static (api_fit.hip).
hipError_t static_scratch_address(double** out);
// A result array of the context's stream into the caller's (pageable) memory, through the
// pinned staging lanes when it is large; returns with the bytes in place.
hipError_t download_array(ccal_ctx* ctx, void* dst, const void* src, size_t bytes);

// Makes the request buffers of the exact and screening paths for a freshly uploaded set (api_exact.hip).
int presize_request_buffers(ccal_ctx* ctx, const CloudSet& cs);

// allow_split: the poses are close to those of the previous such call (a descent), so a small request
// may cut its chains into parts whose units are guessed from that call's sums (same values either way).
int eval_exact(ccal_ctx* ctx, int which, int n_poses, const double* poses6, const double* planes4,
               double* cost, double* per_cloud_mean, double* per_cloud_max, bool allow_split = false,
               const double* expected_sums = nullptr);

// The same request in pieces (at most 4096 poses, one request in flight per context): exact_begin
// enqueues everything and returns; with 0 < n_first < n_poses the values of the first n_first poses
// can be collected (exact_wait_first) while the rest is still being computed, and work enqueued on the
// context's stream in between runs beside the last finishing pass.  exact_wait_rest collects the
// remaining values (all of them when the request was not cut) and ends the request.
// expected_sums ([pose][cloud], or null): what the caller expects the chains to sum to (the screening pass's
// sums of a nearby pose, say).  They only steer the guesses of a split request, never a value.
int exact_begin(ccal_ctx* ctx, int which, int n_poses, const double* poses6, const double* planes4, int n_first,
                bool want_max, bool allow_split, const double* expected_sums = nullptr);
int exact_wait_first(ccal_ctx* ctx, double* cost);
int exact_wait_rest(ccal_ctx* ctx, double* cost, double* per_cloud_mean, double* per_cloud_max);

// Folded FP32 objective of n poses (planes4: per-pose planes, or null for the context's plane)
// with a bound on its distance from eval_exact's value: |cost[i] - exact cost of pose i| <=
// bound[i] whenever both are finite (bound[i] is infinite if the set holds a non-finite point).
// per_cloud_sums ([pose][cloud], optional): the FP32 pass's sum of distances of every cloud.
int eval_screen(ccal_ctx* ctx, int which, int n_poses, const double* poses6, const double* planes4,
                double* cost, double* bound, double* per_cloud_sums = nullptr);
// The same in two steps (at most 1024 poses, one request in flight): enqueue on the context's stream,
// collect later.
int screen_begin(ccal_ctx* ctx, int which, int n_poses, const double* poses6, const double* planes4);
int screen_wait(ccal_ctx* ctx, double* cost, double* bound, double* per_cloud_sums = nullptr);

} // namespace ccal
