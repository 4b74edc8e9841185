/*
 * ccal.h -- C ABI of the MI355X hand-eye plane-calibration hot path.
 *
 * The reference (REXJJ/CameraCalibration) has no plugin or FFI layer: its hot
 * path is a set of C++ member functions on one global `Optimizer` object.  Each
 * entry point below replaces one of those call seams; the comment above it
 * names the reference lines it stands in for (paths relative to the reference
 * tree).  A host program written against the reference binds these instead of
 * the member functions.
 *
 * Conventions
 *   - every function returns an int status: CCAL_OK (0) or a negative code;
 *     no exception crosses the boundary; ccal_last_error() gives the text.
 *   - a context is bound to ONE GPU and used from one host thread at a time;
 *     run one context per GPU (one process per GPU, or one host thread each).
 *   - host pointers are borrowed only for the duration of a call.
 *   - poses are 6 doubles (tx,ty,tz [m], rz,ry,rx [rad]) as in the reference's
 *     `flange_transformation`; matrices are 4x4 row-major doubles.
 *   - the library never falls back to a CPU path: without a usable GPU every
 *     compute entry point fails with CCAL_ERR_DEVICE.
 */
#ifndef CCAL_H
#define CCAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    CCAL_OK            =  0,
    CCAL_ERR_ARG       = -1,  /* bad argument (null pointer, bad size/selector) */
    CCAL_ERR_DEVICE    = -2,  /* HIP error or no device */
    CCAL_ERR_STATE     = -3,  /* clouds / plane not set yet */
    CCAL_ERR_CAPACITY  = -4,  /* a bounded internal list overflowed */
    CCAL_ERR_IO        = -5   /* file could not be read / unsupported format */
};

/* which cloud set an operation uses: the reference keeps two per run,
 * `cloud_downsampled` (objective, camera_calibration_optimization.cpp:295) and
 * `clouds` (raw; plane fit :240-273 and the error report :389-404). */
enum { CCAL_DOWNSAMPLED = 0, CCAL_RAW = 1 };

typedef struct ccal_ctx ccal_ctx;

/* Library / context lifetime.  Replaces the global `Optimizer opti`
 * (camera_calibration_optimization.cpp:409). */
int  ccal_create(int device_id, ccal_ctx** out);
int  ccal_destroy(ccal_ctx* ctx);
const char* ccal_last_error(const ccal_ctx* ctx);
const char* ccal_status_string(int status);
/* Number of GPUs visible to the process (0 when there is none). */
int  ccal_device_count(void);

/* One-time device-resident upload of a cloud set.  Replaces the members
 * `clouds` / `cloud_downsampled` / `inverse_kinematics` / `mapping`
 * (camera_calibration_optimization.cpp:48-60, filled at :158-210).
 *   xyz[j]        n_pts[j] points, AoS float x,y,z (stride 3)
 *   base_T_flange n_clouds 4x4 matrices, ALREADY mapped: entry j is
 *                 inverse_kinematics[mapping[j]] (:290)
 * Empty clouds (n_pts[j]==0) are rejected: the reference divides by the count. */
int ccal_upload_clouds(ccal_ctx* ctx, int which, int n_clouds,
                       const float* const* xyz, const size_t* n_pts,
                       const double* base_T_flange);

/* Builds the DOWNSAMPLED set on the device from the resident RAW set: per cloud the
 * voxel-grid centroid filter that stands in for pcl::VoxelGrid at
 * camera_calibration_optimization.cpp:196-205 (same definition as ccal_voxel_grid, bit-equal
 * output).  n_pts_out: NULL or n_clouds entries.  The raw clouds are uploaded once; nothing but
 * the few hundred centroids per cloud crosses PCIe again. */
int ccal_downsample_raw(ccal_ctx* ctx, float leaf, size_t* n_pts_out);
/* Size / coordinates (AoS xyz, n*3 floats) of one resident cloud. */
int ccal_cloud_size(const ccal_ctx* ctx, int which, int cloud, size_t* n_out);
int ccal_download_cloud(ccal_ctx* ctx, int which, int cloud, float* xyz_out);

/* The reference plane `Optimizer::plane` (Eigen::Vector4f, :53, set at :274). */
int ccal_set_plane(ccal_ctx* ctx, const float plane[4]);
int ccal_get_plane(const ccal_ctx* ctx, float plane[4]);

/* Objective, reference arithmetic (FP64 transform, point rounded to float, FP64
 * distance).  Replaces
 *   double Optimizer::getError(vector<double>)                  :280-313
 *   double Optimizer::getError(vector<double>, vector<double>)  :314-347
 *   the per-cloud mean/max of Optimizer::printError             :348-407
 * for a batch of poses in one pass over the points.
 *   poses6   n_poses*6
 *   planes4  NULL -> every pose uses the context plane (float, as :121);
 *            else n_poses*4 doubles, one caller plane per pose (as :126)
 *   cost     n_poses        mean over clouds of per-cloud mean distance [m]
 *   per_cloud_mean/max      NULL or n_poses*n_clouds                    [m] */
int ccal_eval(ccal_ctx* ctx, int which, int n_poses, const double* poses6,
              const double* planes4, double* cost,
              double* per_cloud_mean, double* per_cloud_max);

/* Per-point distances and error classes for one pose (all points of the set, clouds back to
 * back in upload order).  Replaces the per-point loops of the raw-cloud error report
 * (printError, :379-405) and of the viewer's plane-mode colouring
 * (src/algorithms.cpp:360-406; thresholds 2.5 / 5 / 10 mm -> class 0 / 1 / 2, else 3).
 *   arithmetic  CCAL_POINT_OPTIMISER: the optimiser's (point rounded to float, :303-304)
 *               CCAL_POINT_VIEWER:    the viewer's (FP64 point, generic matrix product,
 *                                     src/algorithms.cpp:379-385, :255-258)
 *   plane4      NULL = context plane; else 4 doubles
 *   dist_out    NULL or total-points doubles [m];  class_out  NULL or total-points bytes */
enum { CCAL_POINT_OPTIMISER = 0, CCAL_POINT_VIEWER = 1 };
int ccal_point_errors(ccal_ctx* ctx, int which, const double pose[6], const double* plane4,
                      int arithmetic, double* dist_out, unsigned char* class_out);

/* The accumulation of :305 (`average += distance`, one rounding per add, in
 * order) for n non-negative doubles, computed on the GPU by the same kernel
 * ccal_eval uses: out = ((0 + v[0]) + v[1]) + ... bit for bit.  Exposed so the
 * parallel emulation of the sequential sum can be validated on arbitrary data. */
int ccal_ordered_sum(ccal_ctx* ctx, const double* values, size_t n, double* out);

/* Same objective for a large batch of poses on the fast FP32 path (folded
 * 4-vector, matrix-core K=4 contraction).  Costs agree with ccal_eval to
 * ~1e-6 relative; used where thousands..millions of poses are scored.
 * Uses the context plane. */
int ccal_eval_fast(ccal_ctx* ctx, int which, int64_t n_poses, const double* poses6,
                   float* cost);

/* The objective of :280-347 for a few poses on the FP32 path TOGETHER WITH A BOUND on its distance
 * from ccal_eval's value: whenever both are finite, |cost[i] - ccal_eval cost of pose i| <= bound[i]
 * (bound[i] is infinite if the set holds a non-finite point).  planes4: per-pose planes as in
 * ccal_eval, or NULL for the context plane.  This is what decides line-search trials in
 * ccal_gd_minimize without the reference arithmetic (see ccal_set_gd_screening); exposed so that the
 * bound can be checked against ccal_eval on arbitrary data. */
int ccal_eval_bounded(ccal_ctx* ctx, int which, int n_poses, const double* poses6,
                      const double* planes4, double* cost, double* bound);

/* Which kernel the fast path runs (no counterpart in the reference):
 *   CCAL_FAST_BF16X3 (default)  operands split exactly into three bf16 pieces,
 *                               contraction on the bf16 matrix pipe, FP32 accumulate;
 *                               16 candidate tiles per wave (blocks of 1024 candidates);
 *   CCAL_FAST_F32               FP32-input matrix instruction (fmaf-chain exact);
 *   CCAL_FAST_BF16X3_WIDE       the bf16x3 contraction as pairs of 32x32x16 matrix
 *                               instructions on point operands split at upload;
 *   CCAL_FAST_BF16X3_8          the default kernel with 8 candidate tiles per wave (three
 *                               waves per SIMD instead of two; a few per cent slower).
 * All meet the same 1e-5 cost tolerance; the final arg-min never depends on it
 * (near-minimum candidates are re-scored with ccal_eval arithmetic). */
enum { CCAL_FAST_BF16X3 = 0, CCAL_FAST_F32 = 1, CCAL_FAST_BF16X3_WIDE = 2, CCAL_FAST_BF16X3_8 = 3 };
int ccal_set_fast_kernel(ccal_ctx* ctx, int kind);

/* How ccal_grid_search scores the grid on the fast path (no counterpart in the reference;
 * the result is the same either way, because the near-minimum candidates are re-scored
 * with ccal_eval arithmetic):
 *   CCAL_SEARCH_DENSE (default)  every (candidate, point) residual is evaluated on the
 *                                matrix cores;
 *   CCAL_SEARCH_SEPARABLE        the grid is a translation x rotation product, and for a
 *                                fixed rotation the cost is a piecewise-linear function of
 *                                one scalar per translation: one histogram pass over the
 *                                points per rotation answers all its translations. */
enum { CCAL_SEARCH_DENSE = 0, CCAL_SEARCH_SEPARABLE = 1 };
int ccal_set_search_mode(ccal_ctx* ctx, int mode);

/* Candidate grid of discreteCombinatorialOptimization (:526-558), generalised:
 * axis k perturbs pose[k]; axes 0..2 in millimetres, 3..5 in degrees
 * (rad = deg*3.141592/180 as helpers.hpp:275-278); value_i = min + i*step;
 * linear candidate index is row-major with axis 5 fastest, which is the
 * reference's loop-nest order x,y,z,zt,yt,xt. */
typedef struct {
    double min[6];
    double step[6];
    int    count[6];
} ccal_grid;

/* The shipped 11^6 grid: +-20 mm step 4, +-5 deg step 1 (:532-538). */
void ccal_grid_default(ccal_grid* g);
int64_t ccal_grid_size(const ccal_grid* g);
void ccal_grid_pose(const ccal_grid* g, const double base_pose[6], int64_t index,
                    double pose[6]);

typedef struct {
    int64_t n_candidates;   /* scored by this call */
    int64_t n_rescored;     /* near-minimum candidates re-scored with ccal_eval arithmetic */
    double  residuals;      /* candidate x point evaluations on the fast path */
    double  kernel_ms;      /* HIP-event time of the residual kernels, summed */
    int     kernel_launches;
    double  total_ms;       /* HIP-event time of the whole call's device work */
    double  band_abs;       /* the error budget the re-score set was chosen with:            */
    double  band_rel;       /* |fast - ccal_eval cost| <= band_abs + band_rel * fast         */
} ccal_grid_stats;

/* Arg-min of the objective over grid candidates [begin,end) on this GPU.
 * Replaces the 6-deep loop nest :533-548.  The winner is the lowest-index
 * minimum of the reference-arithmetic cost (strict '<', :542).  The fast path
 * comes with a bound on its distance from the ccal_eval cost that holds for
 * every candidate of the grid (ccal_grid_error_bound); every candidate that the
 * bound cannot exclude -- fast (1 - rel) <= fast_min (1 + rel) + 2 abs -- is
 * re-scored with ccal_eval arithmetic before the final comparison, in passes of
 * 65536 when there are many (flat grids, scenes with points on the plane), so
 * the call never fails for the number of near-ties.  A NaN cost never wins
 * (:542 is false for it); a set holding non-finite points makes the bound
 * infinite and every candidate is re-scored.
 * Shards of one grid (one per GPU) are merged with ccal_argmin_merge or an
 * all-gather of (cost,index) pairs.  stats may be NULL. */
int ccal_grid_search(ccal_ctx* ctx, int which, const double base_pose[6],
                     const ccal_grid* g, int64_t begin, int64_t end,
                     int64_t* best_index, double* best_cost, double best_pose[6],
                     ccal_grid_stats* stats);

/* Fast-path cost of every candidate of [begin,end) (cost_out[i] = candidate begin + i): the
 * landscape ccal_grid_search minimises before its exact re-score, computed the way the
 * current search mode computes it.  The reference keeps only the running minimum of
 * :541-547; this exposes the whole sweep for inspection and tests. */
int ccal_grid_costs(ccal_ctx* ctx, int which, const double base_pose[6], const ccal_grid* g,
                    int64_t begin, int64_t end, float* cost_out);

/* The arg-min exchange of a one-process-per-GPU run over RCCL (the one collective of the path;
 * the reference is a single process and has none).  Every rank searches its own candidate
 * range with ccal_grid_search and then calls ccal_argmin_allgather with its (cost, index)
 * (index < 0: empty shard): one ncclAllGather of 16-byte records on the context's stream, and
 * every rank takes the lexicographic minimum -- lowest index among equal costs, as the single
 * scan of :542 keeps it.  Rendezvous: rank 0 calls ccal_comm_unique_id and hands the 128 bytes
 * to the other ranks by whatever channel the launcher has (a file, MPI, a socket); every rank
 * then calls ccal_comm_create (collective).  librccl is loaded at the first of these calls. */
#define CCAL_COMM_ID_BYTES 128
typedef struct ccal_comm ccal_comm;
int ccal_comm_unique_id(char id_out[CCAL_COMM_ID_BYTES]);
int ccal_comm_create(ccal_ctx* ctx, int world_size, int rank, const char id[CCAL_COMM_ID_BYTES],
                     ccal_comm** out);
int ccal_comm_destroy(ccal_comm* comm);
int ccal_argmin_allgather(ccal_ctx* ctx, ccal_comm* comm, double cost, int64_t index,
                          double* best_cost, int64_t* best_index, int* winner_rank);
/* A rank whose search failed passes this index so that its peers are not left waiting in the
 * exchange: every rank's ccal_argmin_allgather then returns CCAL_ERR_STATE. */
#define CCAL_SHARD_FAILED (-2)
/* Size and this process's rank of the communicator as RCCL reports them (ncclCommCount,
 * ncclCommUserRank): evidence of how many ranks the exchange really spans. */
int ccal_comm_info(const ccal_comm* comm, int* world_size, int* rank);

/* The bound ccal_grid_search works with, for the current search mode and fast kernel: for
 * every candidate c of the grid, |ccal_grid_costs value of c - ccal_eval cost of c| <=
 * *abs_out + *rel_out * (ccal_grid_costs value of c).  Derived on the host from the cloud
 * centres and mean radii, the robot matrices, the plane and the extent of the grid box; infinite
 * when the set holds a non-finite point.  (The reference has one arithmetic and needs no such
 * bound; this is what makes the index of :542 identical although the sweep runs in FP32.) */
int ccal_grid_error_bound(ccal_ctx* ctx, int which, const double base_pose[6], const ccal_grid* g,
                          double* abs_out, double* rel_out);

/* Free and total device memory of the context's GPU in bytes (hipMemGetInfo): a diagnostic, e.g.
 * to check that a call allocated nothing. */
int ccal_device_memory(ccal_ctx* ctx, size_t* free_bytes, size_t* total_bytes);

/* Lexicographic (cost, index) minimum over n shard results; entries with
 * index < 0 (empty shards) are skipped.  Returns the winning slot or -1. */
int ccal_argmin_merge(int n, const double* costs, const int64_t* indices);

/* Least-squares plane of every point of the set carried to the base frame by
 * `pose`.  Replaces fitPlane (:100-119) applied to the cloud assembled at
 * :240-272: solves a*x + b*y + c = -z, returns (a,b,1,c)/|(a,b,1)| as floats. */
int ccal_fit_plane_lsq(ccal_ctx* ctx, int which, const double pose[6], float plane_out[4]);

/* ccal_fit_plane_lsq for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, code that must not disturb an allocator).  No counterpart in
 * the reference.  Same inputs, same result bit for bit; the per-cloud matrices and the partial
 * moments of the pass's workgroups live in a static device array of the library (256 KiB per
 * GPU, held until the process ends) instead of grow-only buffers, and the call issues kernel
 * launches and one stream wait only -- no hipMalloc / hipFree / copy call.  It does wait for
 * the stream and read the result on the host, so it cannot itself run under stream capture.
 * Limit: at most 1024 clouds in the set (any number of points), else CCAL_ERR_CAPACITY.  All
 * contexts of one GPU share that array: calls that overlap on one GPU race. */
int ccal_fit_plane_lsq_static(ccal_ctx* ctx, int which, const double pose[6], float plane_out[4]);

/* RANSAC plane with the parameters of getPlane (cloud_analyzer.cpp:103-121:
 * SACMODEL_PLANE, SAC_RANSAC, distance threshold, optimise coefficients):
 * n_hyp three-point hypotheses drawn with a counter-based generator from
 * `seed`, every hypothesis scored against every point of the set in one batched
 * pass, best hypothesis (most inliers under the reference-arithmetic distance, lowest
 * index on ties) refitted on its inliers (total least squares).
 * plane_out is (n, d) with |n| = 1 and n.z >= 0; inliers_out / best_hyp_out (either may be
 * NULL) receive the winner's inlier count and its index in [0, n_hyp). */
int ccal_ransac_plane(ccal_ctx* ctx, int which, const double pose[6], int n_hyp,
                      float threshold, uint64_t seed, float plane_out[4],
                      int64_t* inliers_out, int* best_hyp_out);

/* Gradient-descent refinement with the reference's gdcpp settings.  Replaces
 * gradientDescent (:411-465) for dim 6 and gradientDescentWithPlane (:466-524)
 * for dim 10 (x = pose ++ plane).  Every objective value is ccal_eval
 * arithmetic; each finite-difference stencil is one batched device pass. */
typedef struct {
    int    max_iterations;   /* 10000  (:431) */
    double min_gradient_len; /* 1e-6   (:432) */
    double min_step_len;     /* 1e-6   (:433) */
    double momentum;         /* 0.4    (:434) */
    double fd_eps;           /* central-difference step, sqrt(DBL_EPSILON) */
    double wolfe_decrease;   /* 0.8 */
    double wolfe_c1;         /* 1e-4 */
    double wolfe_c2;         /* 0.9 */
    double wolfe_min_step;   /* 1e-10 */
    double wolfe_max_step;   /* 1.0 */
    int    wolfe_max_it;     /* 0 = unlimited */
} ccal_gd_opts;

typedef struct {
    int     iterations;
    int     converged;
    double  fval;
    double  xval[10];
    int64_t n_evals;         /* objective evaluations of the sequential algorithm: 1 + 2*dim per
                              * iterate and per line-search trial it visits (the library evaluates
                              * fewer: gradients of trials that fail Armijo are never needed) */
} ccal_gd_result;

void ccal_gd_default_opts(ccal_gd_opts* o);
int  ccal_gd_minimize(ccal_ctx* ctx, int which, int dim, const double* x0,
                      const ccal_gd_opts* opts, ccal_gd_result* result);
/* n_starts independent descents advanced in lock step, their stencils batched
 * into shared device passes.  x0 is n_starts*dim, results has n_starts slots. */
int  ccal_gd_minimize_multi(ccal_ctx* ctx, int which, int dim, int n_starts,
                            const double* x0, const ccal_gd_opts* opts,
                            ccal_gd_result* results);
/* Line-search screening (on by default; on != 0 enables).  The backtracking trials of
 * WolfeBacktracking (gdcpp, instantiated at :428 and :486) mostly fail the Armijo test by a wide
 * margin: their centres are first scored by the FP32 fast path, which comes with a bound on its
 * distance from the reference-arithmetic value; a trial is decided from that score only when the
 * bound leaves no doubt, every other trial is evaluated with ccal_eval arithmetic.  Iterates,
 * counts and values are the same with screening on or off; off evaluates every trial centre
 * with the reference arithmetic. */
int  ccal_set_gd_screening(ccal_ctx* ctx, int on);
/* Chain splitting (on by default; on != 0 enables).  A descent's requests are a few hundred
 * (pose, cloud) sums of :305, too few to fill the GPU with one workgroup each: every sum is cut into
 * parts that are folded concurrently in the rounding unit the previous evaluation's running sum
 * suggests, and a short pass applies them to the true running sum, summing again whatever the
 * guess did not fit.  Same values on or off (the ordered sum of :305 either way). */
int  ccal_set_chain_split(ccal_ctx* ctx, int on);
/* Look-ahead of a lone descent (off by default; on != 0 enables).  The request that settles a line
 * search -- the undecided trial centres, the expected trial's gradient points, the next iterate's
 * stencil -- is computed in two parts, and as soon as the centres and the stencil are known the next
 * iteration's screening pass is issued on the assumption that the trial's curvature test (:433's
 * WolfeBacktracking) will pass; it runs on the device beside the tail of the current request.  The
 * test is then made with the real gradient, and the early screening result is used only if the
 * descent is where the look-ahead assumed.  Same iterates, counts and values on or off.  Measured on
 * MI355X it does not pay: what the overlap saves (the screening pass and its host round trip, about
 * 50 us per iteration) the two chain launches of a request cut in two lose again (two launch tails
 * and the finishing waves sharing the chip with the second launch), so it stays off. */
int  ccal_set_gd_pipelining(ccal_ctx* ctx, int on);

/* Device-time counters of the exact and screening passes, the plane fits and the point report (off
 * by default; HIP events on the context's stream around the kernels).  The reference times host code with its TIC/TOC macros
 * (src/helpers.hpp:222-234); this is their device-side counterpart and feeds the roofline
 * figures of the benchmark.  pose_points = sum over requests of poses x points of the set. */
typedef struct {
    double  split_chain_ms;      /* split_chain_kernel: chains cut into concurrently folded parts */
    double  split_finish_ms;     /* split_finish_kernel: one wave per chain applies the parts     */
    double  split_pose_points;
    int64_t split_requests;
    double  whole_chain_ms;      /* exact_chain_kernel: one workgroup per (pose, cloud) chain     */
    double  whole_pose_points;
    int64_t whole_requests;
    double  screen_ms;           /* screen_cost_kernel + its reduction (FP32 line-search scores)  */
    double  screen_pose_points;
    int64_t screen_requests;
    /* host side of the same requests (wall clock): before the stream wait (tables, launches), inside
     * it, and after it (results) */
    double  exact_host_ms, exact_wait_ms;
    double  screen_host_ms, screen_wait_ms;
    /* ccal_ransac_plane: the FP32 bracket pass over every (hypothesis, point) pair, and the exact
     * recounts (hypotheses re-counted, device time of those passes) */
    double  ransac_score_ms;
    double  ransac_tests;        /* hypotheses x points (padding included) of the bracket passes   */
    int64_t ransac_calls;
    double  ransac_recount_ms;
    int64_t ransac_recounted;
    /* moments_kernel of ccal_fit_plane_lsq (+ its finalisation): points x 12 bytes is its traffic  */
    double  moments_ms;
    double  moments_points;
    int64_t moments_calls;
    /* point_error_kernel of ccal_point_errors (kernel only, without the copy to the host)         */
    double  point_error_ms;
    double  point_error_points;
} ccal_profile;
int ccal_set_profiling(ccal_ctx* ctx, int on);
/* Copies the counters; reset != 0 zeroes them afterwards. */
int ccal_get_profile(ccal_ctx* ctx, ccal_profile* out, int reset);

/* ---- host-side ingest and the whole run -------------------------------------
 * These three do not touch the GPU. */

/* Cloud file -> malloc'ed AoS xyz (release with ccal_free).  Replaces
 * InputUtilities::readPointCloud (helpers.cpp:108-207): format by extension
 * (.pcd / .ply / .xyz), coordinates divided by 1000 ("mm") or 100 ("cm"). */
int  ccal_read_point_cloud(const char* filename, const char* metric, float** xyz_out,
                           size_t* n_out);
/* Voxel-grid centroid filter standing in for pcl::VoxelGrid at
 * camera_calibration_optimization.cpp:196-205 (leaf 0.02f).  xyz_out must hold
 * n points; n_out receives the number of centroids. */
int  ccal_voxel_grid(const float* xyz, size_t n, float leaf, float* xyz_out, size_t* n_out);
void ccal_free(void* p);
/* The depth cut of camera_calibration_optimization.cpp:167-169 (`pts.z<=0.8`): keeps the points
 * with (double)z <= z_max, in order; NaN z is dropped.  xyz_out must hold n points (it may be
 * xyz itself). */
int  ccal_filter_z(const float* xyz, size_t n, double z_max, float* xyz_out, size_t* n_out);

/* The pose parser of getTransVector (camera_calibration_optimization.cpp:70-98; the rows of
 * readTransformations, helpers.cpp:244-251, go through the same routine with angle = radians):
 * six comma-separated fields read with float precision (std::stof), the first three divided by
 * 1000 ("mm") or 100 ("cm"); with angle == "degree" the last three go through
 * degreeToRadian(int) (helpers.hpp:275-278: the value is TRUNCATED to whole degrees and pi is
 * 3.141592); any other angle string means radians.  CCAL_ERR_IO when fewer than six fields parse. */
int  ccal_parse_pose(const char* value_csv, const char* metric, const char* angle, double pose[6]);

/* Writes a copy of `config_in` whose approximate_transformation holds `pose` (metres,
 * radians): the "output new config file after running the optimization" item of the
 * reference's TODO.md:5.  ccal_run_config calls it when the config has an
 * <output><config>path</config></output> node. */
int  ccal_write_config(const char* config_in, const char* config_out, const double pose[6]);

/* The optimiser executable as one call: replaces main() + getInputs() + the
 * three drivers (camera_calibration_optimization.cpp:560-581, :148-278,
 * :526-558, :411-524).  Reads the reference's XML config, appends to the two
 * text files in the reference's format.  On failure returns a negative status
 * and copies a message into errbuf: CCAL_ERR_IO for anything that went wrong
 * reading or writing files (the reference's bare `throw;`, helpers.cpp:126,205),
 * CCAL_ERR_DEVICE for a missing GPU or a HIP / RCCL failure, CCAL_ERR_ARG for
 * a config that asks for something impossible.
 * Optional config nodes beside the reference's schema: <search> (grid extent,
 * mode, iteration cap), <device><gpus>N</gpus> (contexts of this process, one
 * host thread each; capped at the GPUs present unless <share>1</share> allows
 * several contexts per GPU), <output><config>.
 * One process per GPU: with CCAL_WORLD_SIZE / CCAL_RANK / CCAL_LOCAL_RANK /
 * CCAL_COMM_FILE in the environment every process loads the inputs, searches its
 * contiguous share of the grid and joins one ccal_argmin_allgather; rank 0 then
 * refines and writes the files, the others return after the exchange. */
int  ccal_run_config(const char* config_path, const char* results_path, const char* errors_path,
                     int verbose, char* errbuf, size_t errbuf_len);

/* Pose math shared with callers (helpers.cpp:45-71): ZYX pose -> 4x4. */
void ccal_pose_to_matrix(const double pose[6], double T[16]);

#ifdef __cplusplus
}
#endif
#endif /* CCAL_H */
