/*
 * ccal_oracle.h -- CPU restatement of the reference optimiser's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library, the CLI or the
 * Python package may include, link or call this.  Only tests/, the smoke check
 * and the cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Parity status: "parity unpinned".  The reference has no tests or golden
 * vectors and cannot be built here (Eigen, PCL, Boost.PropertyTree and gdcpp.h
 * are absent), so every function below is a line-by-line restatement of the
 * cited reference lines, and the arithmetic that lives in third-party code
 * (Eigen products, PCL VoxelGrid, gdcpp) is restated from its published
 * algorithm with the summation orders written out in ccal_oracle.c.
 *
 * All matrices are 4x4 row-major doubles.  Clouds are AoS float xyz, stride 3.
 */
#ifndef CCAL_ORACLE_H
#define CCAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/helpers.hpp:275-278  degreeToRadian(int) */
double orc_degree_to_radian(int angle);

/* src/helpers.cpp:45-71  eul2rot (ZYX) + vectorToTransformationMatrix */
void orc_pose_to_matrix(const double pose[6], double T[16]);

/* 4x4 dynamic MatrixXd product as in camera_calibration_optimization.cpp:290 */
void orc_matmul4(const double A[16], const double B[16], double C[16]);

/* src/helpers.hpp:139-151 (SSE2 Transformer<double>::se3), via :192-196 */
void orc_transform_point(const double M[16], const float src[3], float out[3]);

/* src/camera_calibration_optimization.cpp:121-124 (Vector4f plane) */
double orc_distance_planef(const float plane[4], const float q[3]);
/* src/camera_calibration_optimization.cpp:126-129 (vector<double> plane) */
double orc_distance_planed(const double plane[4], const float q[3]);

/* A set of clouds plus the per-cloud base_T_flange already mapped through
 * mapping[j] (camera_calibration_optimization.cpp:161-163,290). */
typedef struct {
    int n_clouds;
    const float* const* xyz;   /* n_clouds pointers, each n_pts[j]*3 floats */
    const size_t* n_pts;
    const double* base_T_flange; /* n_clouds*16 */
} orc_scene;

/* :280-313  getError(pose) with the float plane */
double orc_get_error(const orc_scene* s, const double pose[6], const float plane[4]);
/* :314-347  getError(pose, plane) with a double plane */
double orc_get_error_plane(const orc_scene* s, const double pose[6], const double plane[4]);
/* :348-407  printError: per-cloud mean and max distance (metres, not yet x1000) */
void orc_per_cloud_error(const orc_scene* s, const double pose[6], const float plane[4],
                         double* mean_out, double* max_out);

/* Same as orc_get_error but with the base-frame matrices M_j = B_j*F given
 * directly (used to time/verify the inner loop in isolation). */
double orc_get_error_matrices(const orc_scene* s, const double* M /* n_clouds*16 */,
                              const float plane[4]);

/* :526-558 generalised: axis k perturbs pose[k]; axes 0..2 in millimetres,
 * axes 3..5 in degrees; value = min + i*step; linear index row-major with
 * axis 5 fastest.  The shipped grid is min=-20,step=4,count=11 (mm) and
 * min=-5,step=1,count=11 (deg). */
typedef struct {
    double min[6];
    double step[6];
    int    count[6];
} orc_grid;

void orc_grid_default(orc_grid* g);
int64_t orc_grid_size(const orc_grid* g);
void orc_grid_pose(const orc_grid* g, const double base[6], int64_t index, double pose[6]);

/* Scans [begin,end) in index order with strict '<' from error_min = 1e9.
 * costs_out may be NULL; otherwise receives end-begin costs. */
void orc_grid_search(const orc_scene* s, const float plane[4], const double base[6],
                     const orc_grid* g, int64_t begin, int64_t end,
                     int64_t* best_index, double* best_cost, double best_pose[6],
                     double* costs_out);

/* :100-119 fitPlane: least squares a*x + b*y + c = -z, returns float[4] */
void orc_fit_plane(const float* xyz, size_t n, float plane[4]);

/* :240-277 plane-fit input: every point through B_j*F(pose), rounded to float.
 * out must hold sum(n_pts)*3 floats; returns the count written. */
size_t orc_combined_cloud(const orc_scene* s, const double pose[6], float* out);

/* PCL VoxelGrid<PointXYZ> stand-in used by :196-205 (leaf 0.02f).  Deterministic:
 * voxels in ascending linear index, points inside a voxel in input order,
 * float accumulation.  out must hold n*3 floats; returns centroids written. */
size_t orc_voxel_grid(const float* xyz, size_t n, float leaf, float* out);

/* gdcpp-shaped gradient descent (:411-524).  dim = 6 (pose) or 10 (pose+plane). */
typedef struct {
    int    max_iterations;   /* 10000 */
    double min_gradient_len; /* 1e-6  */
    double min_step_len;     /* 1e-6  */
    double momentum;         /* 0.4   */
    double fd_eps;           /* sqrt(DBL_EPSILON) */
    double wolfe_decrease;   /* 0.8   */
    double wolfe_c1;         /* 1e-4  */
    double wolfe_c2;         /* 0.9   */
    double wolfe_min_step;   /* 1e-10 */
    double wolfe_max_step;   /* 1.0   */
    int    wolfe_max_it;     /* 0 = unlimited */
} orc_gd_opts;

typedef struct {
    int    iterations;
    int    converged;
    double fval;
    double xval[10];
    int64_t n_evals;         /* objective evaluations performed */
} orc_gd_result;

void orc_gd_default_opts(orc_gd_opts* o);
void orc_gd_minimize(const orc_scene* s, const float plane[4], int dim, const double* x0,
                     const orc_gd_opts* o, orc_gd_result* r);

/* CPU statement of the library's RANSAC scheme (three-point hypotheses from
 * splitmix64(seed+3h+k), exact-arithmetic inlier count, lowest index on ties,
 * covariance refit).  Not a reference restatement: see ccal_oracle.c. */
void orc_ransac_plane(const orc_scene* s, const double pose[6], int n_hyp, float threshold,
                      uint64_t seed, double plane_out[5], int64_t* inliers_out, int* best_hyp_out,
                      double moments_out[10], double org_out[3]);
/* Plane (a, b, c, d, |n|) of hypothesis h alone; 0 when its sample is collinear. */
int orc_ransac_hypothesis_plane(const orc_scene* s, const double pose[6], uint64_t seed, int h, double pl[5]);
/* Inlier counts of the listed hypotheses over every point of the set, the point loop spread over
 * host threads (counts are integers: the result does not depend on the thread count).
 * counts_out[k] = -1 for a collinear sample.  (ccal_oracle_omp.c) */
void orc_ransac_counts(const orc_scene* s, const double pose[6], uint64_t seed, float threshold,
                       const int* hyp, int n, int64_t* counts_out);

/* Per-point distances / classes: optimiser arithmetic (0, :389-404) or the viewer's
 * plane-mode colouring arithmetic (1, src/algorithms.cpp:379-402). */
void orc_point_errors(const orc_scene* s, const double pose[6], const double plane[4],
                      int arithmetic, double* dist_out, unsigned char* class_out);

/* stof-based parsers: camera_calibration_optimization.cpp:70-98, helpers.cpp:238-259 */
int orc_parse_pose_csv(const char* line, double scale, int angle_is_degree, double pose[6]);
int orc_file_id(const char* filename); /* :131-146 */

#ifdef __cplusplus
}
#endif
#endif
