/*
 * ccal_oracle.c -- CPU restatement of the reference optimiser's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ccal_oracle.h).  Parity unpinned: the
 * reference ships no golden vectors and cannot be built in this image.
 *
 * Build with:  gcc -O2 -ffp-contract=off -fno-fast-math  (x86-64 baseline = SSE2,
 * no FMA), which is how the reference's -O3 build evaluates the same lines.
 *
 * Citations are relative to the reference tree (src/...).
 */
#define _GNU_SOURCE /* sincos() */
#include "ccal_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------ */
/* pose math                                                           */
/* ------------------------------------------------------------------ */

/* src/helpers.hpp:275-278: constexpr double degreeToRadian(int angle)
 * { return angle*3.141592/180; }  (int argument: callers' doubles truncate) */
double orc_degree_to_radian(int angle)
{
    return angle * 3.141592 / 180;
}

/* Eigen::AngleAxisd(angle, unit axis).toRotationMatrix() for axis = e_k.
 * Eigen 3.3 AngleAxis.h: off-diagonals are tmp -/+ sin*axis with tmp = 0 for a
 * unit axis, the diagonal is cos1_axis*axis + c, i.e. c off-axis and
 * (1-c)*1 + c on the axis (not forced to exactly 1). */
static void axis_rotation(int k, double angle, double R[9])
{
    /* Eigen takes sin and cos of the same angle; gcc turns such a pair into one sincos() call at
     * any optimisation level, and glibc's sincos() differs from sin()/cos() in the last place for
     * about one argument in a thousand.  Called explicitly so that the result does not hang on
     * the optimiser. */
    double s, c;
    sincos(angle, &s, &c);
    double on_axis = (1.0 - c) * 1.0 + c;
    for (int i = 0; i < 9; ++i) R[i] = 0.0;
    if (k == 2) {            /* Z */
        R[0] = c;  R[1] = -s;
        R[3] = s;  R[4] = c;
        R[8] = on_axis;
    } else if (k == 1) {     /* Y */
        R[0] = c;  R[2] = s;
        R[4] = on_axis;
        R[6] = -s; R[8] = c;
    } else {                 /* X */
        R[0] = on_axis;
        R[4] = c;  R[5] = -s;
        R[7] = s;  R[8] = c;
    }
}

/* Fixed-size Matrix3d product, coefficient (i,j) = a0 + (a1 + a2)
 * (Eigen's unrolled non-vectorised reduction splits the range in halves). */
static void matmul3(const double A[9], const double B[9], double C[9])
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a0 = A[i * 3 + 0] * B[0 * 3 + j];
            double a1 = A[i * 3 + 1] * B[1 * 3 + j];
            double a2 = A[i * 3 + 2] * B[2 * 3 + j];
            T[i * 3 + j] = a0 + (a1 + a2);
        }
    memcpy(C, T, sizeof T);
}

/* src/helpers.cpp:45-59 eul2rot with seq "ZYX": rot = ((I*Rz(v3))*Ry(v4))*Rx(v5);
 * src/helpers.cpp:61-71 vectorToTransformationMatrix: t = v[0..2]. */
void orc_pose_to_matrix(const double pose[6], double T[16])
{
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double A[9];
    axis_rotation(2, pose[3], A); matmul3(R, A, R);
    axis_rotation(1, pose[4], A); matmul3(R, A, R);
    axis_rotation(0, pose[5], A); matmul3(R, A, R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = pose[i];
    }
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
}

/* Dynamic 4x4 MatrixXd product (camera_calibration_optimization.cpp:290):
 * Eigen's small-size lazy product accumulates res = 0; res = a_k*b_k + res for
 * k = 0..3 in order. */
void orc_matmul4(const double A[16], const double B[16], double C[16])
{
    double T[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double r = 0.0;
            for (int k = 0; k < 4; ++k) r = A[i * 4 + k] * B[k * 4 + j] + r;
            T[i * 4 + j] = r;
        }
    memcpy(C, T, sizeof T);
}

/* src/helpers.hpp:139-151: p = col3; for i in 0..2: p = v_i*col_i + p; cvt to f32.
 * (x86-64 -O3 without -march selects the SSE2 specialisation.) */
void orc_transform_point(const double M[16], const float src[3], float out[3])
{
    for (int r = 0; r < 3; ++r) {
        double p = M[r * 4 + 3];
        for (int i = 0; i < 3; ++i) {
            double v = (double)src[i];
            p = v * M[r * 4 + i] + p;
        }
        out[r] = (float)p;
    }
}

/* :121-124.  plane(k) is float, pt[k] is double (widened from float out[]). */
double orc_distance_planef(const float plane[4], const float q[3])
{
    double x = q[0], y = q[1], z = q[2];
    double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    return fabs(a * x + b * y + c * z + d) / (sqrt(pow(a, 2) + pow(b, 2) + pow(c, 2)));
}

/* :126-129 */
double orc_distance_planed(const double plane[4], const float q[3])
{
    double x = q[0], y = q[1], z = q[2];
    return fabs(plane[0] * x + plane[1] * y + plane[2] * z + plane[3]) /
           (sqrt(pow(plane[0], 2) + pow(plane[1], 2) + pow(plane[2], 2)));
}

/* ------------------------------------------------------------------ */
/* objective                                                           */
/* ------------------------------------------------------------------ */

static void cloud_matrix(const orc_scene* s, int j, const double pose[6], double M[16])
{
    double F[16];
    orc_pose_to_matrix(pose, F);                    /* :289 */
    orc_matmul4(s->base_T_flange + 16 * j, F, M);   /* :290 */
}

/* :280-313 */
double orc_get_error(const orc_scene* s, const double pose[6], const float plane[4])
{
    double error = 0.0;
    for (int j = 0; j < s->n_clouds; ++j) {
        double average = 0.0;
        double M[16];
        cloud_matrix(s, j, pose, M);
        const float* pts = s->xyz[j];
        size_t n = s->n_pts[j];
        for (size_t i = 0; i < n; ++i) {
            float out[3];
            orc_transform_point(M, pts + 3 * i, out);         /* :303 */
            double distance = orc_distance_planef(plane, out); /* :304 */
            average += distance;
        }
        error += average / n;                                   /* :309 */
    }
    return error / s->n_clouds;                                 /* :312 */
}

/* :314-347 */
double orc_get_error_plane(const orc_scene* s, const double pose[6], const double plane[4])
{
    double error = 0.0;
    for (int j = 0; j < s->n_clouds; ++j) {
        double average = 0.0;
        double M[16];
        cloud_matrix(s, j, pose, M);
        const float* pts = s->xyz[j];
        size_t n = s->n_pts[j];
        for (size_t i = 0; i < n; ++i) {
            float out[3];
            orc_transform_point(M, pts + 3 * i, out);
            double distance = orc_distance_planed(plane, out);
            average += distance;
        }
        error += average / n;
    }
    return error / s->n_clouds;
}

/* :348-407 (one pass; the caller runs it on the downsampled and the raw set) */
void orc_per_cloud_error(const orc_scene* s, const double pose[6], const float plane[4],
                         double* mean_out, double* max_out)
{
    for (int j = 0; j < s->n_clouds; ++j) {
        double average = 0.0;
        double error_mx = -1e9;
        double M[16];
        cloud_matrix(s, j, pose, M);
        const float* pts = s->xyz[j];
        size_t n = s->n_pts[j];
        for (size_t i = 0; i < n; ++i) {
            float out[3];
            orc_transform_point(M, pts + 3 * i, out);
            double distance = orc_distance_planef(plane, out);
            average += distance;
            if (distance > error_mx) error_mx = distance;
        }
        mean_out[j] = average / n;
        max_out[j] = error_mx;
    }
}

double orc_get_error_matrices(const orc_scene* s, const double* M, const float plane[4])
{
    double error = 0.0;
    for (int j = 0; j < s->n_clouds; ++j) {
        double average = 0.0;
        const float* pts = s->xyz[j];
        size_t n = s->n_pts[j];
        for (size_t i = 0; i < n; ++i) {
            float out[3];
            orc_transform_point(M + 16 * j, pts + 3 * i, out);
            average += orc_distance_planef(plane, out);
        }
        error += average / n;
    }
    return error / s->n_clouds;
}

/* ------------------------------------------------------------------ */
/* discrete grid search                                                */
/* ------------------------------------------------------------------ */

void orc_grid_default(orc_grid* g)
{
    /* :532  t_min=-20,t_max=20 step 4; r_min=-5,r_max=5 step 1 */
    for (int k = 0; k < 3; ++k) { g->min[k] = -20; g->step[k] = 4; g->count[k] = 11; }
    for (int k = 3; k < 6; ++k) { g->min[k] = -5;  g->step[k] = 1; g->count[k] = 11; }
}

int64_t orc_grid_size(const orc_grid* g)
{
    int64_t n = 1;
    for (int k = 0; k < 6; ++k) n *= g->count[k];
    return n;
}

/* :540.  Translations: base + v/1000.0.  Rotations: base + v*3.141592/180, which
 * equals degreeToRadian(int) for the integer-valued steps the reference uses. */
void orc_grid_pose(const orc_grid* g, const double base[6], int64_t index, double pose[6])
{
    int idx[6];
    for (int k = 5; k >= 0; --k) {
        idx[k] = (int)(index % g->count[k]);
        index /= g->count[k];
    }
    for (int k = 0; k < 6; ++k) {
        double v = g->min[k] + idx[k] * g->step[k];
        if (k < 3) pose[k] = base[k] + v / 1000.0;
        else       pose[k] = base[k] + v * 3.141592 / 180;
    }
}

/* :526-558 */
void orc_grid_search(const orc_scene* s, const float plane[4], const double base[6],
                     const orc_grid* g, int64_t begin, int64_t end,
                     int64_t* best_index, double* best_cost, double best_pose[6],
                     double* costs_out)
{
    double error_min = 1e9;                     /* :531 */
    int64_t best = -1;
    for (int k = 0; k < 6; ++k) best_pose[k] = base[k];
    for (int64_t c = begin; c < end; ++c) {
        double pose[6];
        orc_grid_pose(g, base, c, pose);
        double total_error = orc_get_error(s, pose, plane);
        if (costs_out) costs_out[c - begin] = total_error;
        if (total_error < error_min) {          /* :542 strict: first minimum wins */
            error_min = total_error;
            best = c;
            memcpy(best_pose, pose, 6 * sizeof(double));
        }
    }
    *best_index = best;
    *best_cost = error_min;
}

/* ------------------------------------------------------------------ */
/* plane fit                                                           */
/* ------------------------------------------------------------------ */

/* :100-119.  The reference solves min ||[x y 1](a b c)^T + z|| with JacobiSVD;
 * the minimiser is unique here, so it is obtained by Householder QR in FP64
 * (a different factorisation of the same N x 3 problem, agreeing with the SVD
 * solve far below the final cast to float). */
void orc_fit_plane(const float* xyz, size_t n, float plane[4])
{
    double* A = (double*)malloc(n * 4 * sizeof(double)); /* [x y 1 | -z] */
    for (size_t i = 0; i < n; ++i) {
        A[i * 4 + 0] = xyz[i * 3 + 0];
        A[i * 4 + 1] = xyz[i * 3 + 1];
        A[i * 4 + 2] = 1.0;
        A[i * 4 + 3] = -1.0 * xyz[i * 3 + 2];
    }
    for (int k = 0; k < 3; ++k) {
        double nrm = 0.0;
        for (size_t i = k; i < n; ++i) nrm += A[i * 4 + k] * A[i * 4 + k];
        nrm = sqrt(nrm);
        double alpha = A[k * 4 + k] > 0 ? -nrm : nrm;
        double v0 = A[k * 4 + k] - alpha;
        double vnorm2 = nrm * nrm - A[k * 4 + k] * A[k * 4 + k] + v0 * v0;
        A[k * 4 + k] = v0;
        if (vnorm2 > 0) {
            for (int c = k + 1; c < 4; ++c) {
                double dot = 0.0;
                for (size_t i = k; i < n; ++i) dot += A[i * 4 + k] * A[i * 4 + c];
                double f = 2.0 * dot / vnorm2;
                for (size_t i = k; i < n; ++i) A[i * 4 + c] -= f * A[i * 4 + k];
            }
        }
        A[k * 4 + k] = alpha;
    }
    double p[3];
    for (int k = 2; k >= 0; --k) {
        double r = A[k * 4 + 3];
        for (int c = k + 1; c < 3; ++c) r -= A[k * 4 + c] * p[c];
        p[k] = r / A[k * 4 + k];
    }
    free(A);
    double normal[3] = {p[0], p[1], 1.0};                      /* :114 */
    double length = sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
    plane[0] = (float)(normal[0] / length);                    /* :116,118 */
    plane[1] = (float)(normal[1] / length);
    plane[2] = (float)(normal[2] / length);
    plane[3] = (float)(p[2] / length);                         /* :117 */
}

/* :240-272.  (z > 0.8 points were already removed at :167-169.) */
size_t orc_combined_cloud(const orc_scene* s, const double pose[6], float* out)
{
    size_t w = 0;
    for (int j = 0; j < s->n_clouds; ++j) {
        double M[16];
        cloud_matrix(s, j, pose, M);
        const float* pts = s->xyz[j];
        for (size_t i = 0; i < s->n_pts[j]; ++i) {
            if (pts[3 * i + 2] > 0.8) continue;               /* :250 */
            orc_transform_point(M, pts + 3 * i, out + 3 * w);
            ++w;
        }
    }
    return w;
}

/* ------------------------------------------------------------------ */
/* voxel grid                                                          */
/* ------------------------------------------------------------------ */

typedef struct { int64_t key; uint32_t idx; } vox_rec;

static int vox_cmp(const void* a, const void* b)
{
    const vox_rec* x = (const vox_rec*)a;
    const vox_rec* y = (const vox_rec*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* PCL VoxelGrid (filters/voxel_grid.hpp applyFilter) as used at :199-203:
 * inverse_leaf = 1/leaf (float); ijk = floor(p*inverse_leaf) - min_b in float
 * arithmetic; linear index = i + j*dx + k*dx*dy; centroid = float sum / count.
 * PCL orders equal keys by std::sort (unspecified); this stand-in keeps input
 * order inside a voxel. */
size_t orc_voxel_grid(const float* xyz, size_t n, float leaf, float* out)
{
    if (n == 0) return 0;
    float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = xyz[3 * i + k];
            if (!isfinite(v)) continue;
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    int64_t minb[3], div[3];
    for (int k = 0; k < 3; ++k) {
        minb[k] = (int64_t)floorf(mn[k] * inv);
        int64_t maxb = (int64_t)floorf(mx[k] * inv);
        div[k] = maxb - minb[k] + 1;
    }
    vox_rec* rec = (vox_rec*)malloc(n * sizeof(vox_rec));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        const float* p = xyz + 3 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        int64_t i0 = (int64_t)(floorf(p[0] * inv) - (float)minb[0]);
        int64_t i1 = (int64_t)(floorf(p[1] * inv) - (float)minb[1]);
        int64_t i2 = (int64_t)(floorf(p[2] * inv) - (float)minb[2]);
        rec[m].key = i0 + i1 * div[0] + i2 * div[0] * div[1];
        rec[m].idx = (uint32_t)i;
        ++m;
    }
    qsort(rec, m, sizeof(vox_rec), vox_cmp);
    size_t w = 0, i = 0;
    while (i < m) {
        size_t j = i;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        while (j < m && rec[j].key == rec[i].key) {
            const float* p = xyz + 3 * (size_t)rec[j].idx;
            sx += p[0]; sy += p[1]; sz += p[2];
            ++j;
        }
        float cnt = (float)(j - i);
        out[3 * w + 0] = sx / cnt;
        out[3 * w + 1] = sy / cnt;
        out[3 * w + 2] = sz / cnt;
        ++w;
        i = j;
    }
    free(rec);
    return w;
}

/* ------------------------------------------------------------------ */
/* gradient descent (gdcpp-shaped)                                     */
/* ------------------------------------------------------------------ */

void orc_gd_default_opts(orc_gd_opts* o)
{
    o->max_iterations = 10000;      /* :431 */
    o->min_gradient_len = 1e-6;     /* :432 */
    o->min_step_len = 1e-6;         /* :433 */
    o->momentum = 0.4;              /* :434 */
    o->fd_eps = sqrt(DBL_EPSILON);  /* gdcpp CentralDifferences default */
    o->wolfe_decrease = 0.8;        /* gdcpp WolfeBacktracking defaults */
    o->wolfe_c1 = 1e-4;
    o->wolfe_c2 = 0.9;
    o->wolfe_min_step = 1e-10;
    o->wolfe_max_step = 1.0;
    o->wolfe_max_it = 0;
}

typedef struct {
    const orc_scene* s;
    const float* plane;
    int dim;
    int64_t n_evals;
} gd_env;

/* Functor :419-426 (6-D) and :474-484 (10-D) */
static double gd_objective(gd_env* e, const double* x)
{
    e->n_evals++;
    if (e->dim == 6) return orc_get_error(e->s, x, e->plane);
    return orc_get_error_plane(e->s, x, x + 6);
}

/* gdcpp evaluateObjective + CentralDifferences: f(x), then for i in 0..2D-1:
 * x[i/2] +/- eps/2, gradient_k = (f_{2k} - f_{2k+1}) / eps. */
static double gd_eval(gd_env* e, const orc_gd_opts* o, const double* x, double* grad)
{
    int D = e->dim;
    double fval = gd_objective(e, x);
    double fvals[20];
    for (int i = 0; i < 2 * D; ++i) {
        double xn[10];
        memcpy(xn, x, D * sizeof(double));
        if (i % 2 == 0) xn[i / 2] += o->fd_eps / 2;
        else            xn[i / 2] -= o->fd_eps / 2;
        fvals[i] = gd_objective(e, xn);
    }
    for (int k = 0; k < D; ++k) grad[k] = (fvals[2 * k] - fvals[2 * k + 1]) / o->fd_eps;
    return fval;
}

/* VectorXd::dot / ::squaredNorm (and ::norm = sqrt of it) as Eigen 3.3 evaluates them: Redux.h,
 * redux_impl<..., LinearVectorizedTraversal, NoUnrolling> over the coefficient-wise products, SSE2
 * packets of two doubles on x86-64 without -march:
 *   p0 = packet(0); p1 = packet(2); for i = 4, 8, ... below alignedEnd2: p0 += packet(i), p1 += packet(i+2);
 *   p0 += p1; if a whole packet is left: p0 += packet(alignedEnd2); res = p0.lo + p0.hi; then the odd tail.
 * Eigen's aligned allocation puts a VectorXd's data on a 16-byte boundary, so alignedStart = 0. */
static double dotn(const double* a, const double* b, int n)
{
    double v[16];
    for (int i = 0; i < n; ++i) v[i] = a[i] * b[i];
    if (n < 2) return n == 1 ? v[0] : 0.0;
    int packet = 2;
    int aligned_end2 = (n / (2 * packet)) * (2 * packet);
    int aligned_end = (n / packet) * packet;
    double p0[2] = {v[0], v[1]};
    if (aligned_end > packet) {
        double p1[2] = {v[2], v[3]};
        for (int i = 2 * packet; i < aligned_end2; i += 2 * packet) {
            p0[0] = p0[0] + v[i];     p0[1] = p0[1] + v[i + 1];
            p1[0] = p1[0] + v[i + 2]; p1[1] = p1[1] + v[i + 3];
        }
        p0[0] = p0[0] + p1[0]; p0[1] = p0[1] + p1[1];
        if (aligned_end > aligned_end2) {
            p0[0] = p0[0] + v[aligned_end2]; p0[1] = p0[1] + v[aligned_end2 + 1];
        }
    }
    double res = p0[0] + p0[1];
    for (int i = aligned_end; i < n; ++i) res = res + v[i];
    return res;
}

/* gdcpp WolfeBacktracking::operator() */
static double gd_wolfe(gd_env* e, const orc_gd_opts* o, const double* x, double fval,
                       const double* grad)
{
    int D = e->dim;
    double step = o->wolfe_max_step / o->wolfe_decrease;
    int armijo = 0, wolfe = 0;
    int it = 0;
    double gg = dotn(grad, grad, D);
    while ((o->wolfe_max_it <= 0 || it < o->wolfe_max_it) &&
           step * o->wolfe_decrease >= o->wolfe_min_step &&
           !(armijo && wolfe)) {
        double xn[10], gn[10];
        step = o->wolfe_decrease * step;
        for (int k = 0; k < D; ++k) xn[k] = x[k] - step * grad[k];
        double fn = gd_eval(e, o, xn, gn);
        armijo = fn <= fval - o->wolfe_c1 * step * gg;
        wolfe = dotn(grad, gn, D) <= o->wolfe_c2 * gg;
        ++it;
    }
    return step;
}

/* gdcpp GradientDescent::minimize with momentum; settings :431-435 */
void orc_gd_minimize(const orc_scene* s, const float plane[4], int dim, const double* x0,
                     const orc_gd_opts* o, orc_gd_result* r)
{
    gd_env e = {s, plane, dim, 0};
    int D = dim;
    double x[10], grad[10], step[10];
    memcpy(x, x0, D * sizeof(double));
    memset(step, 0, sizeof step);
    memset(grad, 0, sizeof grad);
    double fval = 0.0;
    double grad_len = o->min_gradient_len + 1;
    double step_len = o->min_step_len + 1;
    int it = 0;
    while ((o->max_iterations <= 0 || it < o->max_iterations) &&
           grad_len >= o->min_gradient_len && step_len >= o->min_step_len) {
        for (int k = 0; k < D; ++k) x[k] -= step[k];
        fval = gd_eval(&e, o, x, grad);
        grad_len = sqrt(dotn(grad, grad, D));
        double alpha = gd_wolfe(&e, o, x, fval, grad);
        for (int k = 0; k < D; ++k)
            step[k] = o->momentum * step[k] + (1 - o->momentum) * alpha * grad[k];
        step_len = sqrt(dotn(step, step, D));
        ++it;
    }
    r->iterations = it;
    r->converged = grad_len < o->min_gradient_len || step_len < o->min_step_len;
    r->fval = fval;
    memset(r->xval, 0, sizeof r->xval);
    memcpy(r->xval, x, D * sizeof(double));
    r->n_evals = e.n_evals;
}

/* ------------------------------------------------------------------ */
/* parsers                                                             */
/* ------------------------------------------------------------------ */

/* :70-98 getTransVector / helpers.cpp:244-251: split on ',', stof each field,
 * first three divided by the metric scale, the rest optionally through
 * degreeToRadian(int).  Returns the number of fields parsed (<= 6 stored). */
int orc_parse_pose_csv(const char* line, double scale, int angle_is_degree, double pose[6])
{
    int n = 0;
    const char* p = line;
    while (*p && n < 6) {
        char* endp;
        float f = strtof(p, &endp);            /* std::stof */
        if (endp == p) break;
        if (n < 3) pose[n] = f / scale;
        else {
            double value = f;
            if (angle_is_degree) value = orc_degree_to_radian((int)value);
            pose[n] = value;
        }
        ++n;
        p = strchr(endp, ',');
        if (!p) break;
        ++p;
    }
    return n;
}

/* :131-146: basename -> text before first '.' -> field 1 of split on '_' */
int orc_file_id(const char* filename)
{
    const char* base = strrchr(filename, '/');
    base = base ? base + 1 : filename;
    char buf[512];
    size_t len = strcspn(base, ".");
    if (len >= sizeof buf) len = sizeof buf - 1;
    memcpy(buf, base, len);
    buf[len] = 0;
    char* us = strchr(buf, '_');
    if (!us) return -1;
    char* f = us + 1;
    char* us2 = strchr(f, '_');
    if (us2) *us2 = 0;
    return atoi(f);
}

/* ------------------------------------------------------------------ */
/* RANSAC plane                                                        */
/* ------------------------------------------------------------------ */

/* The reference never calls its RANSAC helper (cloud_analyzer.cpp:103-121 is not
 * built) and PCL is not in its tree, so there is nothing to restate line by
 * line: this is the CPU statement of the library's own sampling scheme, scored
 * with the reference's distance arithmetic (:121-124) on q = float(M p).
 * Parity unpinned.
 *
 * What the call at cloud_analyzer.cpp:103-121 would do inside PCL (1.7-1.12, from its published
 * sources; stated for orientation, none of it can be run here):
 *   - SACSegmentation defaults the reference leaves untouched: max_iterations 50, probability 0.99;
 *     RandomSampleConsensus draws 3 distinct indices with rand(), skips collinear samples, counts
 *     points with |a x + b y + c z + d| < threshold (strict), keeps a model only if its count is
 *     GREATER than the best so far (first found wins ties), and shrinks the iteration bound from the
 *     inlier ratio;
 *   - setOptimizeCoefficients(true): SampleConsensusModelPlane::optimizeModelCoefficients takes the
 *     centroid and covariance of the winner's inliers, the eigenvector of the smallest eigenvalue as
 *     the normal, d = -n.centroid.  The inliers are not selected again with the refined plane.
 * Kept from that: the strict threshold test, first-found wins ties, the covariance refit on the
 * winner's inliers.  Replaced: rand() and the adaptive iteration count (a fixed number of
 * hypotheses from a counter-based generator, so that every hypothesis can be scored at once). */
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* Point `g` of the set (clouds back to back) carried to the base frame: float(M_j p), :303. */
static void ransac_point(const orc_scene* s, const double* M, const int64_t* off, int64_t g, float q[3])
{
    int j = 0;
    while (g >= off[j + 1]) ++j;
    orc_transform_point(M + 16 * j, s->xyz[j] + 3 * (g - off[j]), q);
}

/* Hypothesis h: indices i_k = splitmix64(seed + 3h + k) mod N, a repeat bumped to the next index;
 * n = (q1 - q0) x (q2 - q0) normalised with n.z >= 0, d = -n.q0; pl = (a, b, c, d, |n|).
 * Returns 0 for a (nearly) collinear sample. */
static int ransac_hypothesis(const orc_scene* s, const double* M, const int64_t* off, uint64_t seed,
                             int h, double pl[5])
{
    int64_t total = off[s->n_clouds];
    int64_t idx[3];
    for (int k = 0; k < 3; ++k) {
        int64_t v = (int64_t)(splitmix64(seed + 3ull * (uint64_t)h + (uint64_t)k) % (uint64_t)total);
        for (int t = 0; t < k; ++t)
            if (v == idx[t]) { v = (v + 1) % total; t = -1; }
        idx[k] = v;
    }
    double p[3][3];
    for (int k = 0; k < 3; ++k) {
        float q[3];
        ransac_point(s, M, off, idx[k], q);
        for (int r = 0; r < 3; ++r) p[k][r] = (double)q[r];
    }
    double u[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
    double v[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
    double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(len > 1e-12)) return 0;
    if (n[2] < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    n[0] = n[0] / len; n[1] = n[1] / len; n[2] = n[2] / len;
    pl[0] = n[0]; pl[1] = n[1]; pl[2] = n[2];
    pl[3] = -(n[0] * p[0][0] + n[1] * p[0][1] + n[2] * p[0][2]);
    pl[4] = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    return 1;
}

/* Shared set-up: per-cloud matrices and point offsets (caller frees both). */
static void ransac_setup(const orc_scene* s, const double pose[6], double** M_out, int64_t** off_out)
{
    int C = s->n_clouds;
    double* M = (double*)malloc((size_t)C * 16 * sizeof(double));
    int64_t* off = (int64_t*)malloc((size_t)(C + 1) * sizeof(int64_t));
    off[0] = 0;
    for (int j = 0; j < C; ++j) {
        cloud_matrix(s, j, pose, M + 16 * j);
        off[j + 1] = off[j] + (int64_t)s->n_pts[j];
    }
    *M_out = M;
    *off_out = off;
}

int orc_ransac_hypothesis_plane(const orc_scene* s, const double pose[6], uint64_t seed, int h, double pl[5])
{
    double* M;
    int64_t* off;
    ransac_setup(s, pose, &M, &off);
    int ok = ransac_hypothesis(s, M, off, seed, h, pl);
    free(M); free(off);
    return ok;
}

/* Winner among n_hyp hypotheses (most inliers, first found on ties), its inlier count, and the ten
 * moments (1, x, y, z, xx, xy, yy, xz, yz, zz about `org`) of its inliers for the refit, which the
 * caller does (oracle.py: numpy's LAPACK eigen-decomposition, so that the refit shares no code and
 * no algorithm with the library's Jacobi sweeps).  plane_out: the winning HYPOTHESIS (not refitted). */
void orc_ransac_plane(const orc_scene* s, const double pose[6], int n_hyp, float threshold,
                      uint64_t seed, double plane_out[5], int64_t* inliers_out, int* best_hyp_out,
                      double moments_out[10], double org_out[3])
{
    double* M;
    int64_t* off;
    ransac_setup(s, pose, &M, &off);
    int C = s->n_clouds;
    int64_t total = off[C];
    float* q = (float*)malloc((size_t)total * 3 * sizeof(float));
    for (int j = 0; j < C; ++j)
        for (size_t i = 0; i < s->n_pts[j]; ++i)
            orc_transform_point(M + 16 * j, s->xyz[j] + 3 * i, q + 3 * (off[j] + (int64_t)i));

    int best_h = -1;
    int64_t best_cnt = -1;
    double best_pl[5] = {0, 0, 0, 0, 1};
    for (int h = 0; h < n_hyp; ++h) {
        double pl[5];
        if (!ransac_hypothesis(s, M, off, seed, h, pl)) continue;
        int64_t cnt = 0;
        for (int64_t i = 0; i < total; ++i) {
            double num = pl[0] * (double)q[3 * i] + pl[1] * (double)q[3 * i + 1] + pl[2] * (double)q[3 * i + 2] + pl[3];
            if (fabs(num) / pl[4] < (double)threshold) ++cnt;
        }
        if (cnt > best_cnt) { best_cnt = cnt; best_h = h; memcpy(best_pl, pl, sizeof pl); }
    }
    /* origin of the moments: the first cloud's centre carried to the base frame (any fixed point
     * near the data keeps the sums well conditioned) */
    double c0[3] = {0, 0, 0};
    for (size_t i = 0; i < s->n_pts[0]; ++i)
        for (int k = 0; k < 3; ++k) c0[k] += s->xyz[0][3 * i + k];
    for (int k = 0; k < 3; ++k) c0[k] /= (double)s->n_pts[0];
    for (int r = 0; r < 3; ++r)
        org_out[r] = M[r * 4 + 0] * c0[0] + M[r * 4 + 1] * c0[1] + M[r * 4 + 2] * c0[2] + M[r * 4 + 3];
    for (int k = 0; k < 10; ++k) moments_out[k] = 0.0;
    if (best_h >= 0)
        for (int64_t i = 0; i < total; ++i) {
            double num = best_pl[0] * (double)q[3 * i] + best_pl[1] * (double)q[3 * i + 1] +
                         best_pl[2] * (double)q[3 * i + 2] + best_pl[3];
            if (!(fabs(num) / best_pl[4] < (double)threshold)) continue;
            double x = (double)q[3 * i] - org_out[0], y = (double)q[3 * i + 1] - org_out[1], z = (double)q[3 * i + 2] - org_out[2];
            double* m = moments_out;
            m[0] += 1.0; m[1] += x; m[2] += y; m[3] += z;
            m[4] += x * x; m[5] += x * y; m[6] += y * y; m[7] += x * z; m[8] += y * z; m[9] += z * z;
        }
    memcpy(plane_out, best_pl, sizeof best_pl);
    if (inliers_out) *inliers_out = best_cnt;
    if (best_hyp_out) *best_hyp_out = best_h;
    free(q); free(off); free(M);
}

/* ------------------------------------------------------------------ */
/* per-point distances and error classes                               */
/* ------------------------------------------------------------------ */

/* arithmetic 0: the optimiser's loop body (camera_calibration_optimization.cpp:389-404);
 * arithmetic 1: the viewer's plane-mode colouring (src/algorithms.cpp:379-385 with
 * apply_transformation = generic FP64 product, helpers.cpp:32-42, and
 * pointToPlaneDistanceD, algorithms.cpp:255-258).  Classes: algorithms.cpp:390-402. */
void orc_point_errors(const orc_scene* s, const double pose[6], const double plane[4],
                      int arithmetic, double* dist_out, unsigned char* class_out)
{
    size_t w = 0;
    for (int j = 0; j < s->n_clouds; ++j) {
        double M[16];
        cloud_matrix(s, j, pose, M);
        const float* pts = s->xyz[j];
        for (size_t i = 0; i < s->n_pts[j]; ++i, ++w) {
            double distance;
            if (arithmetic == 0) {
                float out[3];
                orc_transform_point(M, pts + 3 * i, out);
                distance = orc_distance_planed(plane, out);
            } else {
                double q[3];
                for (int r = 0; r < 3; ++r) {
                    double p = 0.0;
                    p = M[r * 4 + 0] * (double)pts[3 * i + 0] + p;
                    p = M[r * 4 + 1] * (double)pts[3 * i + 1] + p;
                    p = M[r * 4 + 2] * (double)pts[3 * i + 2] + p;
                    p = M[r * 4 + 3] * 1.0 + p;
                    q[r] = p;
                }
                distance = fabs(plane[0] * q[0] + plane[1] * q[1] + plane[2] * q[2] + plane[3]) /
                           (sqrt(pow(plane[0], 2) + pow(plane[1], 2) + pow(plane[2], 2)));
            }
            if (dist_out) dist_out[w] = distance;
            if (class_out) {
                double mm = distance * 1000.0;
                class_out[w] = mm < 2.5 ? 0 : (mm < 5.0 ? 1 : (mm < 10.0 ? 2 : 3));
            }
        }
    }
}
