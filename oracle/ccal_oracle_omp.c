/*
 * ccal_oracle_omp.c -- the same per-candidate arithmetic as ccal_oracle.c's
 * orc_grid_search, with the candidate loop spread over host threads.
 * TEST INFRASTRUCTURE ONLY: used as the timed multi-core CPU baseline.
 * The reference itself is single-threaded (all its "#pragma omp" lines are
 * commented out); this variant only answers "what would all host cores give".
 */
#include "ccal_oracle.h"
#include <omp.h>

int orc_omp_threads(void) { return omp_get_max_threads(); }

/* Costs for candidates [begin,end) written to costs_out; arg-min (lowest index
 * on ties) is resolved serially afterwards so the result equals the serial scan. */
void orc_grid_costs_omp(const orc_scene* s, const float plane[4], const double base[6],
                        const orc_grid* g, int64_t begin, int64_t end, double* costs_out,
                        int64_t* best_index, double* best_cost)
{
#pragma omp parallel for schedule(static)
    for (int64_t c = begin; c < end; ++c) {
        double pose[6];
        orc_grid_pose(g, base, c, pose);
        costs_out[c - begin] = orc_get_error(s, pose, plane);
    }
    double error_min = 1e9;
    int64_t best = -1;
    for (int64_t c = begin; c < end; ++c)
        if (costs_out[c - begin] < error_min) { error_min = costs_out[c - begin]; best = c; }
    *best_index = best;
    *best_cost = error_min;
}

/* Inlier counts of a few hypotheses at full size: every point is carried to the base frame once
 * (float(M_j p), camera_calibration_optimization.cpp:303) and tested against each plane with the
 * distance arithmetic of :121-124. */
void orc_ransac_counts(const orc_scene* s, const double pose[6], uint64_t seed, float threshold,
                       const int* hyp, int n, int64_t* counts_out)
{
    enum { MAXH = 64 };
    double pl[MAXH][5];
    int valid[MAXH];
    if (n > MAXH) n = MAXH;
    for (int k = 0; k < n; ++k) {
        valid[k] = orc_ransac_hypothesis_plane(s, pose, seed, hyp[k], pl[k]);
        counts_out[k] = valid[k] ? 0 : -1;
    }
    double F[16];
    orc_pose_to_matrix(pose, F);
    for (int j = 0; j < s->n_clouds; ++j) {
        double M[16];
        orc_matmul4(s->base_T_flange + 16 * j, F, M);
        const float* pts = s->xyz[j];
        int64_t np = (int64_t)s->n_pts[j];
        int64_t cnt[MAXH] = {0};
#pragma omp parallel
        {
            int64_t local[MAXH] = {0};
#pragma omp for schedule(static) nowait
            for (int64_t i = 0; i < np; ++i) {
                float q[3];
                orc_transform_point(M, pts + 3 * i, q);
                for (int k = 0; k < n; ++k) {
                    if (!valid[k]) continue;
                    double num = pl[k][0] * (double)q[0] + pl[k][1] * (double)q[1] + pl[k][2] * (double)q[2] + pl[k][3];
                    if (__builtin_fabs(num) / pl[k][4] < (double)threshold) ++local[k];
                }
            }
#pragma omp critical
            for (int k = 0; k < n; ++k) cnt[k] += local[k];
        }
        for (int k = 0; k < n; ++k)
            if (valid[k]) counts_out[k] += cnt[k];
    }
}
