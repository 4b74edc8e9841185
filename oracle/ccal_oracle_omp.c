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
