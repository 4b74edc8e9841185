// kernels_common.hpp -- gfx950 kernels of the calibration hot path: shared definitions.
//
// Arithmetic families:
//   * "fast": FP32, the point-to-plane residual folded into a 4-vector per
//     (candidate, cloud) and contracted with homogeneous points on the matrix
//     cores: v_mfma_f32_16x16x4_f32 is exactly a K=4 contraction of a
//     16-candidate tile with a 16-point tile, bit-identical to an fmaf chain.
//     |.| and the running sum are VALU adds on the MFMA result.
//   * "exact": the reference's own arithmetic (FP64 SE3 apply in the SSE2
//     summation order, point rounded to float, FP64 distance with a correctly
//     rounded division, distances of a cloud added one by one in point order),
//     used for single poses, finite-difference stencils, the error report and
//     for re-scoring near-minimum grid candidates.  Its results are
//     bit-identical to the CPU evaluation of the same lines.
//   * "separable": for product grids, all translations of a rotation from one
//     histogram pass over the points (fast-path accuracy, integer sums).
//
// This header: types, layouts, constants and the per-point distance shared by all kernel headers
// (kernels_ingest / residual / separable / exact / fit / ransac .hpp).
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "pose_math.hpp"

#pragma clang fp contract(off)

namespace ccal {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// Layouts
// ---------------------------------------------------------------------------
// Fast path points: "groups" of 64 points = 4 MFMA tiles of 16 points.  A group
// is 64 lanes x float4 (1 KiB), so one global_load_dwordx4 per lane fetches the
// four B operands of a wave: component u of lane l is element l of tile u,
// element l = k*16 + col holds coordinate k (x,y,z,one) of point u*16 + col.
// Points are stored relative to their cloud's centre; padding points are all
// zero including `one`, so they contribute |0| to every sum.
//
// Fast path candidates: W[cloud][ctile][k*16 + cand] (64 floats per tile), one
// coalesced dword per lane is the A operand of a 16-candidate tile.
constexpr int kGroupPoints = 64;
constexpr int kTilePoints = 16;
constexpr int kFlushGroups = 64;  // FP32 partial sums are promoted every 256 adds

// Constants the host code of several translation units shares with the kernels.
constexpr int kCentreParts = 64;      // strided partial sums per cloud of the centre pass
constexpr int kChainRun = 8;                       // consecutive elements per lane
constexpr int kChainSeg = 64 * kChainRun;          // elements per wave segment
constexpr int kMoments = 10;          // sums of the moment pass: 1, x, y, z, xx, xy, yy, xz, yz, zz
constexpr int kMomentGrid = 2048;     // workgroups of the moment pass

// A block covers a contiguous run of points of one cloud (4096 at most): the unit of work of the
// per-point kernels of the exact path, the moment pass and the screening pass.
struct BlockDesc {
    int cloud;
    int first_of_cloud;  // index of the cloud's first block (for the finaliser)
    int64_t begin;
    int64_t end;
};

// Per (pose, cloud): rows 0..2 of M = base_T_flange_j * F(pose), 12 doubles.
// Per pose: plane a,b,c,d, sqrt(a^2+b^2+c^2), its rounded reciprocal and a flag
// (non-zero when the norm is in the range the reciprocal path covers), padded to 8.
constexpr int kCloudBound = 5;     // doubles per cloud of the screening budget's table: mean radius, norm bound, mean |dx|, |dy|, |dz|
constexpr int kPlaneStride = 8;

__device__ inline double exact_distance(const double* M, const double* pl, float x, float y,
                                        float z)
{
    // src/helpers.hpp:139-151: p = col3; p = v_i*col_i + p (i = 0,1,2); -> float
    const double xd = (double)x, yd = (double)y, zd = (double)z;
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double p = M[r * 4 + 3];
        p = xd * M[r * 4 + 0] + p;
        p = yd * M[r * 4 + 1] + p;
        p = zd * M[r * 4 + 2] + p;
        q[r] = (float)p;
    }
    // src/camera_calibration_optimization.cpp:121-129
    const double num = pl[0] * (double)q[0] + pl[1] * (double)q[1] + pl[2] * (double)q[2] + pl[3];
    return __builtin_fabs(num) / pl[4];
}

// a / b, correctly rounded, for b with a precomputed y = RN(1/b): two
// Newton corrections with exact FMA residuals.  After the first, q1 is a
// faithful rounding of a/b; with y within half an ulp of 1/b the second then
// lands on RN(a/b) (Markstein, "Computation of elementary functions on the IBM
// RISC System/6000 processor", 1990, theorem on q' = RN(q + r*y)).  The residuals
// are exact only away from underflow and overflow, so callers take this path for
// b in [2^-300, 2^300] and a zero or in [2^-600, 2^600] and divide otherwise.
__device__ inline double divide_by_known(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = __builtin_fma(-b, q0, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-b, q1, a);
    return __builtin_fma(r1, y, q1);
}

} // namespace ccal
