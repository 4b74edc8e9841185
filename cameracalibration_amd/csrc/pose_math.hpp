// pose_math.hpp -- FP64 pose algebra shared by host code and HIP kernels.
//
// Every expression is written as separate multiplies and adds in a fixed order
// and the translation units are built with -ffp-contract=off, so the host and
// the device produce bit-identical matrices from the same sines and cosines.
// Reference lines restated here: src/helpers.cpp:45-71 (eul2rot ZYX +
// vectorToTransformationMatrix), src/camera_calibration_optimization.cpp:290
// (M = base_T_flange * flange_T_camera), src/helpers.hpp:139-151 (SE3 apply).
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#define CCAL_HD __host__ __device__ inline
#else
#define CCAL_HD inline
#endif

namespace ccal {

// sin/cos of the three pose angles in pose order: [rz, ry, rx].
struct SinCos3 {
    double s[3];
    double c[3];
};

// Sine and cosine of one angle the way the reference's binary gets them: Eigen's
// AngleAxis::toRotationMatrix takes sin and cos of the same angle, and gcc merges such a pair into
// one call of glibc's sincos() at any optimisation level.  sincos() is not bit-identical to
// sin()/cos(): about one argument in a thousand differs in the last place, so every host-side
// sine/cosine of this library goes through this one function.
inline void sin_cos(double angle, double* s, double* c)
{
    ::sincos(angle, s, c);
}

inline SinCos3 sincos_of_pose(const double pose[6])
{
    SinCos3 sc;
    for (int k = 0; k < 3; ++k) sin_cos(pose[3 + k], &sc.s[k], &sc.c[k]);
    return sc;
}

// R = ((I*Rz)*Ry)*Rx with each factor built the way Eigen's AngleAxis does for
// a unit axis: c off-axis, (1-c)+c on the axis, +-s off-diagonal.  With those
// sparse factors every product coefficient has at most two non-zero terms, so
// the result does not depend on Eigen's reduction order.
CCAL_HD void rotation_zyx(const double s[3], const double c[3], double R[9])
{
    const double sz = s[0], cz = c[0], oz = (1.0 - cz) * 1.0 + cz;
    const double sy = s[1], cy = c[1], oy = (1.0 - cy) * 1.0 + cy;
    const double sx = s[2], cx = c[2], ox = (1.0 - cx) * 1.0 + cx;
    // A = Rz * Ry
    const double a00 = cz * cy, a01 = (-sz) * oy, a02 = cz * sy;
    const double a10 = sz * cy, a11 = cz * oy,    a12 = sz * sy;
    const double a20 = oz * (-sy), a21 = 0.0,     a22 = oz * cy;
    // R = A * Rx ; column 0 = A[:,0]*ox, column 1 = A[:,1]*cx + A[:,2]*sx,
    //              column 2 = A[:,1]*(-sx) + A[:,2]*cx
    R[0] = a00 * ox; R[1] = a01 * cx + a02 * sx; R[2] = a01 * (-sx) + a02 * cx;
    R[3] = a10 * ox; R[4] = a11 * cx + a12 * sx; R[5] = a11 * (-sx) + a12 * cx;
    R[6] = a20 * ox; R[7] = a21 * cx + a22 * sx; R[8] = a21 * (-sx) + a22 * cx;
}

// Rows 0..2 of M = B * F where F = [R t; 0 1] and B is a full 4x4 (row-major),
// accumulated in k order from zero as Eigen's small lazy product does.
CCAL_HD void compose_rows(const double B[16], const double R[9], const double t[3],
                          double M[12])
{
    for (int i = 0; i < 3; ++i) {
        const double b0 = B[i * 4 + 0], b1 = B[i * 4 + 1], b2 = B[i * 4 + 2], b3 = B[i * 4 + 3];
        for (int j = 0; j < 3; ++j) {
            double r = b0 * R[0 * 3 + j];
            r = b1 * R[1 * 3 + j] + r;
            r = b2 * R[2 * 3 + j] + r;
            r = b3 * 0.0 + r;
            M[i * 4 + j] = r;
        }
        double r = b0 * t[0];
        r = b1 * t[1] + r;
        r = b2 * t[2] + r;
        r = b3 * 1.0 + r;
        M[i * 4 + 3] = r;
    }
}

// Folded 4-vector of the residual for one (pose, cloud):
//   (n.(R_M p + t_M) + d)/|n| = w[0..2].p + w[3]
// then re-expressed for points stored relative to `centre`.
CCAL_HD void fold_plane(const double M[12], const double n[3], double d, double inv_norm,
                        const double centre[3], double w[4])
{
    for (int j = 0; j < 3; ++j)
        w[j] = (n[0] * M[0 * 4 + j] + n[1] * M[1 * 4 + j] + n[2] * M[2 * 4 + j]) * inv_norm;
    double w3 = (n[0] * M[3] + n[1] * M[7] + n[2] * M[11] + d) * inv_norm;
    w[3] = w3 + (w[0] * centre[0] + w[1] * centre[1] + w[2] * centre[2]);
}

} // namespace ccal
