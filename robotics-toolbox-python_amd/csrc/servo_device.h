// servo_device.h -- per-tile phases of two small exports of the reference's extension module that take finished
// matrices rather than a chain:
//   fknm.Angle_Axis   (core/fknm.cpp:112-162 -> _angle_axis core/ik.cpp:241-286; consumer tools/p_servo.py:7-43)
//   ETS_hessian0 / ETS_hessiane with a SUPPLIED Jacobian (core/fknm.cpp:583-783 -> _ETS_hessian core/methods.cpp:16-32,
//                      a pure function of J)
// Both are HBM-bound: 2 x 128 B in + 48 B out per pose pair; 48 n B in + 48 n^2 B out per Jacobian.  A tile is 64
// consecutive units = one wavefront; inputs arrive as contiguous runs of 16-byte pieces staged through the wave's LDS
// (a lane-per-unit load would fetch 16-byte pieces at a 128-byte / 48n-byte stride), outputs leave the same way.
// Phase functions are __host__ __device__ so tests/emu can run them lane by lane.
#pragma once
#include "ik_device.h"

namespace rtbhip {

constexpr int kAaStride = 17;   // 16 doubles of a 4x4 + 1: odd, so per-lane row reads are bank-conflict free

// phase A: the wave copies `ncfg` consecutive 4x4 matrices (16 doubles each, contiguous) into LDS rows of kAaStride
RTB_HD void aa_load_tile(const double *__restrict__ src, int ncfg, double *rows, int lane)
{
    const int total = ncfg * 16;
    for (int f = 2 * lane; f < total; f += 2 * kWave) {
        const double2 v = *reinterpret_cast<const double2 *>(src + f);
        double *dst = rows + (f >> 4) * kAaStride + (f & 15);
        dst[0] = v.x; dst[1] = v.y;
    }
}
// The same copy in two steps, for the kernel: ALL of a tile's global loads first (eight 16-byte loads per lane, in flight together), the LDS
// writes afterwards.  The loop above has a run-time trip count: the compiler keeps it rolled and every trip is load -> wait -> LDS store, eight
// dependent HBM round trips per operand (what k_rne's first tile copy suffered from, rne_kernels.hip).
struct AaTile { double2 v[8]; };
__device__ __forceinline__ void aa_fetch_tile(const double *__restrict__ src, int ncfg, int lane, AaTile &t)
{
    const int total = ncfg * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int f = 2 * lane + 2 * kWave * k;
        t.v[k] = f < total ? *reinterpret_cast<const double2 *>(src + f) : double2{0.0, 0.0};
    }
}
__device__ __forceinline__ void aa_store_tile(const AaTile &t, int ncfg, double *rows, int lane)
{
    const int total = ncfg * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int f = 2 * lane + 2 * kWave * k;
        if (f < total) { double *dst = rows + (f >> 4) * kAaStride + (f & 15); dst[0] = t.v[k].x; dst[1] = t.v[k].y; }
    }
}

// phase B: this lane's error vector e = angle_axis(Te, Tep) into the staging rows (stride 7)
RTB_HD void aa_lane(const double *te16, const double *tep16, double *erow)
{
    Pose P;
    P.r00 = te16[0]; P.r01 = te16[1]; P.r02 = te16[2]; P.tx = te16[3];
    P.r10 = te16[4]; P.r11 = te16[5]; P.r12 = te16[6]; P.ty = te16[7];
    P.r20 = te16[8]; P.r21 = te16[9]; P.r22 = te16[10]; P.tz = te16[11];
    double e[6];
    // the accessor ik_angle_axis expects: k < 9 -> R row-major, k >= 9 -> t
    ik_angle_axis(P, [&](int k) { return k < 9 ? tep16[4 * (k / 3) + (k % 3)] : tep16[4 * (k - 9) + 3]; }, e);
#pragma unroll
    for (int k = 0; k < 6; ++k) erow[k] = e[k];
}

// phase B of p_servo's DEFAULT method "rpy" (tools/p_servo.py:88-97): e = [t ; rpy] of eTep = inv(wTe) wTep -- the error seen from
// the end-effector frame -- with rpy = spatialmath.base.tr2rpy(eTep, order "zyx", check=False) = (roll, pitch, yaw),
// R = Rz(yaw) Ry(pitch) Rx(roll).  tr2rpy is third-party (spatialmath-python, absent from the reference tree); its published
// algorithm is restated here branch for branch: |R20| = 1 within 10 eps is the singular case (roll = 0, yaw from R01, R02);
// otherwise roll = atan2(R21, R22), yaw = atan2(R10, R00) and the pitch from R20 over the LARGEST of |R00|, |R10|, |R21|, |R22|
// (the first one on a tie, as numpy.argmax).  The reference inverts wTe with a general 4x4 LU; for a rigid transform that is
// [R^T, -R^T t] up to rounding, which is what is formed here.
RTB_HD void servo_rpy_lane(const double *te, const double *tep, double *erow)
{
    double R[3][3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = tep[4 * k + 3] - te[4 * k + 3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = te[i] * tep[j] + te[4 + i] * tep[4 + j] + te[8 + i] * tep[8 + j];
        erow[i] = te[i] * d[0] + te[4 + i] * d[1] + te[8 + i] * d[2];
    }
    double roll, pitch, yaw;
    if (fabs(fabs(R[2][0]) - 1.0) < 10.0 * 2.220446049250313e-16) {
        roll = 0.0;
        yaw = R[2][0] < 0.0 ? -atan2(R[0][1], R[0][2]) : atan2(-R[0][1], -R[0][2]);
        const double c = R[2][0] < -1.0 ? -1.0 : (R[2][0] > 1.0 ? 1.0 : R[2][0]);
        pitch = -asin(c);
    } else {
        roll = atan2(R[2][1], R[2][2]);
        yaw = atan2(R[1][0], R[0][0]);
        const double m0 = fabs(R[0][0]), m1 = fabs(R[1][0]), m2 = fabs(R[2][1]), m3 = fabs(R[2][2]);
        if (m0 >= m1 && m0 >= m2 && m0 >= m3) pitch = -atan(R[2][0] * cos(yaw) / R[0][0]);
        else if (m1 >= m2 && m1 >= m3) pitch = -atan(R[2][0] * sin(yaw) / R[1][0]);
        else if (m2 >= m3) pitch = -atan(R[2][0] * sin(roll) / R[2][1]);
        else pitch = -atan(R[2][0] * cos(roll) / R[2][2]);
    }
    erow[3] = roll; erow[4] = pitch; erow[5] = yaw;
}

// Hessian from a supplied Jacobian, phase A: `ncfg` consecutive (6,n) Jacobians (W = 6n doubles each, contiguous; W is
// even) into LDS rows of stride W + 1
// (eight 16-byte loads per lane in flight per trip, then their LDS writes: a trip of the plain loop is load -> wait -> LDS store, one HBM round
// trip per 16 bytes -- see aa_fetch_tile)
RTB_HD void hj_load_tile(const double *__restrict__ src, int W, int ncfg, double *rows, int lane)
{
    const int total = ncfg * W;
    int f = 2 * lane;
    for (; f + 7 * 2 * kWave < total; f += 8 * 2 * kWave) {
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double2 *>(src + f + k * 2 * kWave);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = f + k * 2 * kWave, r = g / W, e = g - r * W;
            double *dst = rows + r * (W + 1) + e;
            dst[0] = v[k].x; dst[1] = v[k].y;
        }
    }
    for (; f < total; f += 2 * kWave) {
        const double2 v = *reinterpret_cast<const double2 *>(src + f);
        const int r = f / W, e = f - r * W;      // W even: a piece never straddles two rows
        double *dst = rows + r * (W + 1) + e;
        dst[0] = v.x; dst[1] = v.y;
    }
}

}  // namespace rtbhip
