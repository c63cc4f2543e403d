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

// Hessian from a supplied Jacobian, phase A: `ncfg` consecutive (6,n) Jacobians (W = 6n doubles each, contiguous; W is
// even) into LDS rows of stride W + 1
RTB_HD void hj_load_tile(const double *__restrict__ src, int W, int ncfg, double *rows, int lane)
{
    const int total = ncfg * W;
    for (int f = 2 * lane; f < total; f += 2 * kWave) {
        const double2 v = *reinterpret_cast<const double2 *>(src + f);
        const int r = f / W, e = f - r * W;      // W even: a piece never straddles two rows
        double *dst = rows + r * (W + 1) + e;
        dst[0] = v.x; dst[1] = v.y;
    }
}

}  // namespace rtbhip
