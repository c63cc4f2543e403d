// diff_kernel.h -- k_kin_diff: jacob0_dot / manipulability / jacobm / analytical Jacobians straight from the register-resident Jacobian
// (reference robot/Robot.py:701-905, 964-1235, robot/ETS.py:1671-1819), and the constant-address-space view of a chain table the kinematics
// kernels share.  A header of its own so that it is BOTH part of kin_kernels.hip (the built-in joint counts 1..16) and a translation unit
// hipRTC can instantiate at run time for the joint counts beyond (jit.cpp; kin_kernels.hip: launch_kin_diff) -- the reference loops over any n.
#pragma once
#include "kin_reg.h"
#include "diff_device.h"

namespace rtbhip {

// The chain tables through the constant address space: uniform loads become s_load (SGPR operands).
#define RTB_CONST __attribute__((address_space(4)))
struct ConstChain {
    const RTB_CONST DevSeg *seg;
    const RTB_CONST int32_t *jmeta;
};
__device__ __forceinline__ ConstChain const_view(const DevChain &dc)
{
    ConstChain cv;
    cv.seg = (const RTB_CONST DevSeg *)dc.seg;
    cv.jmeta = (const RTB_CONST int32_t *)dc.jmeta;
    return cv;
}

// k_kin_diff's own view: the same two tables, plus the request for the fused-chain translation form (kin_reg.h: cv_t3fma) -- the form the structured
// segment products are instances of, so that k_kin_diff<NJ, MODE, SIG> returns k_kin_diff<NJ, MODE, 0>'s bits
struct ConstChainF : ConstChain { typedef void t3fma_tag; };
__device__ __forceinline__ ConstChainF const_view_f(const DevChain &dc)
{
    ConstChainF cv;
    cv.seg = (const RTB_CONST DevSeg *)dc.seg;
    cv.jmeta = (const RTB_CONST int32_t *)dc.jmeta;
    return cv;
}

// jacob0_dot / manipulability / jacobm straight from the register-resident Jacobian: the (n,6,n) Hessian
// the reference materialises for each of them (robot/Robot.py:1069, robot/ETS.py:1671) never exists.
// SIG (round 6): the chain's structure signature (kin_reg.h: SegSig; an all-revolute chain of up to 8 joints without flips, called without a tool) --
// the walk multiplies by every constant in the form of its class, as k_ik's instantiations do; compiled at run time for the robot at hand (jit.cpp;
// kin_kernels.hip: launch_kin_diff), the general kernel serving until the code object is there.  Same bits: the walk by construction (exactform.h),
// the consumers because their sums of products are written out (diff_device.h).
constexpr int kDiffMax = 16;   // jacob_dot / manipulability / jacobm / analytical Jacobian: compile-time joint counts up to here
enum { kDiffJdot = 0, kDiffManip = 1, kDiffJacobm = 2, kDiffAnalytical = 3, kDiffAnalyticalDot = 4 };
#ifndef RTB_DIFF_WAVES
#define RTB_DIFF_WAVES 2        // waves per SIMD the register allocator must leave room for (chains of up to 8 joints; A/B knob)
#endif
#ifndef RTB_DIFF_WAVES_MANIP
#define RTB_DIFF_WAVES_MANIP 4  // manipulability fits 128 registers (hipcc: 123-125 without being asked; hipRTC took 157 -- three waves -- until told)
#endif
template <int NJ, int MODE, SegSig SIG = 0>
__global__ __launch_bounds__(kWave, (NJ <= kRegMaxJoints && MODE != 4 ? (MODE == kDiffManip ? RTB_DIFF_WAVES_MANIP : RTB_DIFF_WAVES) : 1)) void k_kin_diff(KinParams kp, DevChain dc, int axes, const double *__restrict__ q,
                                                       const double *__restrict__ qd, double *__restrict__ out)
{
    static_assert(SIG == 0 || (NJ <= kRegMaxJoints && MODE != kDiffAnalyticalDot), "structure instantiations: register-resident sizes, one walk");
    extern __shared__ __attribute__((aligned(16))) double buf[];
    const ConstChainF cv = const_view_f(dc);
    const int lane = threadIdx.x;
    const int64_t cfg0 = (int64_t)xcd_tile() * kWave, cfg = cfg0 + lane;
    const int64_t left = kp.N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    Pose P;
    double jac[6 * NJ];
    if constexpr (MODE == kDiffAnalyticalDot) {
        // Robot.jacob0_dot with an orientation `representation` (robot/Robot.py:1065-1098): the reference has no closed form
        // ("not actually sure this can be written in closed form") and takes  H = numhess(jacob0_analytical, q)  -- spatialmath's
        // FORWARD difference  H[i] = (Ja(q + dx e_i) - Ja(q)) / dx,  dx = 1e-8 -- then  Jd = tensordot(H, qd, (0, 0)).  Restated
        // as it stands (a drop-in returns the reference's numbers, truncation error included): n + 1 chain walks per lane,
        // everything in registers.
        const bool live = cfg < kp.N;
        double qv[NJ], v[NJ], jd[6 * NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = jm_jq(cv.jmeta[j]);
            qv[j] = live ? q[cfg * kp.qw + col] : 0.0;
            v[j] = live ? qd[cfg * kp.qw + col] : 0.0;
        }
        jacob_analytical_dot<NJ>(cv, kp.tail, qv, v, axes, jd);
        constexpr int W = 6 * NJ;
#pragma unroll
        for (int r = 0; r < kWave / kJRound; ++r) {
            if (lane / kJRound == r) reg_stage_J<NJ>(jd, buf, lane % kJRound);
            __syncthreads();
            int rows = ncfg - r * kJRound;
            rows = rows < 0 ? 0 : (rows > kJRound ? kJRound : rows);
            kin_flush(buf, W + 1, W, rows, out + (cfg0 + r * kJRound) * W, lane);
            __syncthreads();
        }
        return;
    }
    reg_compute<NJ, true, SIG>(kp, cv, q, cfg, P, jac);
    if (MODE == kDiffJdot || MODE == kDiffAnalytical) {
        double jd[6 * NJ];
        if (MODE == kDiffJdot) {
            double v[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[j] = cfg < kp.N ? qd[cfg * kp.qw + jm_jq(cv.jmeta[j])] : 0.0;
            jacob_dot<NJ>(jac, v, jd);
        } else {
            jacob_analytical<NJ>(P, jac, axes, jd);          // axes carries the representation code
        }
        constexpr int W = 6 * NJ;
#pragma unroll
        for (int r = 0; r < kWave / kJRound; ++r) {
            if (lane / kJRound == r) reg_stage_J<NJ>(jd, buf, lane % kJRound);
            __syncthreads();
            int rows = ncfg - r * kJRound;
            rows = rows < 0 ? 0 : (rows > kJRound ? kJRound : rows);
            kin_flush(buf, W + 1, W, rows, out + (cfg0 + r * kJRound) * W, lane);
            __syncthreads();
        }
    } else if (MODE == kDiffManip) {
        // axes: bits 0..5 = Cartesian rows, bits 8..9 = method (0 yoshikawa, 1 minsingular, 2 invcondition)
        const int method = (axes >> 8) & 3;
        const double m = method == 0 ? manipulability_yoshikawa<NJ>(jac, axes & 63) : manipulability_singular<NJ>(jac, axes & 63, method);
        if (cfg < kp.N) out[cfg] = m;                      // 8 bytes per lane, contiguous across the wave
    } else {
        double jm[NJ];
        jacobm<NJ>(jac, axes, jm);
        constexpr int S = NJ | 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) buf[lane * S + j] = jm[j];
        __syncthreads();
        flush_run(buf, S, NJ, ncfg, out + cfg0 * NJ, lane);
    }
}


}  // namespace rtbhip
