// trig.h -- fp64 sin/cos pair for joint angles.
//
// The reference calls libm cos()/sin() per revolute joint (core/fknm.cpp:1324-1325,
// core/frne.c:323-326).  On the GPU the library sincos() is ~100 instructions with a data-dependent
// branch into a Payne-Hanek reduction, which (a) dominates the per-configuration instruction count
// and (b) puts every call in its own basic block, so the n independent evaluations of a
// configuration cannot be interleaved.  Joint angles are small numbers, so the common case is
// served by a branch-free straight-line evaluation:
//     k = rint(x * 2/pi);  r = x - k*pi/2 by three FMAs against a 3 x 53-bit split of pi/2
//     (absolute reduction error < 1e-16 for |x| < 2^20), then the classic minimax kernels for
//     sin r, cos r on |r| <= pi/4 (Sun fdlibm __kernel_sin/__kernel_cos coefficients) and a
//     quadrant swap.  Max abs error vs libm: < 3e-16 (tests/test_kernel_emu.py pins 5e-16).
// |x| >= 2^20 (never a joint angle, but the ABI accepts any double) falls back to the library
// routine, whole wave at once, so parity holds for every input.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <cmath>
#endif

#ifndef RTB_HD
#define RTB_HD __host__ __device__ __forceinline__
#endif

namespace rtbhip {

constexpr double kTrigFastLimit = 1048576.0;  // 2^20

RTB_HD void sincos_reduced(double x, double &s, double &c)
{
    const double k = rint(x * 0x1.45f306dc9c883p-1);        // x * 2/pi
    double r = fma(-k, 0x1.921fb54442d18p+0, x);           // pi/2, high 53 bits
    r = fma(-k, 0x1.1a62633145c07p-54, r);                 // next 53 bits
    r = fma(-k, -0x1.f1976b7ed8fbcp-110, r);               // and the next
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;                               // |k| < 2^20 fits
    const double a = (q & 1) ? cr : sr;
    const double b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// The same evaluation with its 16 constants read from a TABLE (kSincosTable's layout) instead of written as literals.  Why: inside a persistent
// loop (k_ik) the compiler materialises the literals once, in ~30 SGPRs, outside the loop -- and, when the loop body needs the scalar file for
// something else (the chain constants of the FK walk), parks them in VGPR lanes and fetches every one back with a v_readlane per iteration
// (64 of them, round 4).  Read through a laundered constant-address-space pointer they arrive by s_load where they are used and are gone again.
// Same operations in the same order as sincos_reduced: the same bits.
constexpr int kSincosTableLen = 16;
#define RTB_SINCOS_TABLE_INIT { 0x1.45f306dc9c883p-1, 0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54, -0x1.f1976b7ed8fbcp-110, \
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04, \
    8.33333333332248946124e-03, -1.66666666666666324348e-01, \
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05, \
    -1.38888888888741095749e-03, 4.16666666666666019037e-02 }
template <class TAB>
RTB_HD void sincos_reduced_tab(double x, double &s, double &c, TAB t)
{
    const double k = rint(x * t[0]);
    double r = fma(-k, t[1], x);
    r = fma(-k, t[2], r);
    r = fma(-k, t[3], r);
    const double z = r * r;
    double ps = fma(z, t[4], t[5]);
    ps = fma(z, ps, t[6]);
    ps = fma(z, ps, t[7]);
    ps = fma(z, ps, t[8]);
    ps = fma(z, ps, t[9]);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, t[10], t[11]);
    pc = fma(z, pc, t[12]);
    pc = fma(z, pc, t[13]);
    pc = fma(z, pc, t[14]);
    pc = fma(z, pc, t[15]);
    const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double a = (q & 1) ? cr : sr;
    const double b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// Keeps the machine scheduler from hoisting every segment's scalar loads to the top of the
// straight-line walk (which overflows the 102 SGPRs and turns each constant operand into a pair of
// v_readlane from a spill VGPR): loads of segment j+1 may overlap segment j, not run further ahead.
RTB_HD void sched_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// true when any lane of the wavefront holds `pred` (the CPU emulation runs one lane at a time)
// Tile index of this workgroup.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs (id % 8), each
// with its own L2 and its own path to memory; handing XCD x the x-th contiguous eighth of the tiles instead of every 8th
// tile makes neighbouring output runs (which share 4 KiB pages at their seams) come from the same XCD at about the same
// time.  Measured on the headline kernel, 8 interleaved A/B pairs on one box: 0.082-0.090 -> 0.080-0.083 ms, and much
// steadier; jacob0-only 0.068 -> 0.065, jacob0_dot and fkine_all 1-2 %.  Kernels whose runs are already tens of KB per
// wave or that are compute-bound lose 1-3 % with it (Hessian tile, k_partial, the fleet, RNE, the dynamics terms) and keep
// the identity mapping.  Leap-frogging chunks of 16 or 128 consecutive tiles per XCD instead of eighths: slower and as
// unsteady as the identity (0.082-0.090 ms against a steady 0.080).  A bijection of [0, grid) for any grid size.
#ifndef RTB_XCD_REMAP
#define RTB_XCD_REMAP 1
#endif
RTB_HD unsigned xcd_tile_of(unsigned g, unsigned b)     // grid size, workgroup id -> tile
{
    const unsigned x = b & 7u, q8 = g >> 3, r8 = g & 7u;
    return x * q8 + (x < r8 ? x : r8) + (b >> 3);
}
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned xcd_tile()
{
#if RTB_XCD_REMAP
    return xcd_tile_of(gridDim.x, blockIdx.x);
#else
    return blockIdx.x;
#endif
}
#endif

RTB_HD bool wave_any(bool pred)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(pred) != 0;
#else
    return pred;
#endif
}

RTB_HD void rtb_sincos(double x, double *s, double *c)
{
    if (wave_any(!(fabs(x) < kTrigFastLimit))) {   // also catches NaN / inf
        sincos(x, s, c);
        return;
    }
    sincos_reduced(x, *s, *c);
}

}  // namespace rtbhip
