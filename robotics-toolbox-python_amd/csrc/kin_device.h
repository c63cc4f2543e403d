// kin_device.h -- per-lane kinematics primitives shared by the fkine/jacobian/hessian, fleet and IK
// kernels.  One lane owns one configuration; the SE(3) pose is a 3x4 affine held in 12 fp64
// registers and every routine below indexes it statically so nothing spills to scratch.
//
// What it replaces in the reference: _ET_T + rx..tz (core/methods.cpp:354-370,
// core/fknm.cpp:1320-1555), the chain walks of _ETS_fkine (methods.cpp:318-352) and
// _ETS_jacob0/_ETS_jacobe (methods.cpp:112-316), and _ETS_hessian (methods.cpp:16-32).
//
// The Jacobian is NOT computed the way the reference does (reverse walk, end-effector-frame
// columns from rows of U, then a 6x6 block rotation); one forward walk records each joint's axis z_j
// and origin p_j in the start frame, and the columns are closed as  Jv = z_j x (p_e - p_j), Jw = z_j
// once the end-effector position is known.  Same quantity, ~1/3 of the flops, differs from the
// reference only by rounding (tests pin |dJ| <= 1e-10; observed ~1e-15).
//
// All functions are __host__ __device__ so tests/emu can execute the exact kernel body, lane by
// lane, on the CPU build box (which has no GPU).  The product never takes that path.
#pragma once
#include "rtbhip_internal.h"
#ifndef __HIPCC_RTC__
#include <cmath>
#endif

#define RTB_HD __host__ __device__ __forceinline__
#include "trig.h"
#include "exactform.h"

namespace rtbhip {

constexpr int kWave = 64;

struct Pose {
    double r00, r01, r02, r10, r11, r12, r20, r21, r22;
    double tx, ty, tz;
};

RTB_HD void pose_identity(Pose &P)
{
    P.r00 = 1; P.r01 = 0; P.r02 = 0;
    P.r10 = 0; P.r11 = 1; P.r12 = 0;
    P.r20 = 0; P.r21 = 0; P.r22 = 1;
    P.tx = 0; P.ty = 0; P.tz = 0;
}

// P <- P * Rot_axis(c, s): a rotation about a coordinate axis only mixes two columns.
// Every sum of two products below is WRITTEN OUT as one rounded product and one fused multiply-add (fp contract(off): nothing is left to the
// compiler).  Left to it, `a * c + b * s` becomes fma(a, c, round(b s)) or fma(b, s, round(a c)) depending on where a and b CAME FROM (LLVM orders
// the operands of a commutative add by the depth of their expression trees before it fuses) -- measured in round 6: the same source line gave
// different bits in a structure instantiation (a = a bare copy of an input) and in the general kernel (a = the end of a chain of three fused
// multiply-adds), and k_ik's signature kernels drifted from the general one by 1e-10 in q.  scripts/contraction_probe.hip reproduces it.
RTB_HD double mix_pp(double a, double c, double b, double s)      // a c + b s  :=  fma(a, c, round(b s))
{
#pragma clang fp contract(off)
    return __builtin_fma(a, c, b * s);
}
RTB_HD double mix_pm(double b, double c, double a, double s)      // b c - a s  :=  fma(b, c, -round(a s))
{
#pragma clang fp contract(off)
    return __builtin_fma(b, c, -(a * s));
}
RTB_HD double dot3x(double a0, double b0, double a1, double b1, double a2, double b2)      // a0 b0 + a1 b1 + a2 b2  :=  fma(a2, b2, fma(a1, b1, round(a0 b0)))
{
#pragma clang fp contract(off)
    return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));
}
RTB_HD double fmax_(double a, double b, double acc)               // fma(a, b, acc) as such: no `contract` flag, nothing fuses into or out of it
{
#pragma clang fp contract(off)
    return __builtin_fma(a, b, acc);
}
RTB_HD void pose_rotx(Pose &P, double c, double s)
{
    double a, b;
    a = P.r01; b = P.r02; P.r01 = mix_pp(a, c, b, s); P.r02 = mix_pm(b, c, a, s);
    a = P.r11; b = P.r12; P.r11 = mix_pp(a, c, b, s); P.r12 = mix_pm(b, c, a, s);
    a = P.r21; b = P.r22; P.r21 = mix_pp(a, c, b, s); P.r22 = mix_pm(b, c, a, s);
}
RTB_HD void pose_roty(Pose &P, double c, double s)
{
    double a, b;
    a = P.r00; b = P.r02; P.r00 = mix_pm(a, c, b, s); P.r02 = mix_pp(a, s, b, c);
    a = P.r10; b = P.r12; P.r10 = mix_pm(a, c, b, s); P.r12 = mix_pp(a, s, b, c);
    a = P.r20; b = P.r22; P.r20 = mix_pm(a, c, b, s); P.r22 = mix_pp(a, s, b, c);
}
RTB_HD void pose_rotz(Pose &P, double c, double s)
{
    double a, b;
    a = P.r00; b = P.r01; P.r00 = mix_pp(a, c, b, s); P.r01 = mix_pm(b, c, a, s);
    a = P.r10; b = P.r11; P.r10 = mix_pp(a, c, b, s); P.r11 = mix_pm(b, c, a, s);
    a = P.r20; b = P.r21; P.r20 = mix_pp(a, c, b, s); P.r21 = mix_pm(b, c, a, s);
}
// P <- P * Trans(axis, d): t += d * column
RTB_HD void pose_tx(Pose &P, double d) { P.tx = fmax_(d, P.r00, P.tx); P.ty = fmax_(d, P.r10, P.ty); P.tz = fmax_(d, P.r20, P.tz); }
RTB_HD void pose_ty(Pose &P, double d) { P.tx = fmax_(d, P.r01, P.tx); P.ty = fmax_(d, P.r11, P.ty); P.tz = fmax_(d, P.r21, P.tz); }
RTB_HD void pose_tz(Pose &P, double d) { P.tx = fmax_(d, P.r02, P.tx); P.ty = fmax_(d, P.r12, P.ty); P.tz = fmax_(d, P.r22, P.tz); }
// P.t += R (x, y, z) as three fused chains seeded with the old translation (one instruction less per component than sum-then-add; used by
// k_ik's plain walk, A/B switch RTB_POSE_T3_FMA)
RTB_HD void pose_t3_fma(Pose &P, double x, double y, double z)
{
    P.tx = fmax_(z, P.r02, fmax_(y, P.r01, fmax_(x, P.r00, P.tx)));
    P.ty = fmax_(z, P.r12, fmax_(y, P.r11, fmax_(x, P.r10, P.ty)));
    P.tz = fmax_(z, P.r22, fmax_(y, P.r21, fmax_(x, P.r20, P.tz)));
}
RTB_HD void pose_t3(Pose &P, double x, double y, double z)
{
#pragma clang fp contract(off)
    P.tx = P.tx + dot3x(x, P.r00, y, P.r01, z, P.r02);
    P.ty = P.ty + dot3x(x, P.r10, y, P.r11, z, P.r12);
    P.tz = P.tz + dot3x(x, P.r20, y, P.r21, z, P.r22);
}
// one row (x, y, z) of a pose times the constant rotation c (row-major, class CLS):  out_k = y c[3+k] + x c[k] + z c[6+k]  in that fixed order
template <int CLS, class F>
RTB_HD void row_times_const(double &x, double &y, double &z, F c)
{
    const double a = x, b = y, e = z;
    x = dotk<seg_kind(CLS, 3), seg_kind(CLS, 0), seg_kind(CLS, 6)>(c(3), b, c(0), a, c(6), e);
    y = dotk<seg_kind(CLS, 4), seg_kind(CLS, 1), seg_kind(CLS, 7)>(c(4), b, c(1), a, c(7), e);
    z = dotk<seg_kind(CLS, 5), seg_kind(CLS, 2), seg_kind(CLS, 8)>(c(5), b, c(2), a, c(8), e);
}
template <int CLS, class F>
RTB_HD void pose_rot_const(Pose &P, F c)      // P.R <- P.R * c
{
    row_times_const<CLS>(P.r00, P.r01, P.r02, c);
    row_times_const<CLS>(P.r10, P.r11, P.r12, c);
    row_times_const<CLS>(P.r20, P.r21, P.r22, c);
}
// P <- P * A for a general constant affine a = {R row-major (9), t (3)}
template <bool T3FMA = false, class F>
RTB_HD void pose_mul_general(Pose &P, F a)
{
    if (T3FMA) pose_t3_fma(P, a(9), a(10), a(11));
    else pose_t3(P, a(9), a(10), a(11));
    pose_rot_const<kSegGeneral>(P, a);      // every entry as fma(z, a(6+k), fma(x, a(k), round(y a(3+k)))): the form the structured products are instances of
}
// P <- A * P  (used once per configuration for the base transform)
RTB_HD void pose_premul(Pose &P, const double *a /* row-major 3x4 */)
{
    Pose O;
    O.r00 = a[0] * P.r00 + a[1] * P.r10 + a[2] * P.r20;
    O.r01 = a[0] * P.r01 + a[1] * P.r11 + a[2] * P.r21;
    O.r02 = a[0] * P.r02 + a[1] * P.r12 + a[2] * P.r22;
    O.tx = a[0] * P.tx + a[1] * P.ty + a[2] * P.tz + a[3];
    O.r10 = a[4] * P.r00 + a[5] * P.r10 + a[6] * P.r20;
    O.r11 = a[4] * P.r01 + a[5] * P.r11 + a[6] * P.r21;
    O.r12 = a[4] * P.r02 + a[5] * P.r12 + a[6] * P.r22;
    O.ty = a[4] * P.tx + a[5] * P.ty + a[6] * P.tz + a[7];
    O.r20 = a[8] * P.r00 + a[9] * P.r10 + a[10] * P.r20;
    O.r21 = a[8] * P.r01 + a[9] * P.r11 + a[10] * P.r21;
    O.r22 = a[8] * P.r02 + a[9] * P.r12 + a[10] * P.r22;
    O.tz = a[8] * P.tx + a[9] * P.ty + a[10] * P.tz + a[11];
    P = O;
}

// P <- P * C for a constant segment read through a wave-uniform table
template <bool T3FMA = false, class CV>
RTB_HD void pose_mul_seg(Pose &P, const CV &cv, int j)
{
    pose_mul_general<T3FMA>(P, [&](int k) { return k < 9 ? cv.seg[j].r[k] : cv.seg[j].t[k - 9]; });
}
// ---------------------------------------------------------------- P <- P * C_j by the segment's STRUCTURE CLASS
// (rtbhip_internal.h: kSeg*; decided on the host from the EXACT zeros and ones of the folded constant -- cos(pi/2) = 6.1e-17 stays what it is).
// Every structured product is pose_rot_const<CLS> -- the general product's fixed operation sequence with the exact-form rewrites of dotk applied
// entry by entry from the class's kind table -- so it returns the general product's bits by construction (tests/test_segment_classes.py replays all
// 13 classes on the host; on the device: the IK outputs of signature, run-time-compiled and general kernels).  Costs (general: 27 + 9):
//   translation   P.t += R C.t: one fused column per NON-ZERO component, in the order of pose_t3_fma                     3 each
//   identity      nothing                                                                                                 0
//   Rx / Ry / Rz  rotation about one axis, the two other columns mix                                                      12
//   ..P / ..N     the quarter turns (off-diagonal +-1; the diagonal keeps its cos(pi/2))                                  9 (3 mul, 3 add, 3 fma)
//   permA / permB the cyclic column permutations of the axis conjugation: register moves                                 0
template <int TM, class CV>
RTB_HD void pose_seg_translate(Pose &P, const CV &cv, int j)
{
    if (TM & 1) { const double x = cv.seg[j].t[0]; P.tx = fmax_(x, P.r00, P.tx); P.ty = fmax_(x, P.r10, P.ty); P.tz = fmax_(x, P.r20, P.tz); }
    if (TM & 2) { const double y = cv.seg[j].t[1]; P.tx = fmax_(y, P.r01, P.tx); P.ty = fmax_(y, P.r11, P.ty); P.tz = fmax_(y, P.r21, P.tz); }
    if (TM & 4) { const double z = cv.seg[j].t[2]; P.tx = fmax_(z, P.r02, P.tx); P.ty = fmax_(z, P.r12, P.ty); P.tz = fmax_(z, P.r22, P.tz); }
}
template <int CLS, class CV>
RTB_HD void pose_seg_rotate(Pose &P, const CV &cv, int j)
{
    pose_rot_const<CLS>(P, [&](int k) { return cv.seg[j].r[k]; });      // entries whose kind is not kAny are never read
}
// compile-time class and translation mask: straight-line code (k_ik's instantiations for known robots: ik_kernels.hip, kIkSig*)
template <int CLS, int TM, class CV>
RTB_HD void pose_mul_seg_sig(Pose &P, const CV &cv, int j)
{
    if (CLS == kSegGeneral) { pose_mul_seg<true>(P, cv, j); return; }
    pose_seg_translate<TM>(P, cv, j);
    pose_seg_rotate<CLS>(P, cv, j);
}
// run-time class (a wave-uniform switch on the descriptor): measured SLOWER than the general product inside k_ik (round 5 visit c: config 3
// 0.986 against 0.939 ms -- the ~45 scalar branches per iteration cost more than the ~170 vector operations they save); kept for the host
// replay's tests and as an A/B switch (RTB_SEG_CLASSES = 1)
template <class CV>
RTB_HD void pose_mul_seg_cls(Pose &P, const CV &cv, int j, int jm)
{
    const int cls = jm_cls(jm), tm = jm_tmask(jm);
    if (cls == kSegGeneral) { pose_mul_seg<true>(P, cv, j); return; }
    if (tm & 1) pose_seg_translate<1>(P, cv, j);
    if (tm & 2) pose_seg_translate<2>(P, cv, j);
    if (tm & 4) pose_seg_translate<4>(P, cv, j);
    switch (cls) {                                        // wave-uniform
    case kSegIdentity: break;
    case kSegRxP: pose_seg_rotate<kSegRxP>(P, cv, j); break;
    case kSegRxN: pose_seg_rotate<kSegRxN>(P, cv, j); break;
    case kSegRx: pose_seg_rotate<kSegRx>(P, cv, j); break;
    case kSegRyP: pose_seg_rotate<kSegRyP>(P, cv, j); break;
    case kSegRyN: pose_seg_rotate<kSegRyN>(P, cv, j); break;
    case kSegRy: pose_seg_rotate<kSegRy>(P, cv, j); break;
    case kSegRzP: pose_seg_rotate<kSegRzP>(P, cv, j); break;
    case kSegRzN: pose_seg_rotate<kSegRzN>(P, cv, j); break;
    case kSegRz: pose_seg_rotate<kSegRz>(P, cv, j); break;
    case kSegPermA: pose_seg_rotate<kSegPermA>(P, cv, j); break;
    default: pose_seg_rotate<kSegPermB>(P, cv, j); break;
    }
}

template <class CV>
RTB_HD void pose_from_seg(Pose &P, const CV &cv, int j)
{
    P.r00 = cv.seg[j].r[0]; P.r01 = cv.seg[j].r[1]; P.r02 = cv.seg[j].r[2];
    P.r10 = cv.seg[j].r[3]; P.r11 = cv.seg[j].r[4]; P.r12 = cv.seg[j].r[5];
    P.r20 = cv.seg[j].r[6]; P.r21 = cv.seg[j].r[7]; P.r22 = cv.seg[j].r[8];
    P.tx = cv.seg[j].t[0]; P.ty = cv.seg[j].t[1]; P.tz = cv.seg[j].t[2];
}

// ---------------------------------------------------------------- the chain walk (run-time n)
// Executes the canonical segment form  C_0 Z_0(q) C_1 ... Z_{n-1}(q) * tail  for ONE lane
// (tail = C_n * tool, folded on the host).
//   qcol(c)       -> joint coordinate column c of this lane's configuration
//   rec(slot, v)  -> per-lane Jacobian scratch write; slot = r*n + j holds p_j (r=0..2) and
//                    z_j (r=3..5), i.e. exactly where row r of column j of the finished J lives.
// WANT_J = false skips the recording (pure fkine).
template <bool WANT_J, class CV, class QCol, class Rec>
RTB_HD void chain_walk(const CV &cv, int n, const double *tail, Pose &P, QCol qcol, Rec rec)
{
    pose_identity(P);
    for (int j = 0; j < n; ++j) {
        if (j == 0) pose_from_seg(P, cv, 0); else pose_mul_seg(P, cv, j);
        const int jm = cv.jmeta[j];
        double eta = qcol(jm_jq(jm));
        if (jm_flip(jm)) eta = -eta;  // methods.cpp:363-366
        if (WANT_J) {
            rec(j, P.tx); rec(n + j, P.ty); rec(2 * n + j, P.tz);
            rec(3 * n + j, P.r02); rec(4 * n + j, P.r12); rec(5 * n + j, P.r22);
        }
        if (jm_prismatic(jm)) {
            pose_tz(P, eta);
        } else {
            double s, c;
            rtb_sincos(eta, &s, &c);  // full fp64 accuracy (trig.h), no fast-math: fknm.cpp:1324-1325
            pose_rotz(P, c, s);
        }
    }
    pose_mul_general(P, [&](int k) { return tail[k]; });
}

// Closes the Jacobian columns in place in the per-lane scratch once the end-effector pose P is
// known: frame 0 -> jacob0, frame 1 -> jacobe (= blkdiag(Re^T, Re^T) jacob0).
//   get(slot) / put(slot, v) : per-lane scratch access.
template <class CV, class Get, class Put>
RTB_HD void jacobian_close(const CV &cv, int n, const Pose &P, int frame, Get get, Put put)
{
    for (int j = 0; j < n; ++j) {
        const int jm = cv.jmeta[j];
        double zx = get(3 * n + j), zy = get(4 * n + j), zz = get(5 * n + j);
        double vx, vy, vz, wx, wy, wz;
        if (!jm_prismatic(jm)) {
            double dx = P.tx - get(j), dy = P.ty - get(n + j), dz = P.tz - get(2 * n + j);
            vx = mix_pm(zy, dz, zz, dy);
            vy = mix_pm(zz, dx, zx, dz);
            vz = mix_pm(zx, dy, zy, dx);
            wx = zx; wy = zy; wz = zz;
        } else {
            vx = zx; vy = zy; vz = zz;
            wx = 0.0; wy = 0.0; wz = 0.0;
        }
        if (jm_flip(jm)) {  // methods.cpp:142-145,172-175
            vx = -vx; vy = -vy; vz = -vz;
            wx = -wx; wy = -wy; wz = -wz;
        }
        if (frame == 1) {
            double a = vx, b = vy, c = vz;
            vx = dot3x(P.r00, a, P.r10, b, P.r20, c);
            vy = dot3x(P.r01, a, P.r11, b, P.r21, c);
            vz = dot3x(P.r02, a, P.r12, b, P.r22, c);
            a = wx; b = wy; c = wz;
            wx = dot3x(P.r00, a, P.r10, b, P.r20, c);
            wy = dot3x(P.r01, a, P.r11, b, P.r21, c);
            wz = dot3x(P.r02, a, P.r12, b, P.r22, c);
        }
        put(j, vx); put(n + j, vy); put(2 * n + j, vz);
        put(3 * n + j, wx); put(4 * n + j, wy); put(5 * n + j, wz);
    }
}

// Writes the 4x4 (row-major, 16 doubles) of pose P through put(k, v), k = 0..15.
template <class Put>
RTB_HD void pose_store16(const Pose &P, Put put)
{
    put(0, P.r00); put(1, P.r01); put(2, P.r02); put(3, P.tx);
    put(4, P.r10); put(5, P.r11); put(6, P.r12); put(7, P.ty);
    put(8, P.r20); put(9, P.r21); put(10, P.r22); put(11, P.tz);
    put(12, 0.0); put(13, 0.0); put(14, 0.0); put(15, 1.0);
}

// One (j, i>=j) block of the Hessian from a finished Jacobian in per-lane scratch
// (methods.cpp:16-32):  H[j,0:3,i] = Jw_j x Jv_i ; H[j,3:6,i] = Jw_j x Jw_i ; mirrored for i != j.
template <class Get, class Put>
RTB_HD void hessian_from_jacobian(int n, Get get, Put put /* put(index into n*6*n, v) */)
{
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const double wjx = get(3 * n + j), wjy = get(4 * n + j), wjz = get(5 * n + j);
#pragma unroll
        for (int i = j; i < n; ++i) {
            const double vx = get(i), vy = get(n + i), vz = get(2 * n + i);
            const double wx = get(3 * n + i), wy = get(4 * n + i), wz = get(5 * n + i);
            const double ax = wjy * vz - wjz * vy, ay = wjz * vx - wjx * vz, az = wjx * vy - wjy * vx;
            const double bx = wjy * wz - wjz * wy, by = wjz * wx - wjx * wz, bz = wjx * wy - wjy * wx;
            put((j * 6 + 0) * n + i, ax); put((j * 6 + 1) * n + i, ay); put((j * 6 + 2) * n + i, az);
            put((j * 6 + 3) * n + i, bx); put((j * 6 + 4) * n + i, by); put((j * 6 + 5) * n + i, bz);
            if (i != j) {
                put((i * 6 + 0) * n + j, ax); put((i * 6 + 1) * n + j, ay); put((i * 6 + 2) * n + j, az);
                put((i * 6 + 3) * n + j, 0.0); put((i * 6 + 4) * n + j, 0.0); put((i * 6 + 5) * n + j, 0.0);
            }
        }
    }
}

// One entry H[j, row, i] of the Hessian from a finished Jacobian stored as Jrow[r * n + c]
// (methods.cpp:16-32: for j <= i  (w_j x v_i ; w_j x w_i), mirrored translational block, zero
// rotational block below the diagonal).
RTB_HD double hessian_entry(const double *Jrow, int n, int j, int row, int i)
{
    const bool up = j <= i;
    if (row >= 3 && !up) return 0.0;
    const int ac = up ? j : i, bc = up ? i : j;
    const int k = row >= 3 ? row - 3 : row;
    const int k1 = k == 2 ? 0 : k + 1, k2 = k == 0 ? 2 : k - 1;
    const int bo = row >= 3 ? 3 : 0;
    const double a1 = Jrow[(3 + k1) * n + ac], a2 = Jrow[(3 + k2) * n + ac];
    const double b1 = Jrow[(bo + k1) * n + bc], b2 = Jrow[(bo + k2) * n + bc];
    return a1 * b2 - a2 * b1;
}

// The tile's Hessians are one contiguous run of ncfg * n*6*n doubles; lane l produces the 16-byte
// pieces l, l+64, ... of that run straight from the wave's staged Jacobians (jl: row stride jstride),
// so the (N,n,6,n) output is written fully coalesced with no second staging buffer.
template <int NJ, class Store>
RTB_HD void hessian_run(const double *jl, int jstride, int ncfg, int lane, Store store /* store(f, a, b, both) */)
{
    constexpr int HW = NJ * 6 * NJ, BW = 6 * NJ;
    const int total = ncfg * HW;
    // (r, j, row, i) of entry f = 2*lane, then advanced by the wave's stride of 128 entries with carries --
    // compile-time increments instead of six divisions per piece
    constexpr int STEP = 2 * kWave;
    constexpr int dR = STEP / HW, remR = STEP % HW, dJ = remR / BW, remJ = remR % BW, dRow = remJ / NJ, dI = remJ % NJ;
    int f = 2 * lane;
    int r = f / HW, w = f - r * HW;
    int j = w / BW, w2 = w - j * BW;
    int row = w2 / NJ, i = w2 - row * NJ;
    for (; f < total; f += STEP) {
        const double *Jr = jl + r * jstride;
        const double a = hessian_entry(Jr, NJ, j, row, i);
        // the neighbour entry f + 1: next column, wrapping into the next row / block / configuration
        int i2 = i + 1, row2 = row, j2 = j, r2 = r;
        if (i2 == NJ) { i2 = 0; if (++row2 == 6) { row2 = 0; if (++j2 == NJ) { j2 = 0; ++r2; } } }
        const bool both = f + 1 < total;
        const double b = both ? hessian_entry(jl + r2 * jstride, NJ, j2, row2, i2) : 0.0;
        store(f, a, b, both);
        i += dI; if (i >= NJ) { i -= NJ; ++row; }
        row += dRow; if (row >= 6) { row -= 6; ++j; }
        j += dJ; if (j >= NJ) { j -= NJ; ++r; }
        r += dR;
    }
}

// Contiguous run of ncfg rows of W doubles (W even, row stride `stride` in the staging buffer) handed out as
// 16-byte pieces: lane l gets pieces l, l+64, ...; (row, column) advance by compile-time steps with a carry.
template <int W, class Store>
RTB_HD void flush_rows(const double *rows, int stride, int ncfg, int lane, Store store /* store(f, a, b) */)
{
    static_assert(W % 2 == 0, "pieces must not straddle rows");
    constexpr int STEP = 2 * kWave, dR = STEP / W, dE = STEP % W;
    const int total = ncfg * W;
    int f = 2 * lane;
    int r = f / W, e = f - r * W;
    for (; f < total; f += STEP) {
        const double *p = rows + r * stride + e;
        store(f, p[0], p[1]);
        e += dE; r += dR;
        if (e >= W) { e -= W; ++r; }
    }
}

}  // namespace rtbhip
