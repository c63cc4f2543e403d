// kin_reg.h -- register-resident variant of the fused fkine + Jacobian tile for chains with a
// compile-time joint count NJ (1..8: every arm of BASELINE.json's configs except the 14-DOF YuMi).
//
// Why a second variant: the run-time-n tile (kin_tile.h) keeps the per-joint (p_j, z_j) scratch in
// LDS because a register array cannot be indexed by a run-time joint number.  That costs 8(6n+qw)
// bytes of LDS per lane -- 25 KB per wave for the Panda -- which caps a CU at 6 waves and leaves the
// kernel latency-bound (first MI355X measurement: 0.178 ms / 1e6 configs = 36 % of the HBM roof,
// ~28k cycles of mostly-stalled wave lifetime).  With NJ a template parameter the canonical
// segment walk is one straight-line block -- no branches, no interpreter -- in which
//   * all NJ sin/cos pairs are evaluated up front (NJ independent dependency chains for the
//     scheduler to interleave) before the pose-dependent products start;
//   * p_j, z_j, sin_j, cos_j are named registers; q is read straight into registers;
//   * LDS is only the output transposer, time-shared in rounds of 32 lanes for J and one round of
//     64 for T: 32*(6n+1)*8 B = 11 KB per wave for n = 7  =>  occupancy is set by VGPRs, not LDS.
#pragma once
#include "kin_tile.h"
#include "trig.h"

namespace rtbhip {

constexpr int kRegMaxJoints = 8;    // register-resident consumers (IK, Hessian, jacob_dot, manipulability ...)
constexpr int kIkMaxJoints = 16;    // IK: chains of 9..12 joints run at one wave per SIMD (the whole 512-register budget); 13..16 also compile
                                    // -- their normal equations no longer fit the register file and spill to scratch: slow, but served
constexpr int kKinRegMax = 10;      // fkine / Jacobian tiles: 9 and 10 joints still fit the register file at 2 waves per SIMD
#ifndef RTB_JROUND
#define RTB_JROUND 32
#endif
constexpr int kJRound = RTB_JROUND;  // lanes staged per J round

RTB_HD int reg_lds_doubles(int n)
{
    const int a = kJRound * (6 * n + 1), b = kWave * 17;
    return a > b ? a : b;
}

// packed (T | J) rows: lanes staged per round -- each round holds kPRound x (17 + 6n + 1) doubles of LDS (n = 7, 16 lanes: 7.7 KB per wave)
#ifndef RTB_PROUND
#define RTB_PROUND 16
#endif
constexpr int kPRound = RTB_PROUND;
RTB_HD int reg_lds_doubles_packed(int n) { return kPRound * (17 + 6 * n + 1); }

// Per-lane core: joint values qv[] (chain order, as the caller holds them) -> P = C_0 Z_0 ... tail and
// the finished Jacobian in registers.  jac slot r*NJ + j : rows 0..2 = p_j, rows 3..5 = z_j until
// the closing loop finishes them.  Used by the tile kernel (reg_compute) and by the IK loop.
#ifndef RTB_POSE_T3_FMA
#define RTB_POSE_T3_FMA 1
#endif
#ifndef RTB_SEG_CLASSES
#define RTB_SEG_CLASSES 0      // 1: k_ik multiplies by a constant segment through a RUN-TIME switch on its structure class (kin_device.h: pose_mul_seg_cls).
#endif                         //    Measured slower than the general product (round 5 visit c); what ships is the COMPILE-TIME form below.
// A chain's STRUCTURE SIGNATURE: 7 bits per constant segment C_0 .. C_n (class | translation mask << 4, rtbhip_internal.h: kSeg*), bit 63 = "present".
// reg_core<..., SIG != 0> multiplies by every segment through pose_mul_seg_sig<class, mask>: straight-line code, no descriptor is read.  A
// kernel instantiated for a signature serves exactly the chains whose table has it (ik_kernels.hip: the launcher compares).
// (SegSig and its accessors: rtbhip_internal.h)
inline SegSig chain_signature(const int32_t *jmeta, int n)      // host: from the descriptors chain.cpp wrote (n joints + the tail's word)
{
    if (n > 8) return 0;
    SegSig s = kSegSigPresent;
    for (int j = 0; j <= n; ++j) s |= seg_sig_of(j, jm_cls(jmeta[j]), jm_tmask(jmeta[j]));
    return s;
}
template <SegSig SIG, int J, class CV>
RTB_HD void pose_mul_seg_by_sig(Pose &P, const CV &cv) { pose_mul_seg_sig<seg_sig_cls(SIG, J), seg_sig_tm(SIG, J)>(P, cv, J); }
#ifndef RTB_SIG_KEEP_PINS
#define RTB_SIG_KEEP_PINS 0    // 1: signature kernels keep the load pins and every fence of the general plain walk (A/B)
#endif
#ifndef RTB_PIN_SEG_LOADS
#define RTB_PIN_SEG_LOADS 1
#endif
// a chain view may carry a pointer to the sincos constants (k_ik: see sincos_reduced_tab in trig.h); the others use the literals
template <class CV, class = void> struct cv_has_trig { static constexpr bool value = false; };
template <class CV> struct cv_has_trig<CV, decltype((void)(((const CV *)nullptr)->trig))> { static constexpr bool value = true; };
// ... or ask for the fused-chain translation  t += R c  (pose_t3_fma: the form every structured segment product is an instance of) without carrying the
// sincos table: k_kin_diff's views (diff_kernel.h), so that its structure instantiations return its general kernel's bits
template <class CV, class = void> struct cv_t3fma { static constexpr bool value = false; };
template <class CV> struct cv_t3fma<CV, typename CV::t3fma_tag> { static constexpr bool value = true; };

// joint J of the walk (compile-time index: a signature picks the segment's form by it)
template <int NJ, bool WANT_J, bool PLAIN, SegSig SIG, int J, class CV>
RTB_HD void reg_walk_step(const CV &cv, Pose &P, double (&jac)[6 * NJ], const int (&jmv)[NJ], const double (&c)[NJ], const double (&s)[NJ], const double (&d)[NJ])
{
    constexpr int j = J;
    {
#if defined(__HIP_DEVICE_COMPILE__) && RTB_PIN_SEG_LOADS
        // Inside k_ik's persistent loop (the chain views that carry `trig`): tie segment j's table pointer to a value of step j - 1, so that its
        // scalar loads cannot be issued before the walk gets there.  Without this the loads of the later segments were issued early and their
        // results parked in VGPR lanes (v_writelane) until needed (v_readlane): 475 -> 251 such instructions in the kernel, 253 -> 244 VGPRs,
        // -2.4 % (config 3) ... -3.6 % (notebook setting) on one box (round 4 visit l).  RTB_PIN_SEG_LOADS = 2 pins the general (branchy) walk too.
        CV cvj = cv;
        // (a signature kernel reads two or three scalars per segment instead of twelve: there the pin and the fences only cost -- round 5
        // visits f, k, three interleaved rounds each: config 3 0.844 -> 0.833 -> 0.830 ms without them, outputs bit-identical)
        if ((PLAIN || RTB_PIN_SEG_LOADS > 1) && (SIG == 0 || RTB_SIG_KEEP_PINS) && cv_has_trig<CV>::value && j > 0) asm volatile("" : "+s"(cvj.seg), "+v"(P.tx));
        if (j == 0) pose_from_seg(P, cvj, 0);
        else if constexpr (SIG != 0) pose_mul_seg_by_sig<SIG, J>(P, cvj);                                              // k_ik for a known robot: compile-time class
        else if constexpr (RTB_SEG_CLASSES && cv_has_trig<CV>::value) pose_mul_seg_cls(P, cvj, j, cvj.jmeta[j]);      // k_ik: by structure class (run-time switch, A/B)
        else pose_mul_seg<((PLAIN && RTB_POSE_T3_FMA && cv_has_trig<CV>::value) || cv_t3fma<CV>::value)>(P, cvj, j);
#else
        if (j == 0) pose_from_seg(P, cv, 0);
        else if constexpr (SIG != 0) pose_mul_seg_by_sig<SIG, J>(P, cv);
        else if constexpr (RTB_SEG_CLASSES && cv_has_trig<CV>::value) pose_mul_seg_cls(P, cv, j, cv.jmeta[j]);          // (the host replay of k_ik: the same arithmetic)
        else pose_mul_seg<cv_t3fma<CV>::value>(P, cv, j);
#endif
        if (WANT_J) {
            jac[j] = P.tx; jac[NJ + j] = P.ty; jac[2 * NJ + j] = P.tz;
            jac[3 * NJ + j] = P.r02; jac[4 * NJ + j] = P.r12; jac[5 * NJ + j] = P.r22;
        }
        // revolute: rotate by (c, s); prismatic: slide d -- a wave-uniform branch on the joint descriptor (s_cbranch)
        if (jm_prismatic(jmv[j])) pose_tz(P, d[j]);
        else pose_rotz(P, c[j], s[j]);
#if defined(RTB_PLAIN_FENCE_EVERY)
        if (!PLAIN || (j % RTB_PLAIN_FENCE_EVERY) == RTB_PLAIN_FENCE_EVERY - 1) sched_fence();      // A/B: fewer fences in the straight-line walk
#else
        if (SIG == 0 || RTB_SIG_KEEP_PINS) sched_fence();          // (signature kernels: no fence -- visit k: 0.8337 -> 0.8302 ms, every second one 0.8337)
#endif
    }
}
template <int NJ, bool WANT_J, bool PLAIN, SegSig SIG, int J = 0, class CV>
RTB_HD void reg_walk_steps(const CV &cv, Pose &P, double (&jac)[6 * NJ], const int (&jmv)[NJ], const double (&c)[NJ], const double (&s)[NJ], const double (&d)[NJ])
{
    if constexpr (J < NJ) {
        reg_walk_step<NJ, WANT_J, PLAIN, SIG, J>(cv, P, jac, jmv, c, s, d);
        reg_walk_steps<NJ, WANT_J, PLAIN, SIG, J + 1>(cv, P, jac, jmv, c, s, d);
    }
}

template <int NJ, bool WANT_J>
RTB_HD void reg_close_jacobian(const Pose &P, int frame, const int (&jmv)[NJ], double (&jac)[6 * NJ])
{
    if (WANT_J) {
#pragma clang fp contract(off)
        // Jv = z x (p_e - p), Jw = z (revolute) ; Jv = z, Jw = 0 (prismatic); flip negates
        // (methods.cpp:142-195); frame 1 rotates both halves by Re^T.
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            double zx = jac[3 * NJ + j], zy = jac[4 * NJ + j], zz = jac[5 * NJ + j];
            if (jm_flip(jmv[j])) { zx = -zx; zy = -zy; zz = -zz; }          // wave-uniform
            double vx, vy, vz, wx, wy, wz;
            if (jm_prismatic(jmv[j])) {                                      // wave-uniform
                vx = zx; vy = zy; vz = zz;
                wx = 0.0; wy = 0.0; wz = 0.0;
            } else {
                const double dx = P.tx - jac[j], dy = P.ty - jac[NJ + j], dz = P.tz - jac[2 * NJ + j];
                vx = mix_pm(zy, dz, zz, dy); vy = mix_pm(zz, dx, zx, dz); vz = mix_pm(zx, dy, zy, dx);      // (written out: kin_device.h, mix_pp)
                wx = zx; wy = zy; wz = zz;
            }
            if (frame == 1) {
                double a = vx, b = vy, e = vz;
                vx = dot3x(P.r00, a, P.r10, b, P.r20, e);
                vy = dot3x(P.r01, a, P.r11, b, P.r21, e);
                vz = dot3x(P.r02, a, P.r12, b, P.r22, e);
                a = wx; b = wy; e = wz;
                wx = dot3x(P.r00, a, P.r10, b, P.r20, e);
                wy = dot3x(P.r01, a, P.r11, b, P.r21, e);
                wz = dot3x(P.r02, a, P.r12, b, P.r22, e);
            }
            jac[j] = vx; jac[NJ + j] = vy; jac[2 * NJ + j] = vz;
            jac[3 * NJ + j] = wx; jac[4 * NJ + j] = wy; jac[5 * NJ + j] = wz;
        }
    }
}

// PLAIN (compile-time): every joint is revolute and none is flipped (the caller checked the chain's descriptors): no descriptor is read, no
// wave-uniform branch splits the walk -- one straight-line block from the first sine to the last Jacobian column.
template <int NJ, bool WANT_J, bool PLAIN = false, SegSig SIG = 0, class CV, class TL>
RTB_HD void reg_core(const CV &cv, TL tail /* tail[k], k = 0..11 */, int frame, const double (&qv)[NJ], Pose &P,
                     double (&jac)[6 * NJ])
{
    static_assert(SIG == 0 || (PLAIN && NJ <= 8), "a structure signature goes with the straight-line walk");
    double c[NJ], s[NJ], d[NJ];
    // Wave-uniform per-joint blend weights instead of per-lane selects: rv = 1 for a revolute joint,
    // pv = 1 for a prismatic one, sg = -1 where the joint is flipped.  They are re-derived from the
    // one-dword joint descriptor at every use (a few SALU ops) rather than kept as 3*NJ SGPR doubles
    // across the whole walk: inside the persistent IK loop those 42 SGPRs were what overflowed the
    // scalar register file.
    int jmv[NJ];
    bool big = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {   // methods.cpp:363-366 for flip
        jmv[j] = PLAIN ? 0 : cv.jmeta[j];
        d[j] = PLAIN ? qv[j] : qv[j] * (jm_flip(jmv[j]) ? -1.0 : 1.0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {   // NJ independent evaluations, one basic block
        if constexpr (cv_has_trig<CV>::value) sincos_reduced_tab(d[j], s[j], c[j], cv.trig);
        else sincos_reduced(d[j], s[j], c[j]);
        big = big || !(fabs(d[j]) < kTrigFastLimit);
    }
    if (wave_any(big)) {             // |q| >= 2^20, NaN, inf: library path for the whole wave
#pragma unroll
        for (int j = 0; j < NJ; ++j) sincos(d[j], &s[j], &c[j]);
    }
    reg_walk_steps<NJ, WANT_J, PLAIN, SIG>(cv, P, jac, jmv, c, s, d);
    // k_ik's chain views: the tail IS segment NJ of the table (no tool in IK), descriptor NJ carries its class
    if constexpr (SIG != 0) pose_mul_seg_by_sig<SIG, NJ>(P, cv);
    else if constexpr (RTB_SEG_CLASSES && cv_has_trig<CV>::value) pose_mul_seg_cls(P, cv, NJ, cv.jmeta[NJ]);
    else pose_mul_general<((PLAIN && RTB_POSE_T3_FMA && cv_has_trig<CV>::value) || cv_t3fma<CV>::value)>(P, [&](int k) { return tail[k]; });
    sched_fence();
    reg_close_jacobian<NJ, WANT_J>(P, frame, jmv, jac);
}

// whole per-lane compute of one tile: q row in memory -> (P, J)
// SIG != 0: the chain's structure signature (all joints revolute, none flipped, no tool: the tail is segment NJ of the table) -- the straight-line walk
template <int NJ, bool WANT_J, SegSig SIG = 0, class CV>
RTB_HD void reg_compute(const KinParams &kp, const CV &cv, const double *__restrict__ q, int64_t cfg,
                        Pose &P, double (&jac)[6 * NJ])
{
    const bool live = cfg < kp.N;
    const double *qrow = q + cfg * kp.qw;
    double qv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) qv[j] = live ? qrow[jm_jq(cv.jmeta[j])] : 0.0;
    if constexpr (SIG != 0) reg_core<NJ, WANT_J, true, SIG>(cv, &cv.seg[NJ].r[0], kp.frame, qv, P, jac);
    else reg_core<NJ, WANT_J>(cv, kp.tail, kp.frame, qv, P, jac);
}

// staging: lane writes its finished J row / its 4x4 into the wave's LDS transposer
template <int NJ>
RTB_HD void reg_stage_J(const double (&jac)[6 * NJ], double *buf, int slot_lane)
{
    double *mine = buf + slot_lane * (6 * NJ + 1);
#pragma unroll
    for (int k = 0; k < 6 * NJ; ++k) mine[k] = jac[k];
}

// packed rows: the lane's 4x4 (base applied) into the T area, its J into the J area of the same round (kin_flush_packed)
template <int NJ>
RTB_HD void reg_stage_packed(const KinParams &kp, Pose P, const double (&jac)[6 * NJ], double *bufT, double *bufJ, int slot_lane)
{
    if (kp.has_base) pose_premul(P, kp.base);
    double *mt = bufT + slot_lane * 17;
    pose_store16(P, [&](int k, double v) { mt[k] = v; });
    double *mj = bufJ + slot_lane * (6 * NJ + 1);
#pragma unroll
    for (int k = 0; k < 6 * NJ; ++k) mj[k] = jac[k];
}

RTB_HD void reg_stage_T(const KinParams &kp, Pose P, double *buf, int lane)
{
    if (kp.has_base) pose_premul(P, kp.base);
    double *mine = buf + lane * 17;
    pose_store16(P, [&](int k, double v) { mine[k] = v; });
}

}  // namespace rtbhip
