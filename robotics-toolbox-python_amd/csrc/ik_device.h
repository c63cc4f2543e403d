// ik_device.h -- per-lane Levenberg-Marquardt inverse kinematics, the whole search loop resident on
// the device.
//
// Replaces _IK_loop + _IK_LM_Chan/Wampler/Sugihara (core/ik.cpp:19-75,157-209), _angle_axis
// (ik.cpp:241-286), _rand_q (ik.cpp:288-299), _check_lim (ik.cpp:227-239) and, as "flavour 1", the
// Python solver behind ikine_LM (robot/IK.py:297-367 `_solve`, :994-1017 `IK_LM.step`).
//
// One lane runs one SEARCH of one target at a time.  The reference's nested while-loops are restated
// as a per-lane transition (`ik_iter`) that performs exactly ONE LM iteration: every lane of a wave
// executes the same instruction stream (FK + Jacobian + 6-vector error + normal equations + solve)
// regardless of which target / search / iteration it is in, and a per-wave scheduler (below) keeps
// the lanes fed: fresh targets while there are any, then speculative later searches of the wave's
// unresolved targets.
//
// Differences from the reference that cannot be bit-matched and are covered by statistical parity
// (SURVEY.md 8c): restarts come from a counter-based generator keyed by (seed, target, draw, joint)
// instead of an unseeded std::rand; the damped normal equations are solved by an LDL^T
// factorisation in registers (A = J^T W J + wn I is symmetric positive definite for wn > 0) instead
// of forming the explicit inverse with a pivoted LU.
#pragma once
#include "kin_reg.h"
#include "ldl.h"
#include "diff_device.h"

// Two compile-time A/B switches, measured against each other on one box in the steady state (round 4 visit h, scripts/gpu_r4_h.sh: four builds,
// three interleaved rounds; (success, iterations, searches) of all 1e5 / 1e6 targets identical in every build):
//                              config 3 (1e5)          notebook setting       1e6 targets
//   neither                    1.105 1.098 1.102 ms    0.380 0.381 0.382      6.13 6.10 6.17
//   RTB_IK_FAST_RCP only       1.082 1.085 1.082       0.375 0.374 0.375      6.02 6.00 6.08       <- shipped (-1.7 %)
//   RTB_IK_FMOD_FMA only       1.120 1.117 1.112       0.382 0.382 0.381      6.25 6.09 6.20       (+1.3 %: fewer instructions per wrap, but the
//   both                       1.102 1.099 1.099       0.376 0.374 0.375      6.12 6.22 6.19        extra branch lengthens the scheduling pass)
#ifndef RTB_IK_FAST_RCP
#define RTB_IK_FAST_RCP 1       // the LM step's seven pivot reciprocals through rcp_pivot (ldl.h): ~60 VALU instructions of 1633 per iteration
#endif
#ifndef RTB_IK_FMOD_FMA
#define RTB_IK_FMOD_FMA 0       // 1: the end-of-search wrap through an exact FMA remainder instead of the library fmod (ik_fmod_2pi, below)
#endif
#ifndef RTB_IK_UNITW
#define RTB_IK_UNITW 0          // 1: a second, multiplication-free copy of the LM step for a mask of ones.  Measured (round 3, visit y, four builds on
#endif                          // one box): the copy costs 30 VGPRs (223 -> 256 + scratch) and 4-8 % of every IK line; the 48 products it saves do not pay

namespace rtbhip {

constexpr double kIkPi = 3.14159265358979323846264338327950288;   // linalg.h:19
constexpr double kIkPi2 = 6.283185307179586;                      // linalg.h:20
constexpr double kIkPiHalf = 1.57079632679489661923132169163975144;

struct IkDev {   // wave-uniform solver parameters (kernarg)
    int32_t ilimit, slimit, reject_jl, method, flavour, has_q0;
    int32_t fresh_cap, pool_chunk;   // scheduler: fresh targets a wave may start per pass / reserves per refill
    int32_t pass_mask, spec_policy;  // scheduler: the pass runs on iterations with (tick & pass_mask) == 0; 0 round-robin / 1 failure-weighted speculation
    int32_t unit_we, pad_we;         // every we[k] == 1 (the default mask): W J and W e need no products -- the same bits, 48 multiplies fewer per iteration
    double tol, lambda;
    double we[6];
    uint64_t seed;
    int64_t N;
    double kq, km, ps;               // null-space terms of the Python solvers (IK.py:507-576); kq <= 0: none
    double pi[RTBHIP_MAX_JOINTS];    // ... the influence distance, one per joint (IK.py:507-540 and :1441-1442 accept a scalar or an array)
    double ks;                       // IK_QP (method 5): slack gain; its joint-velocity gain kj travels in `lambda`
    int64_t target0;                 // added to a target's row number where it keys the restart generator (rtbhip_ik_target_base):
                                     // a row block of a larger batch then draws what the whole batch would have drawn for those targets
    // flat schedule (below, "one launch, every search range cut into chunks"): flat_chunks > 0 switches it on; then N above counts
    // work items (flat_n targets x flat_chunks), item v = chunk v / flat_n of target v % flat_n
    int32_t flat_chunks, flat_l0, flat_len;
    uint32_t flat_n;
    int32_t *flat_done;              // per target: index of the lowest chunk that has SUCCEEDED so far (kIkFlatNone: none yet); device memory
    unsigned long long *stats;       // diagnostics (RTBHIP_IK_STATS): 6 words per wave (ik_kernels.hip: kIkStatWords) -- loop iterations, scheduling passes, lane-iterations
                                     // spent on a running search, work items started; NULL in normal runs
};

// ---------------------------------------------------------------- restart generator
// uniform in [lo, hi): counter-based (splitmix64 finaliser), identical on host and device.
RTB_HD double ik_uniform(uint64_t seed, int64_t target, int draw, int joint)
{
    uint64_t z = seed ^ ((uint64_t)target * 0xD1B54A32D192ED03ull);
    z += 0x9E3779B97F4A7C15ull * (((uint64_t)(uint32_t)draw << 8) + (uint64_t)joint + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <int NJ, class QL>
RTB_HD void ik_restart(uint64_t seed, int64_t target, int draw, QL qlim, double (&q)[NJ])
{
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double lo = qlim[j], hi = qlim[NJ + j];
        q[j] = lo + ik_uniform(seed, target, draw, j) * (hi - lo);   // ik.cpp:293-296
    }
}

// ---------------------------------------------------------------- pose error (ik.cpp:241-286)
// Te = current pose P, Tep = {R row-major (9), t (3)}.
template <class TD>
RTB_HD void ik_angle_axis(const Pose &P, TD Td, double (&e)[6])
{
#pragma clang fp contract(off)
    // (every sum of products written out -- kin_device.h, mix_pp: the pose comes out of a structure instantiation's walk or out of the general one,
    // and what the compiler would make of `a b + c d + e f` depends on which)
    e[0] = Td(9) - P.tx; e[1] = Td(10) - P.ty; e[2] = Td(11) - P.tz;
    // R = Rd * Re^T ; only the entries the formula reads
    const double r00 = dot3x(Td(0), P.r00, Td(1), P.r01, Td(2), P.r02);
    const double r01 = dot3x(Td(0), P.r10, Td(1), P.r11, Td(2), P.r12);
    const double r02 = dot3x(Td(0), P.r20, Td(1), P.r21, Td(2), P.r22);
    const double r10 = dot3x(Td(3), P.r00, Td(4), P.r01, Td(5), P.r02);
    const double r11 = dot3x(Td(3), P.r10, Td(4), P.r11, Td(5), P.r12);
    const double r12 = dot3x(Td(3), P.r20, Td(4), P.r21, Td(5), P.r22);
    const double r20 = dot3x(Td(6), P.r00, Td(7), P.r01, Td(8), P.r02);
    const double r21 = dot3x(Td(6), P.r10, Td(7), P.r11, Td(8), P.r12);
    const double r22 = dot3x(Td(6), P.r20, Td(7), P.r21, Td(8), P.r22);
    const double lx = r21 - r12, ly = r02 - r20, lz = r10 - r01;
    const double nrm = sqrt(dot3x(lx, lx, ly, ly, lz, lz));
    const double tr = r00 + r11 + r22;
    if (nrm < 1e-6) {
        if (tr > 0) {
            e[3] = 0.0; e[4] = 0.0; e[5] = 0.0;
        } else {
            e[3] = kIkPiHalf * (r00 + 1); e[4] = kIkPiHalf * (r11 + 1); e[5] = kIkPiHalf * (r22 + 1);
        }
    } else {
        const double k = atan2(nrm, tr - 1) / nrm;
        e[3] = k * lx; e[4] = k * ly; e[5] = k * lz;
    }
}

// E = e^T W e / 2 (ik.cpp:46), one fixed chain of fused multiply-adds
template <bool UNITW, class W>
RTB_HD double ik_half_weighted_square(const double (&e)[6], W we)
{
#pragma clang fp contract(off)
    double E = UNITW ? e[0] * e[0] : (e[0] * we[0]) * e[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) E = UNITW ? __builtin_fma(e[k], e[k], E) : __builtin_fma(e[k] * we[k], e[k], E);
    return E * 0.5;
}

// ---------------------------------------------------------------- one LM step
// dq = (J^T W J + wn I)^-1 J^T W e, J in registers (slot r*NJ + j), W = diag(we).
// The same step through the 6 x 6 system (unit weights, more joints than task dimensions):  (J^T J + wn 1)^-1 J^T e  =  J^T (J J^T + wn 1)^-1 e
// for wn > 0 (push-through identity).  For the 7-joint arm: 21 entries of 7 products instead of 28 of 6, a 6 x 6 factorisation instead of a
// 7 x 7 one -- 66 instructions fewer.  wn == 0 (a caller's k = 0) keeps the n x n form: there the reference inverts a singular matrix and the
// two forms would disagree about the garbage.  MEASURED SLOWER and therefore off (round 4 visit o, one box, three interleaved rounds, sustained:
// config 3 0.961 -> 1.004 ms, 1e6 targets 5.22 -> 5.52 ms): fewer instructions, but J must stay alive through the 6 x 6 solve for the final
// J^T y, and the solve is one long dependent chain where the n x n form's right-hand side is formed alongside the matrix.  Kept as an A/B switch.
#ifndef RTB_IK_DUAL
#define RTB_IK_DUAL 0
#endif
template <int NJ>
RTB_HD void ik_lm_step_dual(const double (&jac)[6 * NJ], const double (&e)[6], double wn, double (&dq)[NJ])
{
    double B[6][6], y[6], g[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        g[r] = e[r];
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < NJ; ++k) a += jac[r * NJ + k] * jac[c * NJ + k];
            B[r][c] = (r == c) ? a + wn : a;
        }
    }
    ldl_solve<6, RTB_IK_FAST_RCP != 0>(B, g, y);
    sched_fence();
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) a += jac[r * NJ + k] * y[r];
        dq[k] = a;
    }
}

template <int NJ, bool UNITW = false, class W>
RTB_HD void ik_lm_step(const double (&jac)[6 * NJ], const double (&e)[6], W we /* we[k], k < 6 */, double wn,
                       double (&dq)[NJ])
{
#pragma clang fp contract(off)
    double A[NJ][NJ];   // lower triangle used; after factorisation holds L (unit diagonal implied)
    double g[NJ];
    double we_e[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) we_e[k] = UNITW ? e[k] : we[k] * e[k];       // (x * 1.0 == x exactly: the unit-weight form returns the same bits)
    // row r of W J once (6 products), then each of its normal-equation entries is one rounded product and 5 fused multiply-adds on it, k = 0 .. 5
    // in that order (written out: nothing for the compiler to decide); formed 7 times instead of 28
#pragma unroll
    for (int r = 0; r < NJ; ++r) {
        double wjr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) wjr[k] = UNITW ? jac[k * NJ + r] : jac[k * NJ + r] * we[k];
        double s = jac[r] * we_e[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) s = __builtin_fma(jac[k * NJ + r], we_e[k], s);
        g[r] = s;
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double a = wjr[0] * jac[c];
#pragma unroll
            for (int k = 1; k < 6; ++k) a = __builtin_fma(wjr[k], jac[k * NJ + c], a);
            A[r][c] = (r == c) ? a + wn : a;
        }
    }
    sched_fence();   // J is dead from here on: do not let the factorisation overlap the products above
    ldl_solve<NJ, RTB_IK_FAST_RCP != 0>(A, g, dq);
}

// ---------------------------------------------------------------- Gauss-Newton / Newton-Raphson steps
// _IK_GN (ik.cpp:79-120):  dq solves (J^T W J) dq = J^T W e  through an SVD (use_pinv) or a column-pivoted QR.
// _IK_NR (ik.cpp:122-156): dq = J^+ e with the damped pseudo-inverse V diag(s/(s^2+d^2)) U^T (ik.cpp:211-226),
//                          or J^-1 e for a 6-joint arm without pinv.
// Both are minimum-norm solutions of J dq = e; with J of full row rank they are  dq = J_a^T (J_a J_a^T + d^2 I)^-1 e_a
// exactly (the damped pseudo-inverse identity; for GN the positive weights drop out of the minimum-norm
// solution, rows with a zero weight are simply absent).  That 6x6 symmetric system is solved in registers.
// Deviation, covered by the statistical IK acceptance of SURVEY 8c: at a rank-deficient J the reference's
// SVD truncates a singular value (GN) or divides by s^2 = 0 (NR), and GN's QR branch on a redundant arm
// returns a basic rather than the minimum-norm solution; the search simply fails or restarts here.
// When the used rows outnumber the joints (a 4- or 5-joint arm with a full mask) J_a has full COLUMN rank instead and
// J_a J_a^T is singular; the pseudo-inverse step is then the least-squares one,  (J_a^T W J_a + d^2 I) dq = J_a^T W e_a
// (W = the mask weights for GN -- they matter here --, 1 for NR; what the reference's SVD / damped pseudo-inverse return),
// an n x n system: ik_lm_step with the damping in place of the LM term.
template <int NJ, class W>
RTB_HD void ik_pinv_step(const double (&jac)[6 * NJ], const double (&e)[6], int rows, double d2 /* damping squared */, W we, bool weighted, double (&dq)[NJ])
{
    if constexpr (NJ < 6) {
        int used = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) used += (rows >> r) & 1;
        if (used > NJ) {                                       // wave-uniform
            double w[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) w[r] = ((rows >> r) & 1) ? (weighted ? we[r] : 1.0) : 0.0;
            ik_lm_step<NJ>(jac, e, &w[0], d2, dq);
            return;
        }
    }
    double B[6][6], y[6], g[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const bool ur = (rows >> r) & 1;
        g[r] = ur ? e[r] : 0.0;
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const bool uc = (rows >> c) & 1;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NJ; ++k) s += jac[r * NJ + k] * jac[c * NJ + k];
            B[r][c] = (ur && uc) ? (r == c ? s + d2 : s) : (r == c ? 1.0 : 0.0);
        }
    }
    ldl_solve<6>(B, g, y);
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) s += jac[r * NJ + k] * y[r];
        dq[k] = s;
    }
}

// ---------------------------------------------------------------- null-space motion of the Python solvers
// _calc_qnull (robot/IK.py:542-576) with _null_Sigma (:507-539): the gradient  -Sigma / kq  (joint-limit avoidance inside
// the influence distance pi, minimum distance ps)  +  jacobm(q) / km  (manipulability), projected with I - pinv(J) J.
// As in the reference the projection -- and with it the whole term -- is applied only when kq > 0 (its guard reads
// `lambda_Sigma > 0 or lambda_Sigma > 0`).  For a J of full row rank  I - pinv(J) J = I - J^T (J J^T)^-1 J: the 6x6
// factorisation is shared with jacobm.  Chains of fewer than 6 joints (projector zero away from singularities, not at
// them) are refused by launch_ik when kq > 0 and never come here.
template <int NJ, class PD, class QL, class QA>
RTB_HD void ik_qnull(const double (&jac)[6 * NJ], const PD &p, QL qlim, QA qa, double (&qn)[NJ])
{
    static_assert(NJ >= 6, "null-space motion needs a redundant or square arm");
    double grad[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double qi = qa.get(j), lo = qlim[j], hi = qlim[NJ + j];
        const double pij = p.pi[j], den = (p.ps - pij) * (p.ps - pij);
        double sg = 0.0;
        if (qi - lo <= pij) sg = -(((qi - lo) - pij) * ((qi - lo) - pij)) / den;
        if (hi - qi <= pij) sg = (((hi - qi) - pij) * ((hi - qi) - pij)) / den;
        grad[j] = (1.0 / p.kq) * -sg;
    }
    double B[6][6], dval[6], dinv[6];
    jjt_masked<NJ>(jac, 63, B);
    ldl_factor<6>(B, dval, dinv);
    if (p.km > 0.0) {
        double jm[NJ];
        jacobm_factored<NJ>(jac, 63, B, dval, dinv, jm);
#pragma unroll
        for (int j = 0; j < NJ; ++j) grad[j] += (1.0 / p.km) * jm[j];
    }
    double y[6], x[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < NJ; ++k) a += jac[r * NJ + k] * grad[k];
        y[r] = a;
    }
    ldl_backsolve<6>(B, dinv, y, x);
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        double a = grad[k];
#pragma unroll
        for (int r = 0; r < 6; ++r) a -= jac[r * NJ + k] * x[r];
        qn[k] = a;
    }
}

// ---------------------------------------------------------------- IK_QP (robot/IK.py:1222-1520)
// The step of IK_QP is the quadratic programme   min 1/2 x^T Q x + c^T x   s.t.  [J 1_6] x = e,   x = (dq, delta),
// Q = diag(kj 1_n, (ks / sum|e|) 1_6),  c = (-jacobm(q) / km, 0)  (IK.py:1437-1497); the joint-limit velocity dampers add
// inequality rows only when kq > 0.  Without them it is an equality-constrained strictly convex QP and has the closed form
//      dq = g + J^T (J J^T + d^2 1)^-1 (e - J g),      d^2 = kj sum|e| / ks,     g = jacobm(q) / (kj km)   (0 when km = 0)
// (eliminate the multipliers of the 6 equality rows): a damped minimum-norm step whose damping follows the error -- the
// same 6x6 solve as the Gauss-Newton / Newton-Raphson steps above.  The reference hands the QP to quadprog (an absent
// third-party dependency); the minimiser of a strictly convex QP is unique, so any exact solver returns this x.
template <int NJ, class PD>
RTB_HD void ik_qp_gain(const double (&jac)[6 * NJ], const PD &p, double (&g)[NJ])
{
    static_assert(NJ >= 6, "jacobm needs J J^T invertible: a redundant or square arm");
    if (p.km > 0.0) {                  // wave-uniform
        double jm[NJ];
        jacobm<NJ>(jac, 63, jm);
        const double s = 1.0 / (p.lambda * p.km);
#pragma unroll
        for (int j = 0; j < NJ; ++j) g[j] = s * jm[j];
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) g[j] = 0.0;
    }
}

// kq > 0: the joint-limit velocity dampers (IK.py:1453-1481) add, for every joint inside the influence distance pi of a limit,
// ONE inequality row on its own velocity:   s_i dq_i <= beta_i,
//      near the upper limit  s = +1, beta = ((hi - q) - ps) / (pi - ps) / kq;   near the lower  s = -1, beta = ((q - lo) - ps) / (pi - ps) / kq
// (the lower-limit test comes second in the reference and overrides).  With the slack eliminated the programme is a strictly
// convex QP in dq with one-sided bounds, H = kj 1 + (ks / sum|e|) J^T J.  Primal-dual active set: with a set A of joints held
// on their bounds (dq_i = s_i beta_i) the rest has the closed form above on the remaining columns,
//      y = (J_F J_F^T + d^2 1)^-1 (e - J_A v_A - J_F g_F),     u_i = g_i + J_i^T y   for every joint
// (u_i is the free joint's step, and for a held joint kj s_i (u_i - v_i) is its multiplier), and the next set is simply
//      A' = { i constrained :  s_i u_i > beta_i }
// -- a free joint that violates its bound is held, a held joint whose multiplier turns negative is released.  A' = A is the
// KKT point, the unique minimiser (what quadprog returns to rounding).  Every lane of the wave runs the same number of rounds
// (a converged lane recomputes the same numbers); no fixed point after kIkQpRounds rounds reports failure, which the solver
// loop treats like the reference's "QP Unsolvable" (numpy.linalg.LinAlgError: the search is abandoned, IK.py:320-323).
constexpr int kIkQpRounds = 12;
template <int NJ, class PD, class QL, class QA>
RTB_HD bool ik_qp_bounded(const double (&jac)[6 * NJ], const double (&e)[6], double d2, const double (&g)[NJ], const PD &p, QL qlim, QA qa,
                          double (&dq)[NJ])
{
    double sgn[NJ], beta[NJ];          // sgn 0: the joint has no row
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double qi = qa.get(j), lo = qlim[j], hi = qlim[NJ + j];
        const double pij = p.pi[j], scale = 1.0 / ((pij - p.ps) * p.kq);
        sgn[j] = 0.0; beta[j] = 0.0;
        if (hi - qi <= pij) { beta[j] = ((hi - qi) - p.ps) * scale; sgn[j] = 1.0; }
        if (qi - lo <= pij) { beta[j] = ((qi - lo) - p.ps) * scale; sgn[j] = -1.0; }
    }
    unsigned act = 0;
    bool fixed_point = false;
    for (int round = 0; round < kIkQpRounds; ++round) {
        double B[6][6], ep[6], y[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double a = e[r];
#pragma unroll
            for (int k = 0; k < NJ; ++k) a -= jac[r * NJ + k] * (((act >> k) & 1u) ? sgn[k] * beta[k] : g[k]);
            ep[r] = a;
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double b = 0.0;
#pragma unroll
                for (int k = 0; k < NJ; ++k) b += ((act >> k) & 1u) ? 0.0 : jac[r * NJ + k] * jac[c * NJ + k];
                B[r][c] = (r == c) ? b + d2 : b;
            }
        }
        ldl_solve<6>(B, ep, y);
        unsigned next = 0;
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            double u = g[k];
#pragma unroll
            for (int r = 0; r < 6; ++r) u += jac[r * NJ + k] * y[r];
            if (sgn[k] != 0.0 && sgn[k] * u > beta[k]) next |= 1u << k;
            dq[k] = ((act >> k) & 1u) ? sgn[k] * beta[k] : u;
        }
        fixed_point = next == act;
        if (!wave_any(!fixed_point)) break;
        act = next;                    // (a lane at its fixed point keeps its set: next == act)
    }
    return fixed_point;
}

// ---------------------------------------------------------------- searches as pure functions
// The reference runs, per target, up to `slimit` SEARCHES one after another; each search starts from
// a given or random q and takes at most `ilimit` LM steps (ik.cpp:39-72, IK.py:297-367).  With the
// counter-based restart generator a search is a pure function of (target, search index):
//        (ok, iterations it contributes, final q, last E).
// The answer for a target is the LOWEST-indexed successful search; the reported iteration count is
// the sum of the contributions of all searches up to and including it.  Nothing forces the searches
// to run one after another, and the device scheduler below does not (speculative parallel searches).
//
// Search index conventions (what the reference reports as `searches`):
//   flavour 0 (C, ik.cpp)   s = 1 .. s_last, s_last = max(slimit,1); iter starts at 1 for s = 1 and 0
//                           after a restart (ik.cpp:39,67); start of search s: q0 if given and s == 1,
//                           else restart draw s-1-has_q0; failure reports search = s_last+1 and the
//                           NEXT restart vector as q (ik.cpp:66-69).
//   flavour 1 (IK.py)       s = 0 .. s_last, s_last = max(slimit,1)-1; start: q0 / draw 0 for s == 0,
//                           draw s otherwise (IK.py:222-240,351-357); success reports s+1, failure
//                           reports slimit and the last search's final q.
// The target pose (R row-major (9), t (3)) and the joint vector q are NOT part of the register state:
// each is read once at the top of an iteration (and q written once at the bottom), so the kernel
// keeps them in the wave's LDS (IkWaveShared::Td per target slot, ::q per lane; 38 VGPRs that would
// otherwise be live across the whole iteration and push the kernel into scratch; and a search that
// starts on an already loaded target touches no global memory) and every function below reaches them through
// accessors:  td(k) -> double, tdput(k, v);  qa.get(j) -> double, qa.put(j, v).
template <int NJ>
struct IkLane {
    double E;          // E of the last iteration evaluated
    int32_t iter;      // the reference's in-search iteration counter
    int32_t s;         // search index
    int32_t slot;      // target slot of this wave (scheduler), unused by the sequential driver
    int32_t status;    // kIkIdle / kIkRun / kIkParkedOk / kIkParkedLast
    int32_t fin, ok;   // set by ik_iter when the search ended in this iteration (it contributes `iter` iterations)
};
constexpr int kIkIdle = 0, kIkRun = 1, kIkParkedOk = 2, kIkParkedLast = 3;
constexpr int kIkOkPending = 2;    // IkLane::ok of a search that reached the tolerance and awaits its wrap + limit test (ik_settle)

template <class PD>
RTB_HD int ik_s_first(const PD &p) { return p.flavour == 0 ? 1 : 0; }
template <class PD>
RTB_HD int ik_s_last(const PD &p)
{
    const int sl = p.slimit < 1 ? 1 : p.slimit;
    return p.flavour == 0 ? sl : sl - 1;
}

// Watchdog budget of a wave, in loop iterations: the longest a correct run can go without resolving any of its targets.
// The slowest first resolution is a slot whose searches all fail and run one after another on one lane: s_last + 1 searches
// of up to ilimit + 1 iterations, each followed by a wait of up to pass_mask iterations for the next scheduling pass.
template <class PD>
RTB_HD long long ik_patience(const PD &p, int s_last)
{
    return (long long)(p.ilimit + 2 + p.pass_mask) * (s_last + 3) + 64;
}

template <class TDPut>
RTB_HD void ik_load_target(TDPut tdput, const double *Tep16)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tdput(3 * r + c, Tep16[4 * r + c]);
        tdput(9 + r, Tep16[4 * r + 3]);
    }
}

// Start search s of target tgt in this lane (the target pose must already be loaded).
template <int NJ, class PD, class QL, class QA>
RTB_HD void ik_search_begin(IkLane<NJ> &st, QA qa, const PD &p, QL qlim, int64_t tgt, int s, const double *q0row)
{
    st.s = s;
    st.E = 0.0;
    st.fin = 0; st.ok = 0;
    const int s0 = ik_s_first(p);
    st.iter = (p.flavour == 0 && s == s0) ? 1 : 0;          // ik.cpp:39 vs :67
    if (s == s0 && q0row) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) qa.put(j, q0row[j]);
    } else {
        const int draw = p.flavour == 0 ? s - 1 - (q0row ? 1 : 0) : s;
        double qn[NJ];
        ik_restart<NJ>(p.seed, tgt + p.target0, draw, qlim, qn);
#pragma unroll
        for (int j = 0; j < NJ; ++j) qa.put(j, qn[j]);
    }
    st.status = kIkRun;
}

// fmod(x, 2 pi) for the wrap at the end of a search, EXACT like the library's: with c = fl(2 pi) and n = the integer nearest to x / c, the
// remainder r = x - n c is a multiple of 2^-50 of magnitude < 4 and therefore a double, so ONE fused multiply-add returns it without error
// (an n that lands on the wrong side of a tie still gives |r| <= c / 2); fmod's result -- the remainder with the sign of x -- is r or r +- c,
// again exact because fmod's own result is representable.  ~10 instructions against the library routine's ~35 (a bit-serial reduction loop);
// seven of them per scheduling pass.  Beyond |x| = 2^40 (a diverged search) or for a non-finite x the library routine decides.
RTB_HD double ik_fmod_2pi(double x)
{
#if RTB_IK_FMOD_FMA
    if (fabs(x) < 1099511627776.0) {
        const double n = rint(x * 0.15915494309189535);        // fl(1 / (2 pi))
        double r = fma(-n, kIkPi2, x);
        if (x >= 0.0) { if (r < 0.0) r += kIkPi2; }
        else if (r > 0.0) r -= kIkPi2;
        return r;
    }
#endif
    return fmod(x, kIkPi2);
}
RTB_HD double ik_wrap_c(double q) { return ik_fmod_2pi(q + kIkPi) - kIkPi; }        // ik.cpp:51
RTB_HD double ik_wrap_py(double q)                                                      // IK.py:331 (2 * fl(pi) == fl(2 pi): the same modulus)
{
    double r = ik_fmod_2pi(q + kIkPi);
    if (r < 0) r += 2 * kIkPi;
    return r - kIkPi;
}

// ONE LM iteration of the lane's current search.  Every lane of a wave executes this whatever its
// status (idle / parked lanes compute on their stale state and discard the result) so the wave has a
// single instruction stream.  Sets st.fin / st.ok when the search ended.
// STEP bit 0 clear: the three Levenberg-Marquardt steps (method 0..2); set: Gauss-Newton / Newton-Raphson (method 3 / 4).
// STEP bit 1: the Python solvers' null-space motion is added to the step.  Compile-time choices so that no kernel
// carries another's step in its register budget.
constexpr int kIkStepPinv = 1, kIkStepNull = 2;
// UNITW (compile-time, LM steps only): the mask is all ones -- W J, W e and e^T W e need no products.  The same bits (x * 1.0 == x), 54
// multiplies fewer per iteration; as a run-time switch inside one kernel it cost 30 VGPRs (round 3), as a kernel instantiation it costs nothing.
// PLAIN (compile-time): an all-revolute chain without flipped joints -- reg_core<..., PLAIN> (kin_reg.h).
// SIG (compile-time, with PLAIN): the chain's structure signature (kin_reg.h: SegSig) -- every constant segment multiplied in the form of its class.
template <int NJ, int STEP, bool UNITW = false, bool PLAIN = false, SegSig SIG = 0, class PD, class CV, class QL, class TD, class QA>
RTB_HD void ik_iter(IkLane<NJ> &st, const PD &p, const CV &cv, QL qlim, TD td, QA qa)
{
    constexpr bool PINV = (STEP & kIkStepPinv) != 0, NULLSP = (STEP & kIkStepNull) != 0 && NJ >= 6;
    Pose P;
    double jac[6 * NJ], e[6], dq[NJ];
    bool q_finite = true;
    {
        double qv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            qv[j] = qa.get(j);
            q_finite = q_finite && __builtin_isfinite(qv[j]);
        }
        // the chain's last constant C_n (no tool in IK) is segment NJ of the table: {r[9], t[3]} contiguous
        reg_core<NJ, true, PLAIN, SIG>(cv, &cv.seg[NJ].r[0], 0, qv, P, jac);   // ik.cpp:44,56 / IK.py:994,1009
    }
    sched_fence();
    ik_angle_axis(P, td, e);
    double E = ik_half_weighted_square<(UNITW && !PINV)>(e, p.we);   // ik.cpp:46
    double qn[NULLSP ? NJ : 1];
    if constexpr (NULLSP) {
        if (PINV && p.method == 5) ik_qp_gain<NJ>(jac, p, qn);  // IK_QP's manipulability term (wave-uniform branch)
        else ik_qnull<NJ>(jac, p, qlim, qa, qn);                // IK.py:753,1011,1210 (before the step: J dies in it)
    }
    bool qp_ok = true, qp_done = false;
    if (PINV) {                     // 3 Gauss-Newton, 4 Newton-Raphson (`lambda` carries pinv_damping), 5 IK_QP (`lambda` carries kj)
        int rows = 63;
        double d2 = 0.0;
        if (p.method == 3) {
            rows = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) rows |= (p.we[k] != 0.0) ? (1 << k) : 0;
        } else if (p.method == 4) {
            d2 = p.lambda * p.lambda;
        } else if (p.method == 5) {
            double se = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) se += fabs(e[k]);
            d2 = p.lambda * se / p.ks;                          // IK.py:1442-1446
            if constexpr (NULLSP) {
                if (p.kq > 0.0) {                               // inequality rows: active-set rounds (wave-uniform branch)
                    qp_ok = ik_qp_bounded<NJ>(jac, e, d2, qn, p, qlim, qa, dq);
                    qp_done = true;
                } else {                                        // e - J g
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        double a = e[r];
#pragma unroll
                        for (int k = 0; k < NJ; ++k) a -= jac[r * NJ + k] * qn[k];
                        e[r] = a;
                    }
                }
            }
        }
        if (!qp_done) ik_pinv_step<NJ>(jac, e, rows, d2, &p.we[0], p.method == 3, dq);
    } else {
        const double wn = (p.method == 1) ? p.lambda : (p.method == 2) ? E + p.lambda : p.lambda * E;   // ik.cpp:169,183,205
        if constexpr (UNITW && RTB_IK_DUAL && NJ >= 7) {
            if (p.lambda > 0.0 || p.method == 2) ik_lm_step_dual<NJ>(jac, e, wn, dq);      // wave-uniform; (sugihara: wn = E + lambda > 0 off the solution)
            else ik_lm_step<NJ, true>(jac, e, &p.we[0], wn, dq);
        } else if constexpr (UNITW) ik_lm_step<NJ, true>(jac, e, &p.we[0], wn, dq);
        else if (RTB_IK_UNITW && p.unit_we) ik_lm_step<NJ, true>(jac, e, &p.we[0], wn, dq);          // wave-uniform
        else ik_lm_step<NJ, false>(jac, e, &p.we[0], wn, dq);
    }
    if constexpr (NULLSP) {
        if (!qp_done) {                // (the bounded QP returns the whole step)
#pragma unroll
            for (int j = 0; j < NJ; ++j) dq[j] += qn[j];
        }
    }
    if (st.status != kIkRun || st.fin) return;    // a search that has ended waits, untouched, for the next pass
    const bool arrived = E < p.tol;
    if (p.flavour == 0) {
        st.E = E;
        if (arrived) {                                          // ik.cpp:48-54: the wrap and the limit test happen in ik_settle
            st.fin = 1; st.ok = kIkOkPending;
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) qa.put(j, qa.get(j) + dq[j]);      // ik.cpp:57
            st.iter++;
            if (st.iter > p.ilimit) { st.fin = 1; st.ok = 0; }
        }
    } else {
        st.iter++;                                              // IK.py:315
        if ((PINV || NULLSP) && (!q_finite || !qp_ok)) {
            // numpy.linalg.pinv raises LinAlgError on a non-finite J (LAPACK gesdd refuses NaN: info != 0), and J -- a polynomial in
            // the sines and cosines -- is non-finite exactly when q is.  IK.py:320-323: the search is abandoned, the iteration
            // counted, E and q left as the last completed step had them.  (numpy.linalg.inv of the plain LM step does not raise.)
            st.fin = 1; st.ok = 0;
            return;
        }
        st.E = E;
#pragma unroll
        for (int j = 0; j < NJ; ++j) dq[j] += qa.get(j);        // the step is taken before E is tested (IK.py:319-327)
        if (arrived) {
            st.fin = 1; st.ok = kIkOkPending;                    // IK.py:336-351: wrap and limit test in ik_settle
        } else if (st.iter >= p.ilimit) {
            st.fin = 1; st.ok = 0;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) qa.put(j, dq[j]);
    }
}

// The end of a search that reached the tolerance: wrap q into [-pi, pi) the way the solver does (C: fmod, ik.cpp:51; Python: %,
// IK.py:331) and test the joint limits (ik.cpp:227-239, IK.py:336-351).  Kept OUT of ik_iter: with 64 lanes some lane arrives in
// nearly every iteration, and the wave would execute these ~260 instructions (seven fmod expansions) every time; here they run
// once per scheduling pass (ik_report) -- at most every (pass_mask + 1)-th iteration.  The search is over either way (fin is set
// by ik_iter); only `ok` and the wrapped q are pending.
template <int NJ, class PD, class QL, class QA>
RTB_HD void ik_settle(IkLane<NJ> &st, const PD &p, QL qlim, QA qa)
{
    if (!st.fin || st.ok != kIkOkPending) return;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double w = p.flavour == 0 ? ik_wrap_c(qa.get(j)) : ik_wrap_py(qa.get(j));
        qa.put(j, w);
        if (w < qlim[j] || w > qlim[NJ + j]) ok = false;
    }
    st.ok = (ok || !p.reject_jl) ? 1 : 0;
}

// which compiled step variant serves these parameters (host launcher, emu and the sequential driver agree on it)
template <class PD>
RTB_HD int ik_step_variant(const PD &p, int n)
{
    const bool extra = p.method == 5 ? (p.km > 0.0 || p.kq > 0.0) : p.kq > 0.0;     // IK_QP: manipulability term / inequality rows; the others: null-space motion
    return (p.method >= 3 ? kIkStepPinv : 0) | ((extra && n >= 6 && n <= 12) ? kIkStepNull : 0);
}
template <int NJ, class PD, class CV, class QL, class TD, class QA>
RTB_HD void ik_iter_any(IkLane<NJ> &st, const PD &p, const CV &cv, QL qlim, TD td, QA qa)
{
    switch (ik_step_variant(p, NJ)) {
    case 0: ik_iter<NJ, 0>(st, p, cv, qlim, td, qa); break;
    case 1: ik_iter<NJ, 1>(st, p, cv, qlim, td, qa); break;
    case 2: ik_iter<NJ, 2>(st, p, cv, qlim, td, qa); break;
    default: ik_iter<NJ, 3>(st, p, cv, qlim, td, qa); break;
    }
}

// What the reference reports for a target whose winning / last search is held by this lane.
template <int NJ, class PD, class QL, class QA>
RTB_HD void ik_emit(const IkLane<NJ> &st, QA qa, const PD &p, QL qlim, int64_t tgt, int64_t row, bool has_q0, bool success, int it_total,
                    double *__restrict__ q_out, int32_t *__restrict__ success_out,
                    int32_t *__restrict__ iters, int32_t *__restrict__ searches, double *__restrict__ residual)
{
    double qf[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) qf[j] = qa.get(j);
    int se;
    if (success) {
        se = p.flavour == 0 ? st.s : st.s + 1;
    } else if (p.flavour == 0) {
        se = ik_s_last(p) + 1;
        ik_restart<NJ>(p.seed, tgt + p.target0, ik_s_last(p) - (has_q0 ? 1 : 0), qlim, qf);   // ik.cpp:66-69: q is the next restart
    } else {
        se = p.slimit;                                                             // IK.py:359-366
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) q_out[row * NJ + j] = qf[j];
    success_out[row] = success ? 1 : 0;
    iters[row] = it_total;
    searches[row] = se;
    if (success) residual[row] = st.E;           // on failure ik_report has already left the last search's E there
}

// ---------------------------------------------------------------- per-wave scheduler (speculative searches)
// A wave (single-wave workgroup) owns up to 64 targets at a time, one per SLOT, and its 64 lanes are
// workers that each run one (slot, search) at a time.  While fresh targets remain every idle lane
// takes a new target (search s_first).  When the global supply is exhausted, idle lanes are handed
// the NEXT search indices of the wave's unresolved targets, so the long tail of hard targets
// (dozens of restarts, thousands of sequential iterations in the reference) is walked up to 64
// searches at a time.  Results are accounted strictly in search order, so the outputs are identical
// to running the searches one after another.
// All tables live in the wave's LDS; the phases below are per-lane functions called between
// wave-level barriers by the kernel (lane = threadIdx.x) and, lane by lane, by tests/emu.
// kIkRing = 64: a wave left with ONE unresolved target can put all its 64 lanes on it (CPU replay of config 3, 2048
// waves: longest wave 219 -> 194 iterations, lane utilisation 0.72 -> 0.78 against a ring of 32).  The struct is sized so
// that 8 waves per CU still fit the 160 KB of LDS for chains of up to 8 joints (20288 B with QR = 8): 32-bit target
// indices, 16-bit b / res, 8-bit list, and only as many q rows as the kernel's joint count class needs.
#ifndef RTB_IK_RING
#define RTB_IK_RING 64
#endif
constexpr int kIkRing = RTB_IK_RING;        // outstanding (unaccounted) searches per slot (a power of two)
// A/B knobs of the three-waves-per-SIMD study (profiles/r06_ik_three_waves.txt; scripts/build_ik_variant.sh): twelve waves on a CU leave each
// 13 653 B of LDS -- RTB_IK_REC8 packs a search record into 8 bits (iterations <= 62: the launcher refuses a larger ilimit in such a build),
// RTB_IK_QROWS_EXACT sizes the q rows to the kernel's joint count instead of its class (7 rows for the Panda, not 8); with a ring of 32
// that is 13 624 B for a 7-joint arm.  The product keeps 16-bit records, a ring of 64, 8 rows: 20 280 B, eight waves.
#ifndef RTB_IK_REC8
#define RTB_IK_REC8 0
#endif
#ifndef RTB_IK_QROWS_EXACT
#define RTB_IK_QROWS_EXACT 0
#endif
#if RTB_IK_REC8
typedef uint8_t IkRec;
#else
typedef uint16_t IkRec;
#endif
template <int QR>
struct alignas(16) IkWaveSharedT {
    uint32_t vix[64];                       // work-item index = output row (the target itself without a work list; < 2^32)
    uint32_t tgt[64];                       // target index of the slot's item (E of an item's last search lives in residual[row])
    int16_t b[64];                          // lowest search index not yet accounted (slimit <= 32000)
    int16_t slast[64];                      // last search index of the slot's work item
    int32_t next[64];                       // next search index to hand out (LDS atomic max)
    int32_t best[64];                       // lowest successful search index seen so far (LDS atomic min)
    int32_t it[64];                         // iterations accounted so far
    int16_t res[64];                        // 0 unresolved, 1 won (winner = b), 2 failed
    uint8_t list[64];                       // scratch: compacted slot list
    uint8_t chunk[64];                      // flat schedule: chunk index of the slot's item (0 otherwise)
    IkRec rec[64][kIkRing];                 // per outstanding search: 1 finished | 2 ok | iterations << 2
    double Td[12][64];                      // per SLOT: the target pose (read by every lane working on the slot)
    double q[QR][64];                       // per LANE: the joint vector of that search
};
// q rows by joint-count class: 8 (the register-resident kernels), 16 (the other built-in sizes), RTBHIP_MAX_JOINTS (sizes instantiated at run time)
template <int NJ> using IkWaveSharedFor = IkWaveSharedT<(RTB_IK_QROWS_EXACT ? NJ : (NJ <= kRegMaxJoints ? kRegMaxJoints : (NJ <= kIkMaxJoints ? kIkMaxJoints : RTBHIP_MAX_JOINTS)))>;
static_assert(sizeof(IkWaveSharedT<kRegMaxJoints>) * 8 <= 160 * 1024, "8 IK waves per CU must fit the LDS");

// A work item: searches s0 .. s1 (inclusive, in the flavour's own numbering) of target `tgt`.  Without a work list item v is
// target v with the whole range.  A search is a pure function of (target, search index), so any partition of a target's
// searches into items can be run anywhere, in any order, and merged afterwards IN SEARCH ORDER (ik_merge_*): launch_ik uses
// that to spread the hard targets of a batch that is small against the chip over all the waves (phased schedule, below).
struct IkWork { int32_t tgt; int16_t s0, s1; };
RTB_HD unsigned long long ik_pack(IkWork w) { return (unsigned long long)(uint32_t)w.tgt | ((unsigned long long)(uint16_t)w.s0 << 32) | ((unsigned long long)(uint16_t)w.s1 << 48); }
RTB_HD IkWork ik_unpack(unsigned long long x) { IkWork w; w.tgt = (int32_t)(uint32_t)x; w.s0 = (int16_t)(x >> 32); w.s1 = (int16_t)(x >> 48); return w; }
constexpr int kIkMaxSlimit = 32000;
constexpr int kIkMaxIlimit = RTB_IK_REC8 ? 62 : 16000;         // (ilimit + 1) << 2 must fit the record

template <class SH>
struct IkLdsQT {   // accessor of one lane's q column in the wave's LDS
    SH *sh;
    int lane;
    RTB_HD double get(int j) const { return sh->q[j][lane]; }
    RTB_HD void put(int j, double v) const { sh->q[j][lane] = v; }
};
template <class SH> RTB_HD IkLdsQT<SH> ik_lds_q(SH &sh, int lane) { return IkLdsQT<SH>{&sh, lane}; }
constexpr int kIkNoBest = 0x7fffffff;

RTB_HD void ik_lds_min(int32_t *p, int32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
RTB_HD void ik_lds_max(int32_t *p, int32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}
RTB_HD int ik_rank(unsigned long long mask, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(mask & ((1ull << lane) - 1ull));
#else
    return __builtin_popcountll(mask & ((1ull << lane) - 1ull));
#endif
}

// phase A: a lane whose search just ended posts the result and parks or goes idle
template <int NJ, class SH, class PD, class QL, class QA>
RTB_HD void ik_report(IkLane<NJ> &st, SH &sh, double *__restrict__ residual, const PD &p, QL qlim, QA qa)
{
    if (st.status != kIkRun || !st.fin) return;
    ik_settle<NJ>(st, p, qlim, qa);
    const int s_last = sh.slast[st.slot];
    sh.rec[st.slot][st.s & (kIkRing - 1)] = (IkRec)(1 | (st.ok ? 2 : 0) | (st.iter << 2));
    if (st.ok) ik_lds_min(&sh.best[st.slot], st.s);
    if (st.s == s_last) residual[sh.vix[st.slot]] = st.E;     // the failure output's E; a success found later overwrites it
    st.status = st.ok ? kIkParkedOk : (st.s == s_last ? kIkParkedLast : kIkIdle);
    st.fin = 0;
}

// phase B: lane i accounts slot i in search order
template <class SH>
RTB_HD void ik_account(int i, SH &sh)
{
    const int s_last = sh.slast[i];
    int b = sh.b[i], it = sh.it[i], res = 0;
    for (;;) {
        const int r = sh.rec[i][b & (kIkRing - 1)];
        if (!(r & 1)) break;
        sh.rec[i][b & (kIkRing - 1)] = 0;
        it += r >> 2;
        if (r & 2) { res = 1; break; }
        if (b == s_last) { res = 2; break; }
        ++b;
    }
    sh.b[i] = (int16_t)b; sh.it[i] = it; sh.res[i] = (int16_t)res;
}

// phase C: parked lanes of a resolved slot emit / release; searches beyond a known success are cancelled
template <int NJ, class SH, class PD, class QL>
RTB_HD void ik_finalize(IkLane<NJ> &st, SH &sh, int lane, const PD &p, QL qlim, double *__restrict__ q_out,
                        int32_t *__restrict__ success, int32_t *__restrict__ iters, int32_t *__restrict__ searches,
                        double *__restrict__ residual)
{
    if (st.status == kIkIdle) return;
    const int res = sh.res[st.slot];
    if (res == 1) {
        if (st.status == kIkParkedOk && st.s == sh.b[st.slot])
            ik_emit<NJ>(st, ik_lds_q(sh, lane), p, qlim, sh.tgt[st.slot], sh.vix[st.slot], p.has_q0 != 0, true, sh.it[st.slot], q_out, success, iters, searches, residual);
        st.status = kIkIdle;
    } else if (res == 2) {
        if (st.status == kIkParkedLast)
            ik_emit<NJ>(st, ik_lds_q(sh, lane), p, qlim, sh.tgt[st.slot], sh.vix[st.slot], p.has_q0 != 0, false, sh.it[st.slot], q_out, success, iters, searches, residual);
        st.status = kIkIdle;
    } else if (res == 3) {
        st.status = kIkIdle;      // flat schedule: an EARLIER chunk of this target has succeeded on some other wave -- nothing of this item is needed
    } else if (st.s > sh.best[st.slot]) {
        st.status = kIkIdle;      // a lower-indexed search already succeeded: this one can never be reported
    }
}

// a slot's ring of search records back to "nothing finished": 16-byte stores (the row is 2 * kIkRing bytes, 16-byte aligned inside the wave's
// LDS block) instead of kIkRing 2-byte ones -- the loop runs for the whole wave whenever one lane starts a work item
RTB_HD void ik_clear_ring(IkRec (&row)[kIkRing])
{
    static_assert((kIkRing * sizeof(IkRec)) % 16 == 0, "ring row must be a whole number of 16-byte pieces");
    __builtin_memset(__builtin_assume_aligned(&row[0], 16), 0, sizeof(row));
}

// phase D1 helper: initialise slot `slot` for target tgt and start its first search in this lane
template <int NJ, class SH, class PD, class QL>
RTB_HD void ik_start_target(IkLane<NJ> &st, SH &sh, int lane, const PD &p, QL qlim, int slot, int64_t v, IkWork w,
                            const double *__restrict__ Tep, const double *__restrict__ q0)
{
    const int64_t tgt = w.tgt;
    const int s0 = w.s0;
    sh.vix[slot] = (uint32_t)v; sh.tgt[slot] = (uint32_t)tgt; sh.b[slot] = (int16_t)s0; sh.next[slot] = s0 + 1; sh.best[slot] = kIkNoBest; sh.it[slot] = 0;
    sh.slast[slot] = w.s1;
    sh.res[slot] = 0;
    ik_clear_ring(sh.rec[slot]);
    st.slot = slot;
    ik_load_target([&](int k, double v) { sh.Td[k][slot] = v; }, Tep + 16 * tgt);   // once per target, not per search
    ik_search_begin<NJ>(st, ik_lds_q(sh, lane), p, qlim, tgt, s0, p.has_q0 ? q0 + (int64_t)NJ * tgt : nullptr);
}

// phase D0: a slot none of whose searches is outstanding (its last running search just failed) must be
// continued before anything else is started -- this is the reference's "next restart".
template <class SH>
RTB_HD bool ik_starved(int i, const SH &sh) { return sh.res[i] == 0 && sh.next[i] == sh.b[i]; }

// phase D2, step 1: idle lane number `r` (of the idle lanes) picks a slot round-robin over the nb busy
// slots (sh.list) and the q-th next search index of it; returns whether that index may start now.
template <class SH>
RTB_HD bool ik_pick(const SH &sh, int r, int nb, int ni, int policy, int s_first, int &slot, int &s)
{
    if (policy == 0) {
        slot = sh.list[r % nb];
        s = sh.next[slot] + r / nb;
    } else {
        // failure-weighted: a slot that has already failed f searches gets f + 1 shares of the idle lanes (its explored range
        // doubles per round): the targets that will exhaust every search -- 1 % of a batch, 40 x the mean cost -- are walked wide
        // early instead of keeping their wave busy after all the others have gone
        int W = 0;
        for (int k = 0; k < nb; ++k) { const int f = sh.b[sh.list[k]] - s_first; W += 1 + (f < 0 ? 0 : (f > 63 ? 63 : f)); }
        const int pos = (int)(((long long)r * W) / ni);
        int cum = 0, k = 0, w = 1;
        for (; k < nb; ++k) {
            const int f = sh.b[sh.list[k]] - s_first;
            w = 1 + (f < 0 ? 0 : (f > 63 ? 63 : f));
            if (pos < cum + w) break;
            cum += w;
        }
        if (k >= nb) { k = nb - 1; cum -= w; }
        slot = sh.list[k];
        const int r0 = (int)(((long long)cum * ni + W - 1) / W);      // first idle-lane rank that maps to this slot
        s = sh.next[slot] + (r - r0);
    }
    return s <= sh.slast[slot] && s - sh.b[slot] < kIkRing && s < sh.best[slot];
}
// step 2 (after every lane has picked): claim the index and start the search
template <int NJ, class SH, class PD, class QL>
RTB_HD void ik_start_spec(IkLane<NJ> &st, SH &sh, int lane, const PD &p, QL qlim, int slot, int s,
                          const double *__restrict__ Tep, const double *__restrict__ q0)
{
    ik_lds_max(&sh.next[slot], s + 1);
    const int64_t tgt = sh.tgt[slot];
    st.slot = slot;
    ik_search_begin<NJ>(st, ik_lds_q(sh, lane), p, qlim, tgt, s, p.has_q0 ? q0 + (int64_t)NJ * tgt : nullptr);
}

// ---------------------------------------------------------------- cross-wave sharing of search ranges
// With the whole batch resident at once (BASELINE config 3: 1e5 targets on 131072 lanes) a wave is stuck with the targets it
// drew; the ~1 % that exhaust all `slimit` searches cost 40x the mean, so the waves that drew two or three of them run twice as
// long as the average one and the kernel waits for them (CPU replay of config 3: longest wave 177-217 iterations, mean 104).
// A search is a pure function of (target, search index), so the UNSTARTED part of a slot's search range can be cut off
// and handed to another wave as a new work item, and a target's rows are chained in search order (`link`) for a final merge
// (iterations add up along the chain; the first success, or the chain's last row, supplies the answer) -- exactly the
// sequential loops' result, whoever ran what.
//
// Hand-over protocol.  (The first design -- one shared list every idle wave polled and raced for with compare-and-swap -- was
// correct but 150x SLOWER on the MI355X: each appended item woke ~1500 waiting waves into a retry storm on one address.  The
// second -- tickets, below, but on ONE control word -- worked at last, yet gained nothing: same-address read-modify-writes
// cross the fabric one at a time, ~0.2 us each, and the ~5600 of a config-3 launch (a ticket per wave, two per hand-over)
// add up to more than the kernel's own 1.4 ms.  Hence kIkQueues independent queues, each control word in its own 256-byte line.)
//   * queue g has one 64-bit word  tickets << 32 | appended.  A wave that has run out of work takes a TICKET t in its queue
//     (wave number mod kIkQueues; one fetch-add of 1 << 32) and then polls only ITS OWN word of the item table; a donor
//     reserves indices k .. k+g-1 in a queue with one fetch-add of g and stores the items there: item k goes to the holder of
//     ticket k, nobody races for it, no two waves poll the same address.
//   * tickets - appended of a queue = its waves waiting right now (an exact snapshot: both halves come from one atomic).
//     A donor looks at all queue words with ONE vector load per scheduling pass (lane g reads word g), and only while it holds
//     a range worth cutting; it serves the first queue with waiters, starting from a rotating offset.
//   * termination: after taking its ticket a wave reads all queue words twice.  If the sum of tickets - appended equals the
//     grid size and nothing moved between the two reads, every wave of the grid holds an unserved ticket, nothing can be
//     appended any more, and this wave stores the EXIT mark into every outstanding ticket's word.  (While any wave is busy it
//     holds no ticket, so the sum is short; the wave that takes the LAST ticket reads a quiescent state and sees it.  A queue
//     never fills up: donors stop at `qlimit`, the table has room for one more item per concurrent donor beyond it.)
constexpr int kIkQueues = 16;
constexpr int kIkQueueStride = 32;      // 64-bit words between two queues' control words: one 256-byte line each
struct IkShareCtl {
    unsigned long long *tc;        // word g * kIkQueueStride: tickets handed to waiting waves (high half) | items appended (low half) of queue g
    unsigned long long *wdyn;      // word g * qcap + k: item k of queue g, packed IkWork; ~0 = not written yet
    int32_t *link;                 // row -> next row of the same target, -1 at the end of a chain
    uint32_t qlimit, qcap, waves;  // items a queue may take; words per queue (qlimit + a ticket / a racing donor per wave); grid size
    uint32_t after;                // a slot's range is cut only once this many of the target's searches have FAILED (see ik_donatable)
};
constexpr unsigned long long kIkNoItem = ~0ull, kIkExitItem = ~0ull - 1ull;
constexpr int kIkDonateMin = 4;    // a slot gives away its unstarted searches (all but the next one) when at least this many are left
constexpr int kIkGiveMax = 4;      // ranges a wave hands over per scheduling pass (bounds both the pass and what donors racing past qlimit can add)

RTB_HD unsigned long long ik_aload(const unsigned long long *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}
RTB_HD int32_t ik_aload(const int32_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}
RTB_HD void ik_astore(int32_t *p, int32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}
RTB_HD void ik_astore(unsigned long long *p, unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}
RTB_HD unsigned long long ik_aadd(unsigned long long *p, unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    const unsigned long long o = *p; *p = o + v; return o;
#endif
}
RTB_HD unsigned long long *ik_queue_word(const IkShareCtl &c, int g) { return c.tc + (size_t)g * kIkQueueStride; }
RTB_HD unsigned ik_word_tickets(unsigned long long x) { return (unsigned)(x >> 32); }
RTB_HD unsigned ik_word_count(unsigned long long x) { return (unsigned)(x & 0xffffffffull); }
// waves of a queue waiting for an item (tickets beyond the appended items)
RTB_HD unsigned ik_word_waiting(unsigned long long x) { return ik_word_tickets(x) > ik_word_count(x) ? ik_word_tickets(x) - ik_word_count(x) : 0u; }
// row of item k of queue g in the result temporaries / the link table
RTB_HD int64_t ik_item_row(const IkShareCtl &c, int64_t N, int g, unsigned k) { return N + (int64_t)g * c.qcap + k; }
// A wave that has run dry takes a ticket in queue g (one lane calls this).
RTB_HD unsigned ik_ticket(const IkShareCtl &c, int g) { return ik_word_tickets(ik_aadd(ik_queue_word(c, g), 1ull << 32)); }
// EXIT into the words of the outstanding tickets of queue g, whose control word read x (lane-strided)
RTB_HD void ik_release_queue(const IkShareCtl &c, int g, unsigned long long x, int lane)
{
    for (unsigned k = ik_word_count(x) + (unsigned)lane; k < ik_word_tickets(x); k += (unsigned)kWave)
        ik_astore(c.wdyn + (size_t)g * c.qcap + k, kIkExitItem);
}
// Cut the unstarted searches of slot `slot` (but one) off into item k of queue g (indices reserved by the caller).  One lane
// calls this; the slot's tables are this wave's own.
template <class SH>
RTB_HD void ik_donate(const IkShareCtl &c, int64_t N, SH &sh, int slot, int g, unsigned k)
{
    // everything but the next search: the owner's lanes are all busy at this moment (that is why the searches are unstarted),
    // an idle wave can start them at once.  The slot keeps one unstarted search so that its (new) last search is still ahead --
    // the lane that will run it reports it as the range's last one.
    const int mid = sh.next[slot] + 1, last = sh.slast[slot];
    IkWork w; w.tgt = (int32_t)sh.tgt[slot]; w.s0 = (int16_t)mid; w.s1 = (int16_t)last;
    const int64_t row = ik_item_row(c, N, g, k), mine = sh.vix[slot];
    // the new item continues this slot's range: insert it right after the slot's row.  link[mine] may have been written by the
    // wave (possibly on another XCD, whose L2 this one does not snoop) that donated `mine` itself: agent-scope accesses, not
    // plain ones that could be served from a stale line
    ik_astore(c.link + row, ik_aload(c.link + mine));
    ik_astore(c.link + mine, (int32_t)row);
    sh.slast[slot] = (int16_t)(mid - 1);
    ik_astore(c.wdyn + (size_t)g * c.qcap + k, ik_pack(w));      // hands the item to the holder of ticket k (now or later)
}
// A slot whose range is worth cutting: enough unstarted searches, and a target that has already failed `after` searches (all
// searches below the ring base b have been accounted as failures).  Without the second condition a donor hands out pure
// speculation -- in the ik_benchmark-notebook setting, where nearly every first search succeeds, every range given away at the
// end of the kernel cost its receiver a whole search for nothing (measured: 0.49 -> 0.74 ms per 1e5 targets).
template <class SH>
RTB_HD bool ik_donatable(const SH &sh, int slot, int s_first, int after)
{
    return sh.res[slot] == 0 && (int)sh.slast[slot] - (int)sh.next[slot] + 1 >= kIkDonateMin && (int)sh.b[slot] - s_first >= after;
}
// Final merge of one target's chain of rows (in search order) into the caller's arrays.
RTB_HD void ik_merge_chain(int n, int64_t tgt, const int32_t *link, const double *vq, const int32_t *vok, const int32_t *vit, const int32_t *vse,
                           const double *vE, double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual)
{
    int64_t r = tgt;
    int it = 0;
    for (;;) {
        it += vit[r];
        if (vok[r] || link[r] < 0) break;
        r = link[r];
    }
    for (int j = 0; j < n; ++j) q_out[tgt * n + j] = vq[r * n + j];
    success[tgt] = vok[r]; iters[tgt] = it; searches[tgt] = vse[r]; residual[tgt] = vE[r];
}

// ---------------------------------------------------------------- flat schedule: every search range cut into chunks, ONE launch
// BASELINE config 3 (1e5 targets on 131072 lanes) has the whole batch resident at once, so a wave is stuck with the ~49 targets it
// drew; the ~1 % that exhaust all `slimit` searches cost 40x the mean, and the two or three waves that drew four of them decide when
// the kernel ends (lane-occupancy histogram, profiles/r03_*_ik_occupancy.txt: longest wave ~4x the mean).  Earlier attempts moved work
// between waves AFTER the fact (cross-wave sharing: a hand-over protocol whose atomics cost what it saved) or in separate launches
// (phased: three tails instead of one).  Here nothing is handed over: the search range of EVERY target is cut up front into chunks --
// chunk 0 = its first flat_l0 searches, then flat_len each -- and the work items (target, chunk) are numbered CHUNK-MAJOR, all chunk-0
// items first.  Waves draw item numbers from the one device-wide counter they already use for fresh targets, so the later chunks of
// the hard targets are picked up by whichever waves have idle lanes -- all over the chip, as soon as those lanes exist.
//   * `done[t]` holds the lowest chunk of target t that has succeeded.  An item (t, c) with done[t] < c is skipped when its number is
//     drawn (one load per drawn number, 64 numbers per draw) and, if it is already running, dropped at its wave's next scheduling pass.
//   * later-chunk items are speculation on top of the wave's own: a wave serves its own slots' next searches first and draws
//     later-chunk numbers only for the lanes that are still idle after that.
//   * every item writes its result row (item number); a merge kernel walks each target's rows in chunk order -- iterations add up, the
//     first success or the last chunk supplies the answer -- so the reported (q, success, iterations, searches, residual) are exactly
//     the sequential loops'.  (A row that is needed is always complete: an item is skipped or dropped only when an EARLIER chunk has
//     succeeded, and then the merge stops before it.)
constexpr int32_t kIkFlatNone = 0x7f7f7f7f;      // what hipMemsetAsync(0x7f) leaves
struct IkFlatPlan { int chunks, l0, len; };
template <class PD>
RTB_HD IkFlatPlan ik_flat_plan(const PD &p, int l0, int len)
{
    IkFlatPlan f;
    const int total = ik_s_last(p) - ik_s_first(p) + 1;
    f.l0 = l0 < 1 ? 1 : (l0 > total ? total : l0);
    f.len = len < 1 ? 1 : len;
    f.chunks = 1 + (total - f.l0 + f.len - 1) / f.len;
    return f;
}
// item v of the flat numbering -> (work item, chunk)
template <class PD>
RTB_HD IkWork ik_flat_item(const PD &p, uint32_t v, int *chunk)
{
    const uint32_t c = v / p.flat_n, t = v - c * p.flat_n;
    const int s_first = ik_s_first(p), s_last = ik_s_last(p);
    const int a = c == 0 ? s_first : s_first + p.flat_l0 + (int)(c - 1) * p.flat_len;
    const int b = c == 0 ? s_first + p.flat_l0 - 1 : a + p.flat_len - 1;
    IkWork w;
    w.tgt = (int32_t)t; w.s0 = (int16_t)a; w.s1 = (int16_t)(b > s_last ? s_last : b);
    *chunk = (int)c;
    return w;
}
// is item v still worth starting?  (chunk 0 always is)
template <class PD>
RTB_HD bool ik_flat_live(const PD &p, uint32_t v)
{
    if (v < p.flat_n) return true;
    const uint32_t c = v / p.flat_n, t = v - c * p.flat_n;
    return ik_aload(p.flat_done + t) > (int32_t)c;
}
RTB_HD void ik_flat_publish(int32_t *done, uint32_t t, int c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_min(done + t, (int32_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    if (c < done[t]) done[t] = c;
#endif
}
// Merge of one target's rows, chunk by chunk (rows are item numbers: chunk c of target t is row c * N + t).
RTB_HD void ik_merge_flat(int n, int chunks, int64_t N, int64_t t, const double *vq, const int32_t *vok, const int32_t *vit, const int32_t *vse,
                          const double *vE, double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual)
{
    int it = 0;
    for (int c = 0; c < chunks; ++c) {
        const int64_t r = (int64_t)c * N + t;
        it += vit[r];
        if (vok[r] || c == chunks - 1) {
            for (int j = 0; j < n; ++j) q_out[t * n + j] = vq[r * n + j];
            success[t] = vok[r]; iters[t] = it; searches[t] = vse[r]; residual[t] = vE[r];
            return;
        }
    }
}

// ---------------------------------------------------------------- phased schedule for batches small against the chip
// With the whole batch resident at once (BASELINE config 3: 1e5 targets on 131072 lanes) a wave is stuck with the targets it
// drew: the ~1 % that exhaust all `slimit` searches cost 40x the mean and decide when their wave -- and the kernel -- ends
// (CPU replay: longest wave 194 iterations, mean 97, ideal 57).  So the search range is cut into phases, each its own launch
// over ALL the waves:   phase A  searches [first, first+6) of every target (plain mode);
//                       phase B  the next 16 searches of the targets still unresolved (one work item each);
//                       phase C  the rest of the range of the few still unresolved, split into <= 8 work items per target.
// Items of a target are merged in search order, so the reported (q, success, iterations, searches, residual) are exactly
// those of the sequential loops.  Unresolved targets are compacted with an atomic counter: the ORDER of the work list is
// not deterministic, the results are.
struct IkPhases { int a_last, b_last, c_chunks, c_len, s_last; };   // last search of phase A / B; chunking of phase C
template <class PD>
RTB_HD IkPhases ik_phases(const PD &p)
{
    IkPhases ph;
    const int s0 = ik_s_first(p);
    ph.s_last = ik_s_last(p);
    const int total = ph.s_last - s0 + 1;
    const int na = total < 6 ? total : 6;
    const int nb = total - na < 16 ? total - na : 16;
    ph.a_last = s0 + na - 1;
    ph.b_last = ph.a_last + nb;
    const int rest = ph.s_last - ph.b_last;
    ph.c_chunks = rest <= 0 ? 0 : (rest + 12) / 13 > 8 ? 8 : (rest + 12) / 13;
    ph.c_len = ph.c_chunks ? (rest + ph.c_chunks - 1) / ph.c_chunks : 0;
    return ph;
}
// after phase A (plain mode, results already in the final arrays): an unresolved target becomes item `slot` of phase B
RTB_HD IkWork ik_item_b(const IkPhases &ph, int64_t tgt) { IkWork w; w.tgt = (int32_t)tgt; w.s0 = (int16_t)(ph.a_last + 1); w.s1 = (int16_t)ph.b_last; return w; }
// chunk c of phase C
RTB_HD IkWork ik_item_c(const IkPhases &ph, int64_t tgt, int c)
{
    IkWork w;
    w.tgt = (int32_t)tgt;
    const int a = ph.b_last + 1 + c * ph.c_len, b = a + ph.c_len - 1;
    w.s0 = (int16_t)a; w.s1 = (int16_t)(b > ph.s_last ? ph.s_last : b);
    return w;
}
// Fold the result row `v` of a later item into the target's running result: iterations add up; a success, or the item that
// ends with the range's last search, also supplies (q, success, searches, residual).  Returns whether the target is resolved.
template <int NJ_RT>
RTB_HD bool ik_merge_item(int n, bool range_ends_here, int64_t tgt, int64_t v, const double *vq, const int32_t *vok, const int32_t *vit,
                          const int32_t *vse, const double *vE, double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual)
{
    iters[tgt] += vit[v];
    if (vok[v] || range_ends_here) {
        for (int j = 0; j < n; ++j) q_out[tgt * n + j] = vq[v * n + j];
        success[tgt] = vok[v]; searches[tgt] = vse[v]; residual[tgt] = vE[v];
        return true;
    }
    return false;
}

// Sequential driver (one target, searches in order): the specification the scheduler must reproduce.
// Used by tests/emu; the kernel never calls it.
template <int NJ, class CV, class QL>
RTB_HD void ik_solve_sequential(const IkDev &p, const CV &cv, QL qlim, int64_t tgt, const double *Tep,
                                const double *q0, double *q_out, int32_t *success, int32_t *iters,
                                int32_t *searches, double *residual)
{
    IkLane<NJ> st;
    double Td[12], qs[NJ];
    struct QLocal {
        double *q;
        RTB_HD double get(int j) const { return q[j]; }
        RTB_HD void put(int j, double v) const { q[j] = v; }
    } qa{qs};
    ik_load_target([&](int k, double v) { Td[k] = v; }, Tep + 16 * tgt);
    const int s_last = ik_s_last(p);
    int it = 0;
    for (int s = ik_s_first(p);; ++s) {
        ik_search_begin<NJ>(st, qa, p, qlim, tgt, s, p.has_q0 ? q0 + (int64_t)NJ * tgt : nullptr);
        while (!st.fin) {
            ik_iter_any<NJ>(st, p, cv, qlim, [&](int k) { return Td[k]; }, qa);
            ik_settle<NJ>(st, p, qlim, qa);
        }
        it += st.iter;
        if (st.ok || s == s_last) {
            if (!st.ok) residual[tgt] = st.E;         // what ik_report leaves for the range's last search
            ik_emit<NJ>(st, qa, p, qlim, tgt, tgt, p.has_q0 != 0, st.ok != 0, it, q_out, success, iters, searches, residual);
            return;
        }
    }
}

}  // namespace rtbhip
