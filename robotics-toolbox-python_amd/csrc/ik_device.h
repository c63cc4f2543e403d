// ik_device.h -- per-lane Levenberg-Marquardt inverse kinematics, the whole search loop resident on
// the device.
//
// Replaces _IK_loop + _IK_LM_Chan/Wampler/Sugihara (core/ik.cpp:19-75,157-209), _angle_axis
// (ik.cpp:241-286), _rand_q (ik.cpp:288-299), _check_lim (ik.cpp:227-239) and, as "flavour 1", the
// Python solver behind ikine_LM (robot/IK.py:297-367 `_solve`, :994-1017 `IK_LM.step`).
//
// One lane owns one target pose.  The reference's nested while-loops are restated as a per-lane
// state machine whose transition (`ik_advance`) performs exactly ONE LM iteration: every lane of a
// wave executes the same instruction stream (FK + Jacobian + 6-vector error + normal equations +
// solve) regardless of which search/iteration it is in, so lanes that converge early can be handed
// a new target (persistent lanes, ik_kernels.hip) without divergence.
//
// Differences from the reference that cannot be bit-matched and are covered by statistical parity
// (SURVEY.md 8c): restarts come from a counter-based generator keyed by (seed, target, draw, joint)
// instead of an unseeded std::rand; the damped normal equations are solved by an LDL^T
// factorisation in registers (A = J^T W J + wn I is symmetric positive definite for wn > 0) instead
// of forming the explicit inverse with a pivoted LU.
#pragma once
#include "kin_reg.h"

namespace rtbhip {

constexpr double kIkPi = 3.14159265358979323846264338327950288;   // linalg.h:19
constexpr double kIkPi2 = 6.283185307179586;                      // linalg.h:20
constexpr double kIkPiHalf = 1.57079632679489661923132169163975144;

struct IkDev {   // wave-uniform solver parameters (kernarg)
    int32_t ilimit, slimit, reject_jl, method, flavour, has_q0;
    double tol, lambda;
    double we[6];
    double tail[12];
    uint64_t seed;
    int64_t N;
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
RTB_HD void ik_angle_axis(const Pose &P, const double (&Td)[12], double (&e)[6])
{
    e[0] = Td[9] - P.tx; e[1] = Td[10] - P.ty; e[2] = Td[11] - P.tz;
    // R = Rd * Re^T ; only the entries the formula reads
    const double r00 = Td[0] * P.r00 + Td[1] * P.r01 + Td[2] * P.r02;
    const double r01 = Td[0] * P.r10 + Td[1] * P.r11 + Td[2] * P.r12;
    const double r02 = Td[0] * P.r20 + Td[1] * P.r21 + Td[2] * P.r22;
    const double r10 = Td[3] * P.r00 + Td[4] * P.r01 + Td[5] * P.r02;
    const double r11 = Td[3] * P.r10 + Td[4] * P.r11 + Td[5] * P.r12;
    const double r12 = Td[3] * P.r20 + Td[4] * P.r21 + Td[5] * P.r22;
    const double r20 = Td[6] * P.r00 + Td[7] * P.r01 + Td[8] * P.r02;
    const double r21 = Td[6] * P.r10 + Td[7] * P.r11 + Td[8] * P.r12;
    const double r22 = Td[6] * P.r20 + Td[7] * P.r21 + Td[8] * P.r22;
    const double lx = r21 - r12, ly = r02 - r20, lz = r10 - r01;
    const double nrm = sqrt(lx * lx + ly * ly + lz * lz);
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

// ---------------------------------------------------------------- one LM step
// dq = (J^T W J + wn I)^-1 J^T W e, J in registers (slot r*NJ + j), W = diag(we).
template <int NJ>
RTB_HD void ik_lm_step(const double (&jac)[6 * NJ], const double (&e)[6], const double *we, double wn,
                       double (&dq)[NJ])
{
    double A[NJ][NJ];   // lower triangle used; after factorisation holds L (unit diagonal implied)
    double g[NJ], dval[NJ], dinv[NJ];
    double we_e[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) we_e[k] = we[k] * e[k];
#pragma unroll
    for (int r = 0; r < NJ; ++r) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += jac[k * NJ + r] * we_e[k];
        g[r] = s;
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) a += (jac[k * NJ + r] * we[k]) * jac[k * NJ + c];
            A[r][c] = (r == c) ? a + wn : a;
        }
    }
    // LDL^T: A = L D L^T
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k] * dval[k];
        dval[j] = d;
        dinv[j] = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < NJ; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k] * dval[k];
            A[i][j] = v * dinv[j];
        }
    }
    // forward: L y = g ; diagonal ; backward: L^T x = z
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        double v = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= A[i][k] * g[k];
        g[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) g[i] *= dinv[i];
#pragma unroll
    for (int i = NJ - 1; i >= 0; --i) {
        double v = g[i];
#pragma unroll
        for (int k = i + 1; k < NJ; ++k) v -= A[k][i] * dq[k];
        dq[i] = v;
    }
}

// ---------------------------------------------------------------- per-lane solver state
template <int NJ>
struct IkState {
    double q[NJ];
    double Td[12];     // target: R row-major (9), t (3)
    double E;
    int64_t tgt;       // target index, -1 = lane idle
    int32_t iter, search, it, draws;
};

// Start a target.  Tep16: row-major 4x4 as the caller holds it (IK_LM_c input, fknm.cpp:465-472).
template <int NJ, class QL>
RTB_HD void ik_begin(IkState<NJ> &st, const IkDev &p, QL qlim, int64_t tgt, const double *Tep16,
                     const double *q0row)
{
    st.tgt = tgt;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) st.Td[3 * r + c] = Tep16[4 * r + c];
        st.Td[9 + r] = Tep16[4 * r + 3];
    }
    st.E = 0.0;
    st.it = 0;
    st.draws = 0;
    if (p.flavour == 0) { st.iter = 1; st.search = 1; }    // ik.cpp:39, fknm.cpp:406
    else { st.iter = 0; st.search = 0; }                    // IK.py:299-313
    if (q0row) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) st.q[j] = q0row[j];
        if (p.flavour == 1) st.draws = 1;                   // row 0 of the pre-drawn starts is replaced by q0 (IK.py:229-232)
    } else {
        ik_restart<NJ>(p.seed, tgt, st.draws, qlim, st.q);
        st.draws++;
    }
}

RTB_HD double ik_wrap_c(double q) { return fmod(q + kIkPi, kIkPi2) - kIkPi; }        // ik.cpp:51
RTB_HD double ik_wrap_py(double q)                                                      // IK.py:331
{
    double r = fmod(q + kIkPi, 2 * kIkPi);
    if (r < 0) r += 2 * kIkPi;
    return r - kIkPi;
}

// One LM iteration + the reference's loop bookkeeping.  Returns true when the target is finished;
// then (st.q, success, st.it, st.search, st.E) are the 5 outputs.
template <int NJ, class CV, class QL>
RTB_HD bool ik_advance(IkState<NJ> &st, const IkDev &p, const CV &cv, QL qlim, int &success)
{
    Pose P;
    double jac[6 * NJ], e[6], dq[NJ];
    reg_core<NJ, true>(cv, p.tail, 0, st.q, P, jac);           // ik.cpp:44,56 / IK.py:994,1009
    ik_angle_axis(P, st.Td, e);
    double E = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) E += e[k] * p.we[k] * e[k];
    E *= 0.5;                                                   // ik.cpp:46
    st.E = E;
    const double wn = (p.method == 1) ? p.lambda : (p.method == 2) ? E + p.lambda : p.lambda * E;   // ik.cpp:169,183,205
    ik_lm_step<NJ>(jac, e, p.we, wn, dq);
    const bool arrived = E < p.tol;
    bool finished = false;
    success = 0;
    if (p.flavour == 0) {
        bool end_search = false;
        if (arrived) {                                          // ik.cpp:48-54
            bool ok = true;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                st.q[j] = ik_wrap_c(st.q[j]);
                if (st.q[j] < qlim[j] || st.q[j] > qlim[NJ + j]) ok = false;   // ik.cpp:227-239
            }
            if (!p.reject_jl) ok = true;
            st.it += st.iter;
            if (ok) { success = 1; finished = true; }
            else end_search = true;
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) st.q[j] += dq[j];      // ik.cpp:57
            st.iter++;
            if (st.iter > p.ilimit) { st.it += st.iter; end_search = true; }
        }
        if (end_search) {                                       // ik.cpp:66-69
            st.iter = 0;
            st.search++;
            ik_restart<NJ>(p.seed, st.tgt, st.draws, qlim, st.q);
            st.draws++;
            if (st.search > p.slimit) finished = true;
        }
    } else {
        bool end_search = false;
        st.iter++;                                              // IK.py:315
#pragma unroll
        for (int j = 0; j < NJ; ++j) st.q[j] += dq[j];          // the step is taken before E is tested (IK.py:319-327)
        if (arrived) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                st.q[j] = ik_wrap_py(st.q[j]);
                if (st.q[j] < qlim[j] || st.q[j] > qlim[NJ + j]) ok = false;
            }
            if (ok || !p.reject_jl) {                           // IK.py:336-349
                st.it += st.iter;
                st.search += 1;
                success = 1;
                finished = true;
            } else {
                end_search = true;
            }
        } else if (st.iter >= p.ilimit) {
            end_search = true;
        }
        if (end_search) {                                       // IK.py:351
            st.it += st.iter;
            st.iter = 0;
            st.search++;
            if (st.search >= p.slimit) {
                st.search = p.slimit;                           // IK.py:359-366
                finished = true;
            } else {
                ik_restart<NJ>(p.seed, st.tgt, st.search, qlim, st.q);
            }
        }
    }
    return finished;
}

}  // namespace rtbhip
