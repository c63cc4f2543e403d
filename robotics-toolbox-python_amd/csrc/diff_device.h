// diff_device.h -- per-lane differential-kinematics consumers of a finished Jacobian held in registers
// (slot r*NJ + j = row r of column j, as kin_reg.h leaves it):
//   jacob_dot      d/dt J = H(q) . qd   (reference Robot.jacob0_dot robot/Robot.py:964-1098:
//                  np.tensordot(hessian0(q), qd, (0, 0)); the Hessian blocks of core/methods.cpp:16-32 are
//                  formed on the fly and never stored)
//   manipulability Yoshikawa's measure sqrt|det(J_a J_a^T)|, or |det J_a| when J_a is square
//                  (reference ETS.manipulability robot/ETS.py:1687-1819, `yoshikawa` :1780-1787)
//   jacobm         manipulability Jacobian  Jm[i] = m vec(J H_i^T)^T vec((J J^T)^-1)
//                  (reference ETS.jacobm robot/ETS.py:1628-1685, Robot.jacobm robot/Robot.py:1120-1215)
// `axes` is a 6-bit row mask (bit r = Cartesian row r used): excluded rows are handled by replacing
// their row/column of J J^T by the identity (determinant and the remaining block of the inverse are
// unchanged), so every index stays static.
#pragma once
#include "kin_reg.h"
#include "ldl.h"

namespace rtbhip {

// H[j', :, i] as the reference fills it (methods.cpp:16-32): for j' <= i  (w_j' x v_i ; w_j' x w_i),
// for j' > i  (w_i x v_j' ; 0).
template <int NJ>
RTB_HD void jacob_dot(const double (&jac)[6 * NJ], const double (&qd)[NJ], double (&jd)[6 * NJ])
{
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const double vix = jac[i], viy = jac[NJ + i], viz = jac[2 * NJ + i];
        const double wix = jac[3 * NJ + i], wiy = jac[4 * NJ + i], wiz = jac[5 * NJ + i];
        double ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const double wjx = jac[3 * NJ + j], wjy = jac[4 * NJ + j], wjz = jac[5 * NJ + j];
            if (j <= i) {
                ax += qd[j] * (wjy * viz - wjz * viy); ay += qd[j] * (wjz * vix - wjx * viz); az += qd[j] * (wjx * viy - wjy * vix);
                bx += qd[j] * (wjy * wiz - wjz * wiy); by += qd[j] * (wjz * wix - wjx * wiz); bz += qd[j] * (wjx * wiy - wjy * wix);
            } else {
                const double vjx = jac[j], vjy = jac[NJ + j], vjz = jac[2 * NJ + j];
                ax += qd[j] * (wiy * vjz - wiz * vjy); ay += qd[j] * (wiz * vjx - wix * vjz); az += qd[j] * (wix * vjy - wiy * vjx);
            }
        }
        jd[i] = ax; jd[NJ + i] = ay; jd[2 * NJ + i] = az;
        jd[3 * NJ + i] = bx; jd[4 * NJ + i] = by; jd[5 * NJ + i] = bz;
    }
}

// B = J_a J_a^T embedded in 6x6 (excluded rows/columns -> identity)
template <int NJ>
RTB_HD void jjt_masked(const double (&jac)[6 * NJ], int axes, double (&B)[6][6])
{
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const bool ur = (axes >> r) & 1;
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const bool uc = (axes >> c) & 1;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NJ; ++k) s += jac[r * NJ + k] * jac[c * NJ + k];
            const double v = (ur && uc) ? s : (r == c ? 1.0 : 0.0);
            B[r][c] = v;
            B[c][r] = v;
        }
    }
}

template <int NJ>
RTB_HD double manipulability_yoshikawa(const double (&jac)[6 * NJ], int axes)
{
    if (NJ == 6 && (axes & 63) == 63) {          // square: |det J| (ETS.py:1782-1784)
        double a[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) a[r][c] = jac[r * NJ + (c < NJ ? c : 0)];
        return fabs(det_lu<6>(a));
    }
    double B[6][6];
    jjt_masked<NJ>(jac, axes, B);
    return sqrt(fabs(det_lu<6>(B)));              // ETS.py:1786-1787
}

// Smallest singular value of J_a (mode 1, ETS.py:1793-1796 `minsingular`) or 1/cond_2(J_a) = s_min / s_max
// (mode 2, :1789-1791 `condition`), from the eigenvalues of the smaller Gram matrix: J_a J_a^T when the row
// count does not exceed the joint count, else J_a^T J_a.
template <int NJ>
RTB_HD double manipulability_singular(const double (&jac)[6 * NJ], int axes, int mode)
{
    int rows = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) rows += (axes >> r) & 1;
    double lo = 1e300, hi = 0.0;
    if (rows <= NJ) {
        double B[6][6];
        jjt_masked<NJ>(jac, axes, B);
        jacobi_eigenvalues<6>(B);
#pragma unroll
        for (int r = 0; r < 6; ++r)
            if ((axes >> r) & 1) { lo = B[r][r] < lo ? B[r][r] : lo; hi = B[r][r] > hi ? B[r][r] : hi; }
    } else {
        double G[NJ][NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) s += ((axes >> r) & 1) ? jac[r * NJ + i] * jac[r * NJ + j] : 0.0;
                G[i][j] = s; G[j][i] = s;
            }
        jacobi_eigenvalues<NJ>(G);
#pragma unroll
        for (int i = 0; i < NJ; ++i) { lo = G[i][i] < lo ? G[i][i] : lo; hi = G[i][i] > hi ? G[i][i] : hi; }
    }
    lo = lo < 0.0 ? 0.0 : lo;
    return mode == 1 ? sqrt(lo) : sqrt(lo / hi);
}

// jacobm given the LDL^T factorisation (B, dinv) of the masked J J^T
template <int NJ>
RTB_HD void jacobm_factored(const double (&jac)[6 * NJ], int axes, const double (&B)[6][6], const double (&dinv)[6], double (&jm)[NJ])
{
    const double m = manipulability_yoshikawa<NJ>(jac, axes);     // Robot.py:1216-1222
    // G = (J_a J_a^T)^-1 J_a, column by column (rows outside `axes` come out zero)
    double G[6 * NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        double g[6], x[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) g[r] = ((axes >> r) & 1) ? jac[r * NJ + k] : 0.0;
        ldl_backsolve<6>(B, dinv, g, x);
#pragma unroll
        for (int r = 0; r < 6; ++r) G[r * NJ + k] = x[r];
    }
    // Jm[i] = m sum_{b,k} H[i,b,k] G[b,k]
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const double vix = jac[i], viy = jac[NJ + i], viz = jac[2 * NJ + i];
        const double wix = jac[3 * NJ + i], wiy = jac[4 * NJ + i], wiz = jac[5 * NJ + i];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const double vkx = jac[k], vky = jac[NJ + k], vkz = jac[2 * NJ + k];
            const double wkx = jac[3 * NJ + k], wky = jac[4 * NJ + k], wkz = jac[5 * NJ + k];
            if (k >= i) {       // H[i,:3,k] = w_i x v_k ; H[i,3:,k] = w_i x w_k
                acc += (wiy * vkz - wiz * vky) * G[k] + (wiz * vkx - wix * vkz) * G[NJ + k] + (wix * vky - wiy * vkx) * G[2 * NJ + k];
                acc += (wiy * wkz - wiz * wky) * G[3 * NJ + k] + (wiz * wkx - wix * wkz) * G[4 * NJ + k] + (wix * wky - wiy * wkx) * G[5 * NJ + k];
            } else {            // H[i,:3,k] = w_k x v_i ; H[i,3:,k] = 0
                acc += (wky * viz - wkz * viy) * G[k] + (wkz * vix - wkx * viz) * G[NJ + k] + (wkx * viy - wky * vix) * G[2 * NJ + k];
            }
        }
        jm[i] = m * acc;
    }
}

template <int NJ>
RTB_HD void jacobm(const double (&jac)[6 * NJ], int axes, double (&jm)[NJ])
{
    double B[6][6], dval[6], dinv[6];
    jjt_masked<NJ>(jac, axes, B);
    ldl_factor<6>(B, dval, dinv);
    jacobm_factored<NJ>(jac, axes, B, dinv, jm);
}

}  // namespace rtbhip
