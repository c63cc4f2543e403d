// ldl.h -- in-register solve of a small symmetric positive definite system  A x = g  by an LDL^T
// factorisation (no pivoting, no square roots).  Used for the damped normal equations of the LM step
// (ik_device.h; the reference forms an explicit inverse with a pivoted LU, core/ik.cpp:157-209) and for
// M(q) qdd = tau - tau_0 of the forward dynamics (dyn_kernels.hip; the reference calls
// numpy.linalg.solve, robot/Dynamics.py:505).  N is a compile-time size so every index is static.
#pragma once
#include "trig.h"

namespace rtbhip {

// 1 / d for a pivot of an SPD factorisation: the hardware estimate (v_rcp_f64) and two Newton steps -- 5 instructions, within an ulp or two of
// the quotient -- instead of the correctly rounded division's ~14 (v_div_scale x2, v_rcp, the refinement chain, v_div_fmas, v_div_fixup).  The
// pivots of J^T W J + wn I are ordinary numbers (no denormals, no overflow to guard), and a pivot's reciprocal feeds products that are rounded
// again anyway.  d = 0 / inf / NaN still yield a non-finite result.  On the host (tests/emu) it is the plain quotient.
RTB_HD double rcp_pivot(double d)
{
#pragma clang fp contract(off)
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    return y;
#else
    return 1.0 / d;
#endif
}

// A: lower triangle read; on return holds L (unit diagonal implied) with 1/d_j in dinv and d_j in dval.
// FAST (the LM step of k_ik, where the solve is a fifth of every iteration): the pivots' reciprocals through rcp_pivot, and the products
// L_jk d_k of the textbook recurrence  d_j = a_jj - sum_k L_jk^2 d_k,  L_ij d_j = a_ij - sum_k L_ik L_jk d_k  are not re-formed term by term:
// u_jk = L_jk d_k is the value the recurrence holds BEFORE it divides by the pivot, so it is kept (in the unused upper triangle, A[k][j]) and
// every term is one fused multiply-add on it -- 56 multiplies fewer for N = 7, and each u_jk exact where L_jk d_k was a rounded product.
#ifndef RTB_LDL_KEEP_U
#define RTB_LDL_KEEP_U 1
#endif
// (fp contract(off), every fused multiply-add written: the recurrences are single chains the compiler would fuse the same way, but "would" is not
// a construction -- see kin_device.h, mix_pp)
template <int N, bool FAST = false>
RTB_HD void ldl_factor(double (&A)[N][N], double (&dval)[N], double (&dinv)[N])
{
#pragma clang fp contract(off)
    if constexpr (FAST && RTB_LDL_KEEP_U) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double d = A[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) d = __builtin_fma(-A[j][k], A[k][j], d);
            dval[j] = d;
            dinv[j] = rcp_pivot(d);
#pragma unroll
            for (int i = j + 1; i < N; ++i) {
                double v = A[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) v = __builtin_fma(-A[i][k], A[k][j], v);
                A[j][i] = v;                      // u_ij = L_ij d_j
                A[i][j] = v * dinv[j];
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = __builtin_fma(-(A[j][k] * A[j][k]), dval[k], d);
        dval[j] = d;
        dinv[j] = FAST ? rcp_pivot(d) : 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = __builtin_fma(-(A[i][k] * A[j][k]), dval[k], v);
            A[i][j] = v * dinv[j];
        }
    }
}

// forward: L y = g ; diagonal ; backward: L^T x = z   (g destroyed)
template <int N>
RTB_HD void ldl_backsolve(const double (&A)[N][N], const double (&dinv)[N], double (&g)[N], double (&x)[N])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v = __builtin_fma(-A[i][k], g[k], v);
        g[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] *= dinv[i];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double v = g[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) v = __builtin_fma(-A[k][i], x[k], v);
        x[i] = v;
    }
}

// A: lower triangle read (destroyed: holds L afterwards); g: right-hand side (destroyed); x: solution.
template <int N, bool FAST = false>
RTB_HD void ldl_solve(double (&A)[N][N], double (&g)[N], double (&x)[N])
{
    double dval[N], dinv[N];
    ldl_factor<N, FAST>(A, dval, dinv);
    ldl_backsolve<N>(A, dinv, g, x);
}

// Determinant of a symmetric positive SEMI-definite matrix (a Gram matrix J J^T): the product of the pivots of the LDL^T factorisation without
// pivoting (the u-keeping recurrence of ldl_factor<N, true>).  A pivot that is exactly zero -- the matrix is singular, and in exact arithmetic the
// column below it is zero too -- gets reciprocal 0 instead of infinity: the product, and with it the result, is an exact 0 rather than a NaN.
// A is destroyed.
template <int N>
RTB_HD double det_psd(double (&A)[N][N])
{
#pragma clang fp contract(off)
    double det = 1.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = __builtin_fma(-A[j][k], A[k][j], d);
        det *= d;
        if (j + 1 < N) {
            const double r = d != 0.0 ? rcp_pivot(d) : 0.0;
#pragma unroll
            for (int i = j + 1; i < N; ++i) {
                double v = A[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) v = __builtin_fma(-A[i][k], A[k][j], v);
                A[j][i] = v;
                A[i][j] = v * r;
            }
        }
    }
    return det;
}

// Determinant of a general N x N matrix by LU with partial pivoting (what numpy.linalg.det does through
// LAPACK getrf).  The per-lane row exchange is a compare-and-swap chain on statically indexed
// registers: after processing rows k+1..N-1 the largest |a[i][k]| sits in row k.  a is destroyed.
template <int N>
RTB_HD double det_lu(double (&a)[N][N])
{
    double det = 1.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const bool sw = fabs(a[i][k]) > fabs(a[k][k]);
#pragma unroll
            for (int c = k; c < N; ++c) {
                const double u = a[k][c], v = a[i][c];
                a[k][c] = sw ? v : u;
                a[i][c] = sw ? u : v;
            }
            det = sw ? -det : det;
        }
        const double piv = a[k][k];
        det *= piv;
        const double inv = piv != 0.0 ? 1.0 / piv : 0.0;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double f = a[i][k] * inv;
#pragma unroll
            for (int c = k + 1; c < N; ++c) a[i][c] -= f * a[k][c];
        }
    }
    return det;
}

// Eigenvalues of a symmetric N x N matrix by cyclic Jacobi rotations with a fixed number of sweeps
// (uniform control flow, every index static).  a is destroyed; its diagonal holds the eigenvalues on
// return.  A zero off-diagonal element leaves its pair untouched (t = 0), so identity-embedded rows keep
// their diagonal value exactly.  7 sweeps reach fp64 round-off for N <= 8 (quadratic convergence).
template <int N>
RTB_HD void jacobi_eigenvalues(double (&a)[N][N])
{
#pragma clang fp contract(off)
#pragma unroll 1
    for (int sweep = 0; sweep < 7; ++sweep) {
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const double apq = a[p][q];
                const bool skip = apq == 0.0;
                const double tau = (a[q][q] - a[p][p]) / (2.0 * (skip ? 1.0 : apq));
                double t = 1.0 / (fabs(tau) + sqrt(1.0 + tau * tau));
                t = tau < 0.0 ? -t : t;
                t = skip ? 0.0 : t;
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                a[p][p] -= t * apq;
                a[q][q] += t * apq;
                a[p][q] = 0.0;
                a[q][p] = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    if (k != p && k != q) {
                        const double akp = a[k][p], akq = a[k][q];
                        const double np_ = __builtin_fma(c, akp, -(s * akq)), nq_ = __builtin_fma(s, akp, c * akq);      // (written out: kin_device.h, mix_pp)
                        a[k][p] = np_; a[p][k] = np_;
                        a[k][q] = nq_; a[q][k] = nq_;
                    }
                }
            }
        }
    }
}

// A x = b for a general N x N matrix in MEMORY (row-major, destroyed; b becomes x): Gaussian elimination with partial pivoting, run-time indices,
// no unrolling -- what numpy.linalg.solve (LAPACK gesv) does, for the one case in which the matrix of Dynamics.accel is not symmetric (dyn_device.h:
// a modified-DH chain whose first joint is prismatic).  Not a fast path.
template <int N>
RTB_HD void lu_solve_mem(double *A, double *b)
{
#pragma unroll 1
    for (int c = 0; c < N; ++c) {
        int p = c;
        double best = fabs(A[c * N + c]);
#pragma unroll 1
        for (int r = c + 1; r < N; ++r) {
            const double v = fabs(A[r * N + c]);
            if (v > best) { best = v; p = r; }
        }
        if (p != c) {
#pragma unroll 1
            for (int k = 0; k < N; ++k) { const double t = A[c * N + k]; A[c * N + k] = A[p * N + k]; A[p * N + k] = t; }
            const double t = b[c]; b[c] = b[p]; b[p] = t;
        }
        const double piv = A[c * N + c];
#pragma unroll 1
        for (int r = c + 1; r < N; ++r) {
            const double f = A[r * N + c] / piv;
#pragma unroll 1
            for (int k = c + 1; k < N; ++k) A[r * N + k] -= f * A[c * N + k];
            b[r] -= f * b[c];
        }
    }
#pragma unroll 1
    for (int r = N - 1; r >= 0; --r) {
        double acc = b[r];
#pragma unroll 1
        for (int k = r + 1; k < N; ++k) acc -= A[r * N + k] * b[k];
        b[r] = acc / A[r * N + r];
    }
}

}  // namespace rtbhip
