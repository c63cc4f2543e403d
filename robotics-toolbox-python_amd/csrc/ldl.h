// ldl.h -- in-register solve of a small symmetric positive definite system  A x = g  by an LDL^T
// factorisation (no pivoting, no square roots).  Used for the damped normal equations of the LM step
// (ik_device.h; the reference forms an explicit inverse with a pivoted LU, core/ik.cpp:157-209) and for
// M(q) qdd = tau - tau_0 of the forward dynamics (dyn_kernels.hip; the reference calls
// numpy.linalg.solve, robot/Dynamics.py:505).  N is a compile-time size so every index is static.
#pragma once
#include "trig.h"

namespace rtbhip {

// A: lower triangle read (destroyed: holds L afterwards); g: right-hand side (destroyed); x: solution.
template <int N>
RTB_HD void ldl_solve(double (&A)[N][N], double (&g)[N], double (&x)[N])
{
    double dval[N], dinv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k] * dval[k];
        dval[j] = d;
        dinv[j] = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k] * dval[k];
            A[i][j] = v * dinv[j];
        }
    }
    // forward: L y = g ; diagonal ; backward: L^T x = z
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= A[i][k] * g[k];
        g[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] *= dinv[i];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double v = g[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) v -= A[k][i] * x[k];
        x[i] = v;
    }
}

}  // namespace rtbhip
