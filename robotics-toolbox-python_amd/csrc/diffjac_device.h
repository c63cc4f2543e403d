// diffjac_device.h -- manipulability and its Jacobian from a SUPPLIED Jacobian (and, for jacobm, optionally a supplied Hessian):
// the `J=` / `H=` forms of Robot.manipulability (robot/Robot.py:701-905, `if J is not None: w = [mfunc(self, J, q, axes_list)]` :896)
// and Robot.jacobm (robot/Robot.py:1101-1235, `verifymatrix(J, (6, n))` :1201, `H = self.hessian0(J0=J)` :1206).  Pure functions of J
// (and H): the per-lane bodies of diff_device.h on a Jacobian that arrives from memory instead of from the chain walk.
#pragma once
#include "diff_device.h"

namespace rtbhip {

// Jm[i] = m vec(J_a H_i,a^T)^T vec((J_a J_a^T)^-1) with the caller's H (Robot.py:1218-1233): H(i, b, k) = H[i][b][k] of the (n,6,n) tensor.
// G = (J_a J_a^T)^-1 J_a as in jacobm_factored (rows outside `axes` are zero, so the caller's excluded Hessian rows never count).
template <int NJ, class HG>
RTB_HD void jacobm_with_hessian(const double (&jac)[6 * NJ], int axes, HG H, double (&jm)[NJ])
{
    double B[6][6], dval[6], dinv[6];
    jjt_masked<NJ>(jac, axes, B);
    ldl_factor<6>(B, dval, dinv);
    const double m = manipulability_yoshikawa<NJ>(jac, axes);
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
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int k = 0; k < NJ; ++k) acc += H(i, b, k) * G[b * NJ + k];
        jm[i] = m * acc;
    }
}

}  // namespace rtbhip
