// dyn_device.h -- per-lane body of the dynamics-term kernels (dyn_kernels.hip): all the Newton-Euler
// passes the reference's Dynamics.inertia / coriolis / accel make for ONE configuration
// (robot/Dynamics.py:704-861, 424-509), on the rne_lane recursion of rne_device.h.
// __host__ __device__ so tests/emu can run the same code lane by lane on the CPU.
#pragma once
#include "rne_device.h"
#include "ldl.h"

namespace rtbhip {

enum { kDynInertia = 0, kDynCoriolis = 1, kDynAccel = 2 };

// mine : this lane's inputs  [q (n) | qd (n) | torque (n)]   (what the mode needs)
// mA   : n x n work/output tile (row-major): M for inertia / accel (accel leaves qdd in mA[0..n-1]), C for coriolis
// mB   : n x n scratch (coriolis only): Csq
// With sin/cos fixed across the passes, most of the forward recursion becomes loop-invariant in the
// compiler's eyes and LICM hoists it out of the pass loop -- into ~100 VGPRs the kernel does not have
// (first build: 428 B/lane of scratch, inertia 0.54 -> 0.90 ms).  Making the trig values opaque once per
// pass keeps every pass self-contained.
template <int NJ>
RTB_HD void dyn_opaque(double (&st)[NJ], double (&ct)[NJ])
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(st[j]), "+v"(ct[j]));
#endif
}

template <int NJ, bool MDH, int MODE, bool ALLREV, class LinksP>
RTB_HD void dyn_lane(LinksP links, const double *mine, double *mA, double *mB, V3 grav)
{
    const V3 zero = v3(0, 0, 0);
    auto qin = [&](int j) { return mine[j]; };
    double st[NJ], ct[NJ];
    rne_trig<NJ, ALLREV>(links, qin, st, ct);      // every pass below is at the same q: sin/cos once
    if (MODE == kDynInertia || MODE == kDynAccel) {
        // row i of the result = tau for qdd = e_i, qd = 0, no gravity (Dynamics.py:752-758, :492-496)
#pragma unroll 1
        for (int i = 0; i < NJ; ++i) {
            dyn_opaque<NJ>(st, ct);
            rne_core<NJ, MDH, true, ALLREV, true>(links, NJ, st, ct, zero, zero, zero, qin, [&](int) { return 0.0; },
                                    [&](int j) { return j == i ? 1.0 : 0.0; },
                                    [&](int j, double v) { mA[i * NJ + j] = v; });
        }
    }
    if (MODE == kDynAccel) {
        // tau_0 = rne(q, qd, 0) with gravity and friction (Dynamics.py:500), then M qdd = torque - tau_0
        double b[NJ], x[NJ], M[NJ][NJ];
        dyn_opaque<NJ>(st, ct);
        rne_core<NJ, MDH, true, ALLREV, true>(links, NJ, st, ct, grav, zero, zero, qin, [&](int j) { return mine[NJ + j]; },
                                [&](int) { return 0.0; }, [&](int j, double v) { b[j] = mine[2 * NJ + j] - v; });
#pragma unroll
        for (int r = 0; r < NJ; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) M[r][c] = mA[r * NJ + c];
        ldl_solve<NJ>(M, b, x);
#pragma unroll
        for (int j = 0; j < NJ; ++j) mA[j] = x[j];        // the tile's first n slots become the output row
    }
    if (MODE == kDynCoriolis) {
        // centripetal: QD = e_i -> Csq[:, i] (Dynamics.py:828-833), friction removed (:820)
#pragma unroll 1
        for (int i = 0; i < NJ; ++i) {
            dyn_opaque<NJ>(st, ct);
            rne_core<NJ, MDH, false, ALLREV, true>(links, NJ, st, ct, zero, zero, zero, qin, [&](int j) { return j == i ? 1.0 : 0.0; },
                                     [&](int) { return 0.0; }, [&](int r, double v) { mB[r * NJ + i] = v; });
        }
#pragma unroll
        for (int k = 0; k < NJ * NJ; ++k) mA[k] = 0.0;
        // Coriolis: QD = e_i + e_j, i < j (Dynamics.py:839-854), same accumulation order
#pragma unroll 1
        for (int i = 0; i < NJ; ++i) {
#pragma unroll 1
            for (int j = i + 1; j < NJ; ++j) {
                const double qdi = mine[NJ + i], qdj = mine[NJ + j];
                dyn_opaque<NJ>(st, ct);
                rne_core<NJ, MDH, false, ALLREV, true>(links, NJ, st, ct, zero, zero, zero, qin,
                                         [&](int k) { return (k == i || k == j) ? 1.0 : 0.0; }, [&](int) { return 0.0; },
                                         [&](int r, double tau) {
                                             const double t = tau - mB[r * NJ + j] - mB[r * NJ + i];
                                             mA[r * NJ + j] = mA[r * NJ + j] + t * qdi / 2;
                                             mA[r * NJ + i] = mA[r * NJ + i] + t * qdj / 2;
                                         });
            }
        }
#pragma unroll
        for (int r = 0; r < NJ; ++r)                              // + Csq diag(qd) (Dynamics.py:856)
#pragma unroll
            for (int c = 0; c < NJ; ++c) mA[r * NJ + c] = mA[r * NJ + c] + mB[r * NJ + c] * mine[NJ + c];
    }
}

}  // namespace rtbhip
