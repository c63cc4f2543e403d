// tests/emu/emu_dyn.h -- TEST INFRASTRUCTURE: the dynamics-term lane body on host arrays (templates; instantiated per joint-count range,
// emu_dyn_a..f.cpp: one unit with all 16 joint counts x 2 conventions x 3 modes took eight minutes to compile).
#pragma once
#include "emu_common.h"

namespace rtbhip { int rne_sig_enabled(); }      // rne_kernels.hip: rtbhip_tune("rne_sig")
// dynamics terms: the kernel's per-lane body (dyn_device.h) on host arrays laid out like the LDS tile rows
template <int NJ, bool MDH, int MODE>
static void dyn_run(const Dyn *d, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out)
{
    const DevLink *links = d->links.data();
    std::vector<double> in(3 * NJ), A(NJ * NJ);
    for (int64_t s = 0; s < N; ++s) {
        for (int j = 0; j < NJ; ++j) {
            in[j] = q[s * NJ + j];
            in[NJ + j] = qd ? qd[s * NJ + j] : 0.0;
            in[2 * NJ + j] = tq ? tq[s * NJ + j] : 0.0;
        }
        bool allrev = true;
        for (const DevLink &l : d->links) allrev = allrev && l.sigma == 0;
        // accel of an all-revolute chain with n >= 5: the kernel keeps q | qd | torque in the M tile itself (dyn_kernels.hip: alias_all)
        const bool alias_all = (MODE == kDynAccel && allrev && 3 * NJ <= NJ * (NJ + 1) / 2) || (MODE == kDynCoriolis && allrev && NJ >= 2);
        if (alias_all) for (int k = 0; k < (MODE == kDynAccel ? 3 : 2) * NJ; ++k) A[k] = in[k];
        const double *mine = alias_all ? A.data() : in.data();
        // as launch_dyn dispatches: the instantiation of the robot's structure signature, where this build has one (rne_device.h: kRneSig*)
        const RneSig sig = (rtbhip::rne_sig_enabled() && allrev) ? rne_signature(links, d->n) : 0;
        bool done = false;
        if constexpr (NJ == 7 && MDH) { if (sig == kRneSigPanda) { dyn_lane<7, true, MODE, true, kRneSigPanda>(links, mine, A.data(), g); done = true; } }
        if constexpr (NJ == 6 && !MDH) { if (sig == kRneSigPuma560) { dyn_lane<6, false, MODE, true, kRneSigPuma560>(links, mine, A.data(), g); done = true; } }
        if (done) {}
        else if (allrev) dyn_lane<NJ, MDH, MODE, true>(links, mine, A.data(), g);
        else dyn_lane<NJ, MDH, MODE, false>(links, mine, A.data(), g);
        const int W = MODE == kDynAccel ? NJ : NJ * NJ;
        if (MODE == kDynInertia && allrev) {           // packed lower triangle -> (n, n), as the kernel's flush_symmetric
            for (int r = 0; r < NJ; ++r)
                for (int c = 0; c < NJ; ++c) {
                    const int hi = r > c ? r : c, lo = r > c ? c : r;
                    out[s * W + r * NJ + c] = A[hi * (hi + 1) / 2 + lo];
                }
        } else {
            for (int k = 0; k < W; ++k) out[s * W + k] = A[k];
        }
    }
}
template <int NJ>
static void dyn_nj(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out)
{
    if (d->mdh) {
        if (mode == 0) dyn_run<NJ, true, kDynInertia>(d, q, qd, tq, N, g, out);
        else if (mode == 1) dyn_run<NJ, true, kDynCoriolis>(d, q, qd, tq, N, g, out);
        else dyn_run<NJ, true, kDynAccel>(d, q, qd, tq, N, g, out);
    } else {
        if (mode == 0) dyn_run<NJ, false, kDynInertia>(d, q, qd, tq, N, g, out);
        else if (mode == 1) dyn_run<NJ, false, kDynCoriolis>(d, q, qd, tq, N, g, out);
        else dyn_run<NJ, false, kDynAccel>(d, q, qd, tq, N, g, out);
    }
}
#define RTB_EMU_DYN(NJ) case NJ: dyn_nj<NJ>(d, mode, q, qd, tq, N, g, out); return 0;
#define RTB_EMU_DYN_DISPATCH(NAME, CASES) \
    int NAME(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out) \
    { switch (d->n) { CASES default: return -1; } }
int emu_dyn_r1(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out);
int emu_dyn_r2(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out);
int emu_dyn_r3(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out);
int emu_dyn_r4(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out);
int emu_dyn_r5(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out);
int emu_dyn_r6(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out);
