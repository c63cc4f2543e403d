// tests/emu/emu_tree.h -- TEST INFRASTRUCTURE: the tree dynamics lane body (tree_device.h tree_dyn_lane) on host arrays; instantiated for 1..12 joints in
// emu_misc.cpp and for 13..20 in emu_tree_big.cpp (the long pole of the host build: its own unit, -O1).
#pragma once
#include "emu_common.h"

// Dynamics-mixin terms of an ETS robot: tree_device.h's tree_dyn_lane on the CPU (mode 0 inertia, 1 coriolis, 2 accel)
template <int NG, int MODE, SegSig SIG = 0, TreeTopo TOPO = 0, SegSig SIG2 = 0>
static void tree_dyn_run(const Tree *t, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out)
{
    std::vector<double> slots((size_t)kTreeBilinearSlotDoubles * std::max(1, t->nslots));
    constexpr int W = MODE == kDynAccel ? NG : NG * NG;
    for (int64_t s = 0; s < N; ++s) {
        double mine[3 * NG], A[NG * NG + NG];
        for (int j = 0; j < NG; ++j) { mine[j] = q[s * NG + j]; mine[NG + j] = qd ? qd[s * NG + j] : 0.0; mine[2 * NG + j] = tq ? tq[s * NG + j] : 0.0; }
        tree_dyn_lane<NG, MODE, TreeKnown<SIG, TOPO, SIG2>>(t->groups.data(), t->nslots, mine, A, g, [&](int i) -> double & { return slots[i]; });
        if (MODE == kDynInertia) {                  // the kernel's flush: packed lower triangle -> the full symmetric matrix, rows where the reference has them
            for (int r0 = 0; r0 < NG; ++r0) {
                const int r = tree_row_position<NG>(t->groups.data(), r0);
                for (int c = 0; c < NG; ++c) out[s * W + r0 * NG + c] = A[(r > c ? r : c) * ((r > c ? r : c) + 1) / 2 + (r > c ? c : r)];
            }
            continue;
        }
        for (int k = 0; k < W; ++k) out[s * W + k] = A[k];
    }
}
template <int NG, SegSig SIG = 0, TreeTopo TOPO = 0, SegSig SIG2 = 0>
static void tree_dyn_mode(int mode, const Tree *t, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out)
{
    if (mode == 0) tree_dyn_run<NG, kDynInertia, SIG, TOPO, SIG2>(t, q, qd, tq, N, g, out);
    else if (mode == 1) tree_dyn_run<NG, kDynCoriolis, SIG, TOPO, SIG2>(t, q, qd, tq, N, g, out);
    else tree_dyn_run<NG, kDynAccel, SIG, TOPO, SIG2>(t, q, qd, tq, N, g, out);
}
int emu_tree_dyn_big(const Tree *t, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out);
