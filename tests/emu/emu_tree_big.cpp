// tests/emu/emu_tree_big.cpp -- TEST INFRASTRUCTURE: tree dynamics terms of robots with 13..20 joints replayed on the CPU (emu_tree.h).
#include "emu_tree.h"

int emu_tree_dyn_big(const Tree *t, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out)
{
    switch (t->n) {
    case 13: tree_dyn_mode<13>(mode, t, q, qd, tq, N, g, out); break;
    case 14: tree_dyn_mode<14>(mode, t, q, qd, tq, N, g, out); break;
    case 15: tree_dyn_mode<15>(mode, t, q, qd, tq, N, g, out); break;
    case 16: tree_dyn_mode<16>(mode, t, q, qd, tq, N, g, out); break;
    case 17: tree_dyn_mode<17>(mode, t, q, qd, tq, N, g, out); break;
    case 18: tree_dyn_mode<18>(mode, t, q, qd, tq, N, g, out); break;
    case 19: tree_dyn_mode<19>(mode, t, q, qd, tq, N, g, out); break;
    case 20: tree_dyn_mode<20>(mode, t, q, qd, tq, N, g, out); break;
    default: return -2;
    }
    return 0;
}
