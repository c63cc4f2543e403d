// tree_dyn_kernels.hip -- gfx950 kernels for the Dynamics-mixin terms of ETS robots (link trees): Dynamics.inertia / coriolis / accel over Robot.rne
// (reference robot/Dynamics.py:704-861, 424-509 on robot/Robot.py:1704-1903).  Split from tree_kernels.hip (k_tree_rne) for build time only.
#include "tree_kernels.h"

namespace rtbhip {

// ---- Dynamics-mixin terms of an ETS robot: inertia / coriolis / accel, every pass of a configuration in one lane (tree_device.h:
// tree_dyn_lane).  LDS per lane: the inputs the mode reads ([q] | [q, qd] | [q, qd, torque]) and an n x n tile, plus the tree's slots;
// inputs arrive and results leave through the same coalesced tile copies as k_tree_rne.  Robots of up to 20 joints (13..20 -- YuMi's two arms, 14,
// with its grippers 18 -- hold a 100+ KB tile per wave, the largest ones for 32 configurations only, and spill part of their state: served, not fast).
constexpr int kTreeDynMax = 20;

// packed lower triangles (row r, column c <= r at r (r + 1) / 2 + c) of ncfg lanes -> the full symmetric (n, n) matrices, one contiguous run
// (row r of the reference's matrix is row tree_row_position(r) of the group-ordered one: tree_device.h; the identity for robots numbered in group order)
template <int NG>
__device__ __forceinline__ void tree_flush_symmetric(ConstGroups groups, const double *rows, int stride, int ncfg, double *__restrict__ dst, int lane)
{
    constexpr int W = NG * NG;
    const int total = ncfg * W;
    const bool ordered = tree_in_group_order<NG>(groups);
    auto at = [&](int f) {
        const int cfg = f / W, rem = f - cfg * W, r0 = rem / NG, c = rem - r0 * NG;
        const int r = ordered ? r0 : tree_row_position<NG>(groups, r0);
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        return rows[cfg * stride + hi * (hi + 1) / 2 + lo];
    };
    for (int f = 2 * lane; f < total; f += 2 * kWave) {
        const double a = at(f);
        if (f + 1 < total) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d w = {a, at(f + 1)};
            __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + f));
        } else {
            __builtin_nontemporal_store(a, dst + f);
        }
    }
}

// Robots of up to 7 joints keep two waves per SIMD (the second launch bound: at most 256 registers a lane): their tiles leave room for five or
// more waves on a CU, and an allocation just above 256 -- the general accel kernel for six joints took 288 in one build of round 5, when the
// recursion read a group's bookkeeping words before it needed them -- halves what the registers admit.  From 8 joints on the tile admits four.
template <int NG, int MODE, class KN = TreeNothing>
__global__ __launch_bounds__(kWave, (NG <= 7 ? 2 : 1)) void k_tree_dyn(TreeParams tp, const DevGroup *groups_g, const double *__restrict__ q,
                                                       const double *__restrict__ qd, const double *__restrict__ tq,
                                                       double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    ConstGroups groups = (ConstGroups)groups_g;
    const int lane = threadIdx.x;
    constexpr int K = MODE == kDynInertia ? 1 : (MODE == kDynCoriolis ? 2 : 3);
    // accel and inertia: packed lower triangle (accel: >= n doubles, qdd leaves from its head); coriolis: the full n x n tile
    constexpr int W = MODE == kDynCoriolis ? NG * NG : NG * (NG + 1) / 2 + (MODE == kDynAccel ? NG : 0);
    constexpr int in_stride = (K * NG) | 1, w_stride = W | 1;
    const int T = tp.tile;
    double *A = lds + T * in_stride;
    double *slots = A + T * w_stride;
    const int64_t cfg0 = (int64_t)blockIdx.x * T;
    const int64_t left = tp.N - cfg0;
    const int ncfg = left < T ? (int)left : T;
    const int count = ncfg * NG;
    {
        const double *src[3] = {q + cfg0 * NG, qd ? qd + cfg0 * NG : nullptr, tq ? tq + cfg0 * NG : nullptr};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double r[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) { const int f = lane + kWave * i; r[i] = (f < count && src[k]) ? src[k][f] : 0.0; }
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int f = lane + kWave * i;
                const int row = f / NG, c = f - row * NG;
                lds[row * in_stride + k * NG + c] = r[i];
            }
        }
    }
    __syncthreads();
    if (lane < ncfg)
        tree_dyn_lane<NG, MODE, KN>(groups, tp.nslots, lds + lane * in_stride, A + lane * w_stride, v3(tp.grav[0], tp.grav[1], tp.grav[2]),
                                [&](int i) -> double & { return slots[i * T + lane]; });
    __syncthreads();
    if (MODE == kDynAccel) flush_run(A, w_stride, NG, ncfg, out + cfg0 * NG, lane);
    else if (MODE == kDynInertia) tree_flush_symmetric<NG>(groups, A, w_stride, ncfg, out + cfg0 * (NG * NG), lane);
    else flush_run(A, w_stride, NG * NG, ncfg, out + cfg0 * (NG * NG), lane);
}

#if RTB_HOST_SIDE      // the launchers (the kernel above is also what jit.cpp hands to hipRTC, one instantiation at a time)
hipFunction_t tree_jit_function(const Tree *t, int variant);      // tree_kernels.hip
// tile size and LDS bytes of k_tree_dyn for a run-time group count: the formulas of launch_tree_dyn_one (which has them at compile time)
static void tree_dyn_tile(int n, int mode, int nslots, int32_t *tile, size_t *lds)
{
    const int K = mode == kDynInertia ? 1 : (mode == kDynCoriolis ? 2 : 3);
    const int W = mode == kDynCoriolis ? n * n : n * (n + 1) / 2 + (mode == kDynAccel ? n : 0);
    const size_t per_lane = (size_t)(((K * n) | 1) + (W | 1) + (mode == kDynCoriolis ? kTreeBilinearSlotDoubles : kTreeSlotDoubles) * nslots) * sizeof(double);
    int tl = kWave;
    while (tl > 8 && per_lane * tl > 160 * 1024) tl /= 2;          // 64 / 32 / 16 / 8 configurations per wave (the other lanes idle): served, not fast
    *tile = tl;
    *lds = per_lane * tl;
}
template <int NG, int MODE, SegSig SIG = 0, TreeTopo TOPO = 0, SegSig SIG2 = 0>
static hipError_t launch_tree_dyn_one(dim3 grid, hipStream_t s, int nslots, const TreeParams &tp, const DevGroup *g, const double *q,
                                      const double *qd, const double *tq, double *out, size_t *lds_out)
{
    constexpr int K = MODE == kDynInertia ? 1 : (MODE == kDynCoriolis ? 2 : 3);
    constexpr int W = MODE == kDynCoriolis ? NG * NG : NG * (NG + 1) / 2 + (MODE == kDynAccel ? NG : 0);
    const size_t per_lane = (size_t)(((K * NG) | 1) + (W | 1) + (MODE == kDynCoriolis ? kTreeBilinearSlotDoubles : kTreeSlotDoubles) * nslots) * sizeof(double);
    // 64 configurations per wave; a robot whose tile would not fit a CU's 160 KB (15-16 joints with several branch points: the n x n tile
    // of coriolis plus 24 doubles per branch slot) runs 32 per wave, the upper half of the lanes idle: served, not fast
    TreeParams tq_ = tp;
    tq_.tile = kWave;
    if (per_lane * kWave > 160 * 1024) tq_.tile = kWave / 2;
    const size_t lds = per_lane * tq_.tile;
    *lds_out = lds;
    if (lds > 160 * 1024) return hipSuccess;          // reported by the caller
    const int64_t tiles = (tp.N + tq_.tile - 1) / tq_.tile;
    if (tiles > 0x7fffffff) { *lds_out = 0; return hipErrorInvalidValue; }
    grid = dim3((unsigned)tiles);
    auto k = k_tree_dyn<NG, MODE, TreeKnown<SIG, TOPO, SIG2>>;
    if (lds > 48 * 1024) { hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k, grid, dim3(kWave), lds, s, tq_, g, q, qd, tq, out);
    note_launch((int)grid.x, kWave, (int)lds);
    return hipSuccess;
}

template <int NG, SegSig SIG = 0, TreeTopo TOPO = 0, SegSig SIG2 = 0>
static hipError_t launch_tree_dyn_ng(int mode, dim3 grid, hipStream_t s, int nslots, const TreeParams &tp, const DevGroup *g, const double *q,
                                     const double *qd, const double *tq, double *out, size_t *lds, bool plain = false)
{
    if constexpr (SIG == 0 && NG <= kTreePlainChainMax) {
        if (plain) return launch_tree_dyn_ng<NG, kTreeSigPlainChain>(mode, grid, s, nslots, tp, g, q, qd, tq, out, lds);
    }
    if (mode == kDynInertia) return launch_tree_dyn_one<NG, kDynInertia, SIG, TOPO, SIG2>(grid, s, nslots, tp, g, q, qd, tq, out, lds);
    if (mode == kDynCoriolis) return launch_tree_dyn_one<NG, kDynCoriolis, SIG, TOPO, SIG2>(grid, s, nslots, tp, g, q, qd, tq, out, lds);
    return launch_tree_dyn_one<NG, kDynAccel, SIG, TOPO, SIG2>(grid, s, nslots, tp, g, q, qd, tq, out, lds);
}

int launch_tree_dyn(const Tree *t, const DevGroup *groups, int mode, const double *q, const double *qd, const double *tq, int64_t N,
                    const double *grav3, double *out, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (t->n > RTBHIP_MAX_JOINTS) { set_error("tree inertia/coriolis/accel: more than RTBHIP_MAX_JOINTS joints"); return RTBHIP_ELIMIT; }
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("tree inertia/coriolis/accel: batch too large for one launch"); return RTBHIP_ELIMIT; }
    TreeParams tp;
    tp.n = t->n; tp.nslots = t->nslots; tp.N = N; tp.tile = kWave; tp.pad_ = 0;
    for (int i = 0; i < 3; i++) tp.grav[i] = grav3 ? grav3[i] : 0.0;
    dim3 grid((unsigned)tiles);
    size_t lds = 0;
    hipError_t e = hipSuccess;
    const int g_tree_sig = tree_sig_enabled();
    const SegSig sig = g_tree_sig ? t->sig : 0;
    const bool plain = (sig & kTreeSigPlain) != 0;
    const TreeTopo topo = g_tree_sig ? t->topo : 0;
    const SegSig sig2 = g_tree_sig ? t->sig2 : 0;
    // a robot without a built-in instantiation: its own, compiled at run time (jit.cpp); the general kernels below serve until it is there
    hipFunction_t f = tree_jit_function(t, 2 + mode);
    if (!f && t->n > kTreeDynMax) {
        // beyond the built-in sizes (1 .. 20): the general kernel of this size, instantiated at run time; the caller waits (robot/Dynamics.py:704-861)
        f = t->jit.get_wait("tree_dyn_kernels.hip", 10 + mode, [&] { return "rtbhip::k_tree_dyn<" + std::to_string(t->n) + ", " + std::to_string(mode) + ", rtbhip::TreeNothing>"; });
        if (!f) return RTBHIP_ELIMIT;
    }
    if (f) {
        TreeParams tq_ = tp;
        size_t l = 0;
        tree_dyn_tile(t->n, mode, t->nslots, &tq_.tile, &l);
        if (l > 160 * 1024 && t->n > kTreeDynMax) { set_error("tree inertia/coriolis/accel: the robot needs more LDS than a CU has"); return RTBHIP_ELIMIT; }
        if (l <= 160 * 1024) {
            const int64_t tl = (N + tq_.tile - 1) / tq_.tile;
            void *args[] = {&tq_, &groups, &q, &qd, &tq, &out};
            const int rc = jit_launch(f, dim3((unsigned)tl), dim3(kWave), l, s, args);
            if (rc != RTBHIP_OK) return rc;
            note_launch((int)tl, kWave, (int)l);
            return RTBHIP_OK;
        }
    }
    if (sig == kTreeSigUR && t->n == 6) {       // (a signature does not encode the group count: a trailing General / t = 0 group reads as "absent")
        e = launch_tree_dyn_ng<6, kTreeSigUR>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds);
    } else if (sig == kTreeSigIbx8 && topo == kTreeTopoIbx8 && t->n == 8) {
        e = launch_tree_dyn_ng<8, kTreeSigIbx8, kTreeTopoIbx8>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds);
    } else if (sig == kTreeSigPx100 && topo == kTreeTopoPx100 && t->n == 7) {
        e = launch_tree_dyn_ng<7, kTreeSigPx100, kTreeTopoPx100>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds);
    } else if (sig == kTreeSigIbx9 && sig2 == kTreeSig2Ibx9 && topo == kTreeTopoIbx9 && t->n == 9) {
        e = launch_tree_dyn_ng<9, kTreeSigIbx9, kTreeTopoIbx9, kTreeSig2Ibx9>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds);
    } else if (sig == kTreeSigFetch && sig2 == kTreeSig2Fetch && topo == kTreeTopoFetch && t->n == 10) {
        e = launch_tree_dyn_ng<10, kTreeSigFetch, kTreeTopoFetch, kTreeSig2Fetch>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds);
    } else if (sig == kTreeSigMico && sig2 == kTreeSig2Mico && topo == kTreeTopoMico && t->n == 10) {
        e = launch_tree_dyn_ng<10, kTreeSigMico, kTreeTopoMico, kTreeSig2Mico>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds);
    } else
#ifdef RTB_TREE_DEV_NG
    e = launch_tree_dyn_ng<RTB_TREE_DEV_NG>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain);
#else
    switch (t->n) {
    case 1: e = launch_tree_dyn_ng<1>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 2: e = launch_tree_dyn_ng<2>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 3: e = launch_tree_dyn_ng<3>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 4: e = launch_tree_dyn_ng<4>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 5: e = launch_tree_dyn_ng<5>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 6: e = launch_tree_dyn_ng<6>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 7: e = launch_tree_dyn_ng<7>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 8: e = launch_tree_dyn_ng<8>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 9: e = launch_tree_dyn_ng<9>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 10: e = launch_tree_dyn_ng<10>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 11: e = launch_tree_dyn_ng<11>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 12: e = launch_tree_dyn_ng<12>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 13: e = launch_tree_dyn_ng<13>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 14: e = launch_tree_dyn_ng<14>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 15: e = launch_tree_dyn_ng<15>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 16: e = launch_tree_dyn_ng<16>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 17: e = launch_tree_dyn_ng<17>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 18: e = launch_tree_dyn_ng<18>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    case 19: e = launch_tree_dyn_ng<19>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    default: e = launch_tree_dyn_ng<20>(mode, grid, s, t->nslots, tp, groups, q, qd, tq, out, &lds, plain); break;
    }
#endif
    if (lds > 160 * 1024) { set_error("tree inertia/coriolis/accel: the robot needs more LDS than a CU has"); return RTBHIP_ELIMIT; }
    if (e != hipSuccess) return hip_fail(e, "k_tree_dyn launch");
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_tree_dyn launch");
    return RTBHIP_OK;
}

#endif  // RTB_HOST_SIDE

}  // namespace rtbhip
