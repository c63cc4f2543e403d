// dyn_kernels.hip -- gfx950 kernels for the dynamics terms the reference derives from repeated
// Newton-Euler calls (robot/Dynamics.py): the joint-space inertia matrix M(q) (`inertia`, :704-763:
// n RNE calls per configuration), the Coriolis/centripetal matrix C(q, qd) (`coriolis`, :765-861:
// n + n(n-1)/2 RNE calls) and the forward dynamics qdd = M^-1 (tau - tau_0) (`accel`, :424-509: n + 1
// RNE calls and a dense solve).  In the reference every one of those RNE calls is a Python -> C
// round trip (DHRobot.rne -> frne.frne); here one lane owns one configuration and runs all the
// passes back to back on the rne_lane recursion of rne_device.h, with the pass index a wave-uniform
// loop counter, the unit vectors generated on the fly (no (n,n) input blocks), the partial results in
// the wave's LDS tile, and the (N,n,n) / (N,n) outputs written as contiguous runs.
// Bound: fp64 VALU issue (~1.5 kflop x (n .. n(n+1)/2 + 1) passes against 8n..16n bytes in and
// 8n^2 bytes out per configuration).
#include "dyn_device.h"
#include "kin_tile.h"

namespace rtbhip {

typedef const __attribute__((address_space(4))) DevLink *ConstLinksD;
constexpr int kDW = 64;
constexpr int kDynMaxJoints = 16;

struct DynParams {
    int32_t n, mode;
    int64_t N;
    double grav[3];
    int32_t tile, pad_;      // configurations per single-wave workgroup: 64, or fewer where the (n, n) tiles of 64 lanes exceed a CU's LDS (n > 16)
};


// K input arrays of (N, NJ) each -> lane-major LDS rows [k*NJ + j], all loads in flight at once
template <int NJ, int K>
__device__ __forceinline__ void dyn_load(double *lds, int stride, const double *const (&src)[K], int64_t cfg0, int count, int lane)
{
    double r[K][NJ];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int f = lane + kDW * i;
            r[k][i] = (f < count) ? src[k][cfg0 * NJ + f] : 0.0;
        }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int f = lane + kDW * i;
            const int row = f / NJ, c = f - row * NJ;
            lds[row * stride + k * NJ + c] = r[k][i];
        }
}

// LDS per wave (doubles): inputs 64 x (K*NJ | 1), then the n x n work / output tiles
template <int NJ, int MODE, bool ALLREV = false, bool MDH = false>
struct DynLayout {
    // inertia of an all-revolute chain reads q only for the sines / cosines, before the first pass writes its row: the
    // input row then lives in the output tile itself (28.7 -> 25.1 KB per wave for n = 7: 5 -> 6 waves per CU)
    static constexpr bool alias_in = MODE == kDynInertia && ALLREV;
    // coriolis of an all-revolute chain: the same for q; only the qd row keeps a place of its own (4 -> 5 waves per CU)
    static constexpr bool alias_q = (MODE == kDynCoriolis || MODE == kDynAccel) && ALLREV;
    static constexpr int K = MODE == kDynInertia ? 1 : (MODE == kDynCoriolis ? 2 : 3);
    // accel of an all-revolute chain: q, qd and the torque row all live in the (packed) M tile until the first pass -- the full one,
    // which leaves torque - tau_0 in registers -- has read them (dyn_device.h); needs 3 n <= n (n + 1) / 2, i.e. n >= 5
    static constexpr bool alias_all = (MODE == kDynAccel && ALLREV && 3 * NJ <= NJ * (NJ + 1) / 2) ||
                                      (MODE == kDynCoriolis && ALLREV && NJ >= 2);       // coriolis: q | qd in the n x n tile, qd copied to registers first
    static constexpr int in_stride = (((MODE == kDynCoriolis || MODE == kDynAccel) && ALLREV ? K - 1 : K) * NJ) | 1;
    // accel: packed lower triangle of M; inertia of an all-revolute chain too (its columns come from the mirrored
    // acceleration-only passes of rne_device.h, so the tile holds 28 instead of 49 doubles per lane for n = 7 -- 6 -> 10 waves per
    // CU -- and the flush expands it to the full matrix)
    static constexpr bool packed = (MODE == kDynAccel && !kDynFullTile<MDH, ALLREV>) || (MODE == kDynInertia && ALLREV);
    static constexpr int W = packed ? (NJ * (NJ + 1) / 2 > NJ ? NJ * (NJ + 1) / 2 : NJ) : NJ * NJ;
    static constexpr int w_stride = W | 1;
    static constexpr int tiles = 1;                                // coriolis too: Csq is folded into C as it is produced (dyn_device.h)
    static constexpr int doubles = kDW * ((alias_in || alias_all ? 0 : in_stride) + tiles * w_stride);
};

// packed lower triangles (row r, column c <= r at r (r + 1) / 2 + c) of ncfg lanes -> the full symmetric (n, n) matrices as one
// contiguous run of 16-byte non-temporal stores
template <int NJ>
__device__ __forceinline__ void flush_symmetric(const double *rows, int stride, int ncfg, double *__restrict__ dst, int lane)
{
    constexpr int W = NJ * NJ;
    const int total = ncfg * W;
    auto at = [&](int f) {
        const int cfg = f / W, rem = f - cfg * W, r = rem / NJ, c = rem - r * NJ;
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        return rows[cfg * stride + hi * (hi + 1) / 2 + lo];
    };
    for (int f = 2 * lane; f < total; f += 2 * kDW) {
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

template <int NJ, bool MDH, int MODE, bool ALLREV, RneSig SIG = 0>
__global__ __launch_bounds__(kDW, (NJ <= 8 ? 2 : 1)) void k_dyn(DynParams dp, const DevLink *links_g, const double *__restrict__ q,
                                                const double *__restrict__ qd, const double *__restrict__ tq,
                                                double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    typedef DynLayout<NJ, MODE, ALLREV, MDH> L;
    ConstLinksD links = (ConstLinksD)links_g;
    const int lane = threadIdx.x;
    const int T = dp.tile;
    const int64_t cfg0 = (int64_t)blockIdx.x * T;
    const int64_t left = dp.N - cfg0;
    const int ncfg = left < T ? (int)left : T;
    const int count = ncfg * NJ;
    double *A = lds + (L::alias_in || L::alias_all ? 0 : T * L::in_stride);   // n x n tile: M (inertia, accel) or C (coriolis)
    double *in = (L::alias_in || L::alias_all) ? A : lds;
    constexpr int in_stride = (L::alias_in || L::alias_all) ? L::w_stride : L::in_stride;
    if (MODE == kDynInertia) { const double *const src[1] = {q}; dyn_load<NJ, 1>(in, in_stride, src, cfg0, count, lane); }
    if (MODE == kDynCoriolis) {
        if (L::alias_all) {
            const double *const src[2] = {q, qd};
            dyn_load<NJ, 2>(A, L::w_stride, src, cfg0, count, lane);
        } else if (L::alias_q) {
            const double *const s0[1] = {q}, *const s1[1] = {qd};
            dyn_load<NJ, 1>(A, L::w_stride, s0, cfg0, count, lane);
            dyn_load<NJ, 1>(in, in_stride, s1, cfg0, count, lane);
        } else {
            const double *const src[2] = {q, qd};
            dyn_load<NJ, 2>(in, in_stride, src, cfg0, count, lane);
        }
    }
    if (MODE == kDynAccel) {
        if (L::alias_all) {
            const double *const src[3] = {q, qd, tq};
            dyn_load<NJ, 3>(A, L::w_stride, src, cfg0, count, lane);
        } else if (L::alias_q) {
            const double *const s0[1] = {q}, *const s1[2] = {qd, tq};
            dyn_load<NJ, 1>(A, L::w_stride, s0, cfg0, count, lane);
            dyn_load<NJ, 2>(in, in_stride, s1, cfg0, count, lane);
        } else {
            const double *const src[3] = {q, qd, tq};
            dyn_load<NJ, 3>(in, in_stride, src, cfg0, count, lane);
        }
    }
    __syncthreads();
    if (lane < ncfg) {
        // alias_q: the row holds qd only, at the offset dyn_lane expects it (mine[n + j])
        const double *mine = in + lane * in_stride - ((L::alias_q && !L::alias_all) ? NJ : 0);
        dyn_lane<NJ, MDH, MODE, ALLREV, SIG>(links, mine, A + lane * L::w_stride, v3(dp.grav[0], dp.grav[1], dp.grav[2]),
                                        (L::alias_q && !L::alias_all) ? A + lane * L::w_stride : nullptr);
    }
    __syncthreads();
    if (MODE == kDynAccel) flush_run(A, L::w_stride, NJ, ncfg, out + cfg0 * NJ, lane);
    else if (MODE == kDynInertia && L::packed) flush_symmetric<NJ>(A, L::w_stride, ncfg, out + cfg0 * (NJ * NJ), lane);
    else flush_run(A, L::w_stride, L::W, ncfg, out + cfg0 * L::W, lane);
}

#if RTB_HOST_SIDE      // the launchers (the kernel above is also what jit.cpp hands to hipRTC, one instantiation at a time)
template <int NJ, int MODE, bool MDH, bool ALLREV, RneSig SIG = 0>
static hipError_t launch_one(dim3 grid, hipStream_t s, size_t lds, const DynParams &dp, const DevLink *links, const double *q,
                             const double *qd, const double *tq, double *out)
{
    auto k = k_dyn<NJ, MDH, MODE, ALLREV, SIG>;
    if (lds > 48 * 1024) { hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k, grid, dim3(kDW), lds, s, dp, links, q, qd, tq, out);
    return hipGetLastError();
}

int rne_sig_enabled();      // rne_kernels.hip: rtbhip_tune("rne_sig")
std::string rne_jit_expr(int n, bool mdh, RneSig sig, int variant);
template <int NJ, int MODE>
static hipError_t launch_mode(const Dyn *d, bool mdh, bool allrev, dim3 grid, hipStream_t s, const DynParams &dp, const DevLink *links, const double *q,
                              const double *qd, const double *tq, double *out, size_t *lds_out, RneSig sig)
{
    const size_t lds = (size_t)(allrev ? DynLayout<NJ, MODE, true>::doubles
                                       : (mdh ? DynLayout<NJ, MODE, false, true>::doubles : DynLayout<NJ, MODE, false, false>::doubles)) * sizeof(double);
    *lds_out = lds;
    // a robot whose link table has a structure signature this build is instantiated for (rne_device.h: kRneSig*)
    if constexpr (NJ == 7) { if (sig == kRneSigPanda && mdh && allrev && jit_builtin_enabled()) return launch_one<7, MODE, true, true, kRneSigPanda>(grid, s, lds, dp, links, q, qd, tq, out); }
    if constexpr (NJ == 6) { if (sig == kRneSigPuma560 && !mdh && allrev && jit_builtin_enabled()) return launch_one<6, MODE, false, true, kRneSigPuma560>(grid, s, lds, dp, links, q, qd, tq, out); }
    if constexpr (NJ <= kRneSigMaxLinks) {
        // any other robot with a signature: its own instantiation, compiled at run time (jit.cpp); the general kernels below serve until it is there
        const bool builtin = jit_builtin_enabled() && ((NJ == 7 && mdh && sig == kRneSigPanda) || (NJ == 6 && !mdh && sig == kRneSigPuma560));
        if (sig && allrev && !builtin && jit_enabled()) {
            if (hipFunction_t f = d->jit.get("dyn_kernels.hip", 2 + MODE, [&] { return rne_jit_expr(NJ, mdh, sig, 2 + MODE); })) {
                DynParams dpv = dp;
                void *args[] = {&dpv, &links, &q, &qd, &tq, &out};
                return jit_launch(f, grid, dim3(kDW), lds, s, args) == RTBHIP_OK ? hipSuccess : hipErrorLaunchFailure;
            }
        }
    }
    if (mdh) return allrev ? launch_one<NJ, MODE, true, true>(grid, s, lds, dp, links, q, qd, tq, out)
                           : launch_one<NJ, MODE, true, false>(grid, s, lds, dp, links, q, qd, tq, out);
    return allrev ? launch_one<NJ, MODE, false, true>(grid, s, lds, dp, links, q, qd, tq, out)
                  : launch_one<NJ, MODE, false, false>(grid, s, lds, dp, links, q, qd, tq, out);
}

template <int NJ>
static hipError_t launch_nj(const Dyn *d, int mode, bool mdh, bool allrev, dim3 grid, hipStream_t s, const DynParams &dp, const DevLink *links,
                            const double *q, const double *qd, const double *tq, double *out, size_t *lds, RneSig sig = 0)
{
    if (mode == kDynInertia) return launch_mode<NJ, kDynInertia>(d, mdh, allrev, grid, s, dp, links, q, qd, tq, out, lds, sig);
    if (mode == kDynCoriolis) return launch_mode<NJ, kDynCoriolis>(d, mdh, allrev, grid, s, dp, links, q, qd, tq, out, lds, sig);
    return launch_mode<NJ, kDynAccel>(d, mdh, allrev, grid, s, dp, links, q, qd, tq, out, lds, sig);
}

// Chains of 17 .. RTBHIP_MAX_JOINTS joints: no built-in instantiation -- the same k_dyn template instantiated at run time (jit.cpp; the caller waits:
// seconds to a minute on first use, a file read afterwards), on tiles of fewer than 64 configurations where 64 (n, n) tiles do not fit a CU's
// LDS.  Served, not fast; the reference's loops take any n (robot/Dynamics.py:704-861, 424-509).
static int launch_dyn_runtime_size(const Dyn *d, const DevLink *links, int mode, const double *q, const double *qd, const double *tq, int64_t N,
                                   const double *grav3, double *out, hipStream_t s)
{
    const int n = d->n;
    const bool mdh = d->mdh != 0;
    bool allrev = true;
    for (const DevLink &l : d->links) allrev = allrev && l.sigma == 0;
    // DynLayout's arithmetic for a run-time n (the kernel has it at compile time)
    const int K = mode == kDynInertia ? 1 : (mode == kDynCoriolis ? 2 : 3);
    const bool alias_in = mode == kDynInertia && allrev;
    const bool alias_all = (mode == kDynAccel && allrev && 3 * n <= n * (n + 1) / 2) || (mode == kDynCoriolis && allrev && n >= 2);
    const int in_stride = (((mode == kDynCoriolis || mode == kDynAccel) && allrev ? K - 1 : K) * n) | 1;
    const bool full_tile = mdh && !allrev;                                                  // kDynFullTile<MDH, ALLREV> (dyn_device.h)
    const bool packed = (mode == kDynAccel && !full_tile) || (mode == kDynInertia && allrev);
    const int W = packed ? (n * (n + 1) / 2 > n ? n * (n + 1) / 2 : n) : n * n;
    const size_t per_lane = (size_t)((alias_in || alias_all ? 0 : in_stride) + (W | 1)) * sizeof(double);
    int tile = kDW;
    while (tile > 8 && per_lane * tile > 160 * 1024) tile /= 2;
    const size_t lds = per_lane * tile;
    if (lds > 160 * 1024) { set_error("inertia/coriolis/accel: the chain needs more LDS than a CU has"); return RTBHIP_ELIMIT; }
    const int64_t tiles = (N + tile - 1) / tile;
    if (tiles > 0x7fffffff) { set_error("inertia/coriolis/accel: batch too large for one launch"); return RTBHIP_ELIMIT; }
    DynParams dp;
    dp.n = n; dp.mode = mode; dp.N = N; dp.tile = tile; dp.pad_ = 0;
    for (int i = 0; i < 3; i++) dp.grav[i] = grav3 ? grav3[i] : 0.0;
    hipFunction_t f = d->jit.get_wait("dyn_kernels.hip", 8 + mode, [&] {
        return "rtbhip::k_dyn<" + std::to_string(n) + ", " + (mdh ? "true" : "false") + ", " + std::to_string(mode) + ", " + (allrev ? "true" : "false") + ", 0>"; });
    if (!f) return RTBHIP_ELIMIT;
    void *args[] = {(void *)&dp, (void *)&links, (void *)&q, (void *)&qd, (void *)&tq, (void *)&out};
    const int rc = jit_launch(f, dim3((unsigned)tiles), dim3(kDW), lds, s, args);
    if (rc != RTBHIP_OK) return rc;
    note_launch((int)tiles, kDW, (int)lds);
    return RTBHIP_OK;
}

int launch_dyn(const Dyn *d, const DevLink *links, int mode, const double *q, const double *qd, const double *tq,
               int64_t N, const double *grav3, double *out, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (d->n > RTBHIP_MAX_JOINTS) { set_error("inertia/coriolis/accel: more than RTBHIP_MAX_JOINTS joints"); return RTBHIP_ELIMIT; }
    if (d->n > kDynMaxJoints) return launch_dyn_runtime_size(d, links, mode, q, qd, tq, N, grav3, out, s);
    const int64_t tiles = (N + kDW - 1) / kDW;
    if (tiles > 0x7fffffff) { set_error("inertia/coriolis/accel: batch too large for one launch"); return RTBHIP_ELIMIT; }
    DynParams dp;
    dp.n = d->n; dp.mode = mode; dp.N = N; dp.tile = kDW; dp.pad_ = 0;
    for (int i = 0; i < 3; i++) dp.grav[i] = grav3 ? grav3[i] : 0.0;
    dim3 grid((unsigned)tiles);
    const bool mdh = d->mdh != 0;
    bool allrev = true;
    for (const DevLink &l : d->links) allrev = allrev && l.sigma == 0;
    hipError_t e = hipSuccess;
    size_t lds = 0;
    const RneSig sig = (rne_sig_enabled() && allrev) ? rne_signature(d->links.data(), d->n) : 0;
    switch (d->n) {
    case 1: e = launch_nj<1>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 2: e = launch_nj<2>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 3: e = launch_nj<3>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 4: e = launch_nj<4>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 5: e = launch_nj<5>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 6: e = launch_nj<6>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 7: e = launch_nj<7>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 8: e = launch_nj<8>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds, sig); break;
    case 9: e = launch_nj<9>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;   // 9, 10: one wave per SIMD
    case 10: e = launch_nj<10>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;
    // 11..16: the per-link state of the recursion no longer fits the register file (scratch) and the (n,n) tile of a wave
    // takes most of a CU's LDS -- one wave per CU; served, not fast
    case 11: e = launch_nj<11>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;
    case 12: e = launch_nj<12>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;
    case 13: e = launch_nj<13>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;
    case 14: e = launch_nj<14>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;
    case 15: e = launch_nj<15>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;
    default: e = launch_nj<16>(d, mode, mdh, allrev, grid, s, dp, links, q, qd, tq, out, &lds); break;
    }
    if (lds > 160 * 1024) { set_error("inertia/coriolis/accel: the chain needs more LDS than a CU has"); return RTBHIP_ELIMIT; }
    note_launch((int)grid.x, kDW, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "k_dyn launch");
    return RTBHIP_OK;
}

#endif  // RTB_HOST_SIDE

}  // namespace rtbhip
