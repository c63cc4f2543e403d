// partial_kernels.hip -- gfx950 kernel for ETS.partial_fkine0 (robot/ETS.py:1821-2013): the order-c
// tensor of the forward kinematics' partial derivatives from the lower-order ones.  The output
// (N, n^(c-1), 6, n) is a flat array of (6, n) blocks, block index = global column / n.  One lane owns one
// 6-vector column; a workgroup of 256 lanes owns floor(256 / n) consecutive blocks, assembles them in LDS and
// writes them as one contiguous run of 16-byte stores (lane-per-column stores would touch each 128-byte line
// six times with 8n useful bytes).  The Jacobians and Hessians of the one or two configurations a workgroup
// touches are copied to LDS first: every column reads 9 x 2^(c-2) of their elements at lane-scattered addresses,
// and as global loads those share the texture-address path with the stores (measured: 0.34 ms of loads +
// 0.30 ms of stores = 0.64 ms per 1e5 Panda configurations at order 3, not overlapped).  Bound: HBM writes, 48 n^c bytes per configuration (16.5 KB at order 3,
// 115 KB at order 4 for the Panda) against 2^(c-2) x 18 flops per column.
#include "partial_device.h"
#include "rtbhip_internal.h"
#include <string>

namespace rtbhip {

struct PartialSrc {
    const double *p[kPartialMaxOrder];
};

constexpr int kPartialStage = 2560;                                  // most doubles of LDS spent on staged Jacobians + Hessians

constexpr int kPartialBlock = 256;
#ifndef RTB_PARTIAL_U
#define RTB_PARTIAL_U 2
#endif
constexpr int kPartialU = RTB_PARTIAL_U;      // columns per lane: the kernel is bound by the bytes its resident workgroups keep in flight

// (6, n) blocks per workgroup: as many as 256 lanes hold, rounded down so that a workgroup's run of 48 n bytes per
// block is a whole number of 128-byte lines (neighbouring workgroups -- possibly on different XCDs, i.e. different
// L2s -- then never share a line)
__host__ __device__ inline int partial_blocks_per_group(int n)
{
    int per = kPartialBlock / n, m = 8;
    for (int g = 3 * n; m > 1 && g % m != 0;) m >>= 1;              // m = gcd(3 n, 8)
    m = 8 / m;
    return per >= m ? per - per % m : per;
}

// LDS: [tile: per x 6n doubles][stage: stage_cfgs x (6n + 6n^2) doubles], sized by the launcher -- the kernel is bound by
// the bytes its resident workgroups keep in flight, so every KB of LDS not asked for is occupancy
template <int C>
__global__ __launch_bounds__(kPartialBlock) void k_partial(PartialPlan plan, PartialSrc src, int stage_cfgs, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = plan.n, tid = threadIdx.x;
    const int per = partial_blocks_per_group(n);
    double *tile = lds, *stage = lds + ((per * kPartialU * 6 * n + 1) & ~1);
    const uint32_t bpc = (uint32_t)plan.cols / (uint32_t)n;           // blocks per configuration
    const int perU = per * kPartialU;
    const int64_t b0 = (int64_t)blockIdx.x * perU;
    const int64_t left = plan.N * bpc - b0;
    const int nb = left < perU ? (int)left : perU;
    const int64_t cfg0 = b0 / bpc;                                   // wave-uniform: scalar unit
    const uint32_t lb0 = (uint32_t)(b0 - cfg0 * bpc);
    // orders 1 and 2 of the K configurations this workgroup touches -> LDS: [K Jacobians][K Hessians]
    const int K = (int)((lb0 + nb - 1) / bpc) + 1;
    const int szj = 6 * n, szh = 6 * n * n;
    const bool staged = K <= stage_cfgs;
    if (staged) {
        const double *J = src.p[0] + cfg0 * szj, *H = src.p[1] + cfg0 * szh;
        // [K Jacobians][K Hessians] are two contiguous source runs; all loads in flight before the first LDS write
        constexpr int kIt = (kPartialStage + kPartialBlock - 1) / kPartialBlock;
        double r[kIt];
        const int tj = K * szj, tot = K * (szj + szh);
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int i = tid + u * kPartialBlock;
            r[u] = i < tj ? J[i] : (i < tot ? H[i - tj] : 0.0);
        }
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int i = tid + u * kPartialBlock;
            if (i < tot) stage[i] = r[u];
        }
        __syncthreads();
    }
    typedef const __attribute__((address_space(3))) double *LdsPtr;     // explicit LDS loads (ds_read), never flat
    const LdsPtr sj = (LdsPtr)stage, sh = (LdsPtr)stage + K * szj;
    uint32_t j, lbr;
    const int bl0 = (int)divmod24((uint32_t)tid, (uint32_t)n, 1.0f / (float)n, &j);
#pragma unroll
    for (int u = 0; u < kPartialU; ++u) {
        const int bl = bl0 + u * per;
        if (bl0 >= per || bl >= nb) continue;
        const uint32_t dc = divmod24(lb0 + bl, bpc, 1.0f / (float)bpc, &lbr);     // this workgroup may straddle configurations
        double *t = tile + mad24((uint32_t)bl, 6u * n, j);
        const int64_t cfg = cfg0 + dc;
        const uint32_t col = mad24(lbr, (uint32_t)n, j);
        auto put = [&](int r, double v) { t[r * n] = v; };
        if (staged) {
            const LdsPtr cj = sj + mad24(dc, (uint32_t)szj, 0), ch = sh + mad24(dc, (uint32_t)szh, 0);
            partial_column<C>(plan, [&](int order, int64_t c, int off) -> double {
                if (order == 1) return cj[off];
                if (order == 2) return ch[off];
                return src.p[order - 1][c * plan.size[order] + off];
            }, cfg, col, put);
        } else {
            partial_column<C>(plan, [&](int order, int64_t c, int off) -> double { return src.p[order - 1][c * plan.size[order] + off]; },
                              cfg, col, put);
        }
    }
    __syncthreads();
    const int pairs = nb * 3 * n;                                    // 6n doubles per block, two per store
    double *dst = out + b0 * 6 * n;                                  // 48 n bytes per block: 16-byte aligned
    for (int i = tid; i < pairs; i += kPartialBlock) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d w = *reinterpret_cast<const v2d *>(tile + 2 * i);
        __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + 2 * i));          // global_store_dwordx4 ... nt
    }
}

// Order 3 -- the (N, n, n, 6, n) tensor, what ETS.partial_fkine0(q, n=3) returns -- on workgroups that own WHOLE configurations:
// G of them each (partial3_geometry), their Jacobians and Hessians staged once (the general kernel's runs of floor(256/n) blocks
// straddle configurations and re-stage them: 1.8x the input bytes, profiles/r01_t_pmc_secondary.txt), the G x n^3 columns dealt to
// the lanes U at a time, the output assembled in LDS and written as ONE contiguous run of 16-byte non-temporal stores.
template <int U>
__global__ __launch_bounds__(kPartialBlock) void k_partial3(int n, int G, int64_t N, const double *__restrict__ J, const double *__restrict__ H,
                                                            double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const uint32_t nn = (uint32_t)(n * n), cols = nn * (uint32_t)n;
    const int szj = 6 * n, szh = 6 * n * n;
    const int64_t cfg0 = (int64_t)blockIdx.x * G;
    const int g = (int)(N - cfg0 < G ? N - cfg0 : G);
    double *tile = lds, *stage = lds + ((G * 6 * (int)cols + 1) & ~1);          // [g output runs][g Jacobians][g Hessians]
    if (H == nullptr) {
        // no Hessian tensor supplied (rtbhip_partial_fkine0 at order 3): stage the Jacobians and form the Hessians from them right here --
        // H[j, row, i] = hessian_entry (kin_device.h: the expression the Hessian kernel evaluates, so the same bits) -- instead of having a
        // launch write 2352 B per configuration that this one reads back: the call's second launch and 470 MB of traffic per 1e5 Panda
        // configurations go away
        const double *Js = J + cfg0 * szj;
        const int tj = g * szj;
        for (int i = tid; i < tj; i += kPartialBlock) stage[i] = Js[i];
        __syncthreads();
        const int th = g * szh;
        for (int i = tid; i < th; i += kPartialBlock) {
            const int c = i / szh, w = i - c * szh;
            const int j = w / szj, w2 = w - j * szj;
            const int row = w2 / n, col = w2 - row * n;
            stage[tj + i] = hessian_entry(stage + c * szj, n, j, row, col);
        }
        __syncthreads();
    } else {
        const double *Js = J + cfg0 * szj, *Hs = H + cfg0 * szh;
        const int tj = g * szj, tot = g * (szj + szh);
        for (int i0 = 0; i0 < tot; i0 += 4 * kPartialBlock) {                   // four loads in flight per lane and round
            double r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + tid + u * kPartialBlock;
                r[u] = i < tj ? Js[i] : (i < tot ? Hs[i - tj] : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + tid + u * kPartialBlock;
                if (i < tot) stage[i] = r[u];
            }
        }
        __syncthreads();
    }
    typedef const __attribute__((address_space(3))) double *LdsPtr;
    const LdsPtr sj = (LdsPtr)stage, sh = (LdsPtr)stage + g * szj;
    const float inv_n = 1.0f / (float)n, inv_cols = 1.0f / (float)cols;
    const uint32_t total = (uint32_t)g * cols;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t idx = (uint32_t)tid + (uint32_t)u * kPartialBlock;
        if (idx >= total) continue;
        uint32_t col, d0, d1, d2;
        const uint32_t dc = divmod24(idx, cols, inv_cols, &col);
        uint32_t r = divmod24(col, (uint32_t)n, inv_n, &d0);
        d2 = divmod24(r, (uint32_t)n, inv_n, &d1);
        const LdsPtr cj = sj + mad24(dc, (uint32_t)szj, 0), ch = sh + mad24(dc, (uint32_t)szh, 0);
        double *t = tile + mad24(dc, 6u * cols, mad24(r, 6u * (uint32_t)n, d0));       // block r = col / n of configuration dc, column d0
        partial3_column(n, [&](uint32_t off) -> double { return cj[off]; }, [&](uint32_t off) -> double { return ch[off]; }, d0, d1, d2,
                        [&](int row, double v) { t[row * n] = v; });
    }
    __syncthreads();
    const int pairs = g * 3 * (int)cols;
    double *dst = out + cfg0 * 6 * (int64_t)cols;
    for (int i = tid; i < pairs; i += kPartialBlock) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d w = *reinterpret_cast<const v2d *>(tile + 2 * i);
        __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + 2 * i));
    }
}

static int g_partial3 = 1;        // rtbhip_tune("partial3", 0): the general kernel at order 3 too (A/B)
static int g_partial3_fused = 1;  // rtbhip_tune("partial3_fused", 0): feed k_partial3 a Hessian tensor from its own launch, as before (A/B)
void partial_tune(const char *key, int value)
{
    if (std::string(key) == "partial3") g_partial3 = value != 0;
    if (std::string(key) == "partial3_fused") g_partial3_fused = value != 0;
}

// order 3 of an n-joint chain goes to k_partial3, which needs no Hessian tensor (it forms the Hessians from the Jacobians it stages)
bool partial3_needs_no_hessian(int n)
{
    int G = 0, U = 0;
    partial3_geometry(n, kPartialBlock, &G, &U);
    return g_partial3 && g_partial3_fused && G > 0 && (int64_t)6 * n * n * n * G < (1 << 24);
}

// lower[a-1] = order-a tensor (device), a = 1 .. order-1; out = order-`order` tensor
int launch_partial(int n, int order, const double *const *lower, int64_t N, double *out, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (order == 3 && g_partial3) {
        int G = 0, U = 0;
        partial3_geometry(n, kPartialBlock, &G, &U);
        if (G > 0 && (int64_t)6 * n * n * n * G < (1 << 24)) {
            const int64_t blocks = (N + G - 1) / G;
            if (blocks > 0x7fffffff) { set_error("partial_fkine0: batch too large for one launch"); return RTBHIP_ELIMIT; }
            const size_t lds = (size_t)(((G * 6 * n * n * n + 1) & ~1) + G * (6 * n + 6 * n * n)) * sizeof(double);
            const dim3 grid((unsigned)blocks), block(kPartialBlock);
            switch (U) {
            case 1: hipLaunchKernelGGL(k_partial3<1>, grid, block, lds, s, n, G, N, lower[0], lower[1], out); break;
            case 2: hipLaunchKernelGGL(k_partial3<2>, grid, block, lds, s, n, G, N, lower[0], lower[1], out); break;
            case 3: hipLaunchKernelGGL(k_partial3<3>, grid, block, lds, s, n, G, N, lower[0], lower[1], out); break;
            default: hipLaunchKernelGGL(k_partial3<4>, grid, block, lds, s, n, G, N, lower[0], lower[1], out); break;
            }
            note_launch((int)blocks, kPartialBlock, (int)lds);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "k_partial3 launch");
            return RTBHIP_OK;
        }
    }
    PartialPlan plan;
    partial_plan(n, order, &plan);
    plan.N = N;
    PartialSrc src;
    for (int a = 0; a < kPartialMaxOrder; ++a) src.p[a] = a < order - 1 ? lower[a] : nullptr;
    const int per = partial_blocks_per_group(n), perU = per * kPartialU;
    const int64_t blocks = (N * (int64_t)(plan.cols / n) + perU - 1) / perU;
    if (plan.size[order - 1] >= (1 << 24) || plan.cols >= (1 << 24)) { set_error("partial_fkine0: tensor too large (n^order must stay below 2^24)"); return RTBHIP_ELIMIT; }
    if (blocks > 0x7fffffff) { set_error("partial_fkine0: batch too large for one launch"); return RTBHIP_ELIMIT; }
    const dim3 grid((unsigned)blocks), block(kPartialBlock);
    const int64_t bpc = plan.cols / n;
    int stage_cfgs = (int)((perU + bpc - 2) / bpc) + 1;                // most configurations one workgroup can touch
    if ((int64_t)stage_cfgs * (6 * n + 6 * n * n) > kPartialStage) stage_cfgs = 0;
    const size_t lds = (size_t)(((per * kPartialU * 6 * n + 1) & ~1) + stage_cfgs * (6 * n + 6 * n * n)) * sizeof(double);
    switch (order) {
    case 3: hipLaunchKernelGGL(k_partial<3>, grid, block, lds, s, plan, src, stage_cfgs, out); break;
    case 4: hipLaunchKernelGGL(k_partial<4>, grid, block, lds, s, plan, src, stage_cfgs, out); break;
    case 5: hipLaunchKernelGGL(k_partial<5>, grid, block, lds, s, plan, src, stage_cfgs, out); break;
    default: hipLaunchKernelGGL(k_partial<6>, grid, block, lds, s, plan, src, stage_cfgs, out); break;
    }
    note_launch((int)blocks, kPartialBlock, (int)lds);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_partial launch");
    return RTBHIP_OK;
}

}  // namespace rtbhip
