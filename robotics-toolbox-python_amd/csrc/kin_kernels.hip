// kin_kernels.hip -- gfx950 kernels for batched fkine / jacob0 / jacobe / hessian (headline path).
//
// Replaces the hot loops of core/fknm.cpp:1038-1052 (ETS_fkine trajectory loop) and the Python
// per-row loops around ETS_jacob0/ETS_jacobe/ETS_hessian0 (fknm.cpp:583-921).
//
// Launch shape: one wavefront (64 lanes) per workgroup, one configuration per lane, grid-stride
// over 64-configuration tiles.  Single-wave workgroups make every LDS hand-off wave-private (the
// s_barrier of a one-wave group is free) and let the dispatcher balance tiles across the 256 CUs.
// The chain program is read through the constant address space => s_load into SGPRs.
// Roofline: HBM-bound by construction -- 8*qw bytes in, 128 + 48n bytes out per configuration
// (520 B for the Panda), ~0.6 kflop of fp64 VALU + n sincos per configuration.
#include "kin_reg.h"
#include "diff_device.h"
#include "servo_device.h"
#include "diff_kernel.h"
#include <algorithm>
#include <cstring>

namespace rtbhip {

// Store policy of the fused fkine + Jacobian kernels.  Both output arrays are streaming writes; written non-temporally, the pair's speed depends on
// WHERE the allocator put T and J relative to one another -- 78 or 90 us per 1e6 Panda configurations, fixed for a given pair of buffers, about
// two pairs in three slow (round 4: profiles/r04_headline_stores.txt).  When the pose array (128 B per configuration) is small enough to stay in
// the part's 256 MB memory-side cache, writing IT with ordinary stores (and first) removes the dependence: a loop that writes the same output
// arrays every step -- bench.py, a control loop with preallocated outputs -- then runs at 78 us on every pair, because the pose array is
// absorbed by that cache and rewritten in place (with three or more output sets used in rotation both forms take 90 us).  Beyond that size
// ordinary stores lose (2e6 configurations: 188-200 us against 148-171), so the launcher turns them on (KinParams.pad bit 0) only for a
// pose array of at most kPoseCacheBytes; everything else -- fkine alone, the fleet, long batches -- streams non-temporally as before.
// RTB_T_PLAIN_MAX_BYTES = 0 restores the form of rounds 1-3 (A/B).
#ifndef RTB_T_FIRST
#define RTB_T_FIRST 1
#endif
#ifndef RTB_T_PLAIN_MAX_BYTES
#define RTB_T_PLAIN_MAX_BYTES (136ll << 20)
#endif
constexpr long long kPoseCacheBytes = RTB_T_PLAIN_MAX_BYTES;
constexpr int kKinPosePlain = 1;      // KinParams.pad bit 0
#ifndef RTB_PACKED_XCD
#define RTB_PACKED_XCD 1
#endif
constexpr int kKinPacked = 2;         // KinParams.pad bit 1: T is the packed (N, 16 + 6n) array [T | J], J is not written separately (run-time-n tile)

template <bool WANT_T, bool WANT_J, bool WANT_H, bool COALESCED>
__global__ __launch_bounds__(kWave) void k_kin(KinParams kp, DevChain dc,
                                              const double *__restrict__ q, double *__restrict__ T,
                                              double *__restrict__ J, double *__restrict__ H)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *rows = lds;
    double *qs = lds + kWave * kp.stride;
    const ConstChain ops = const_view(dc);
    const int lane = threadIdx.x;
    const int64_t tiles = (kp.N + kWave - 1) / kWave;
    const int W = 6 * kp.n;

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t cfg0 = tile * kWave;
        const int64_t cfg = cfg0 + lane;
        const int64_t left = kp.N - cfg0;
        const int ncfg = left < kWave ? (int)left : kWave;
        const bool live = lane < ncfg;

        kin_load_q(kp, q, cfg, lane, qs);
        Pose P;
        kin_walk<(WANT_J || WANT_H)>(kp, ops, lane, qs, rows, P);
        if (WANT_H) kin_hessian(kp, lane, rows, live, H + cfg * (int64_t)(kp.n * W));
        if (WANT_T && WANT_J && !WANT_H && COALESCED && (kp.pad & kKinPacked)) {      // wave-uniform: one contiguous run of [T | J] rows
            double *rowsT = qs + kWave * kp.qw;
            if (kp.has_base) pose_premul(P, kp.base);
            pose_store16(P, [&](int k, double v) { rowsT[lane * 17 + k] = v; });
            __syncthreads();
            kin_flush_packed(rowsT, rows, kp.stride, W, ncfg, T + cfg0 * (16 + W), lane);
            __syncthreads();
            continue;
        }
        if (WANT_J) {
            if (COALESCED) {
                __syncthreads();
                kin_flush(rows, kp.stride, W, ncfg, J + cfg0 * W, lane);
            } else {
                kin_store_own(rows, kp.stride, W, live, J + cfg * W, lane);
            }
        }
        if (WANT_T) {
            if (COALESCED) __syncthreads();
            kin_stage_T(kp, lane, rows, P);
            if (COALESCED) {
                __syncthreads();
                if (kp.pad & kKinPosePlain) kin_flush<false>(rows, kp.stride, 16, ncfg, T + cfg0 * 16, lane);      // wave-uniform
                else kin_flush<true>(rows, kp.stride, 16, ncfg, T + cfg0 * 16, lane);
            } else {
                kin_store_own(rows, kp.stride, 16, live, T + cfg * 16, lane);
            }
        }
        __syncthreads();  // rows/qs are rewritten by the next tile
    }
}

// ---------------------------------------------------------------- register-resident variant (n <= 8)
#ifndef RTB_REG_WAVES
#define RTB_REG_WAVES 3   // waves per SIMD the register allocator must leave room for (<= 168 VGPRs)
#endif
// One register-resident tile (64 configurations) of a chain with NJ joints; shared by k_kin_reg and k_fleet.
template <int NJ, bool WANT_T, bool WANT_J, bool PACKED = false>
__device__ __forceinline__ void reg_tile(const KinParams &kp, const ConstChain &cv, const double *__restrict__ q,
                                         double *__restrict__ T, double *__restrict__ J, double *buf, int lane,
                                         int64_t tile)
{
    constexpr int W = 6 * NJ;
    const int64_t cfg0 = tile * kWave;
    const int64_t left = kp.N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    Pose P;
    double jac[6 * NJ];
    reg_compute<NJ, WANT_J>(kp, cv, q, cfg0 + lane, P, jac);
    if constexpr (PACKED) {
        // T is the packed (N, 16 + W) array: rounds of kPRound lanes stage [T | J] and the wave writes the round's rows as one contiguous run
        static_assert(!PACKED || (WANT_T && WANT_J), "packed rows carry both");
        double *bufT = buf, *bufJ = buf + kPRound * 17;
#pragma unroll
        for (int r = 0; r < kWave / kPRound; ++r) {
            if (lane / kPRound == r) reg_stage_packed<NJ>(kp, P, jac, bufT, bufJ, lane % kPRound);
            __syncthreads();
            int rows = ncfg - r * kPRound;
            rows = rows < 0 ? 0 : (rows > kPRound ? kPRound : rows);
            kin_flush_packed(bufT, bufJ, W + 1, W, rows, T + (cfg0 + r * kPRound) * (16 + W), lane);
            __syncthreads();
        }
        return;
    }
#if RTB_T_FIRST
    if (WANT_T) {
        reg_stage_T(kp, P, buf, lane);
        __syncthreads();
        if (kp.pad & kKinPosePlain) kin_flush<false>(buf, 17, 16, ncfg, T + cfg0 * 16, lane);      // wave-uniform
        else kin_flush<true>(buf, 17, 16, ncfg, T + cfg0 * 16, lane);
        __syncthreads();
    }
#endif
    if (WANT_J) {
#pragma unroll
        for (int r = 0; r < kWave / kJRound; ++r) {
            if (lane / kJRound == r) reg_stage_J<NJ>(jac, buf, lane % kJRound);
            __syncthreads();
            int rows = ncfg - r * kJRound;
            rows = rows < 0 ? 0 : (rows > kJRound ? kJRound : rows);
            kin_flush(buf, W + 1, W, rows, J + (cfg0 + r * kJRound) * W, lane);
            __syncthreads();
        }
    }
#if !RTB_T_FIRST
    if (WANT_T) {
        reg_stage_T(kp, P, buf, lane);
        __syncthreads();
        if (kp.pad & kKinPosePlain) kin_flush<false>(buf, 17, 16, ncfg, T + cfg0 * 16, lane);
        else kin_flush<true>(buf, 17, 16, ncfg, T + cfg0 * 16, lane);
    }
#endif
}

// ONE tile per single-wave workgroup, no grid-stride loop: with a loop LICM hoists every segment's
// (loop-invariant) scalar loads into the preheader, where they overflow the SGPR file and come back
// as v_readlane pairs on each use.  The dispatcher balances the tiles instead.
template <int NJ, bool WANT_T, bool WANT_J, bool PACKED = false>
__global__ __launch_bounds__(kWave, (NJ <= kRegMaxJoints ? RTB_REG_WAVES : 2)) void k_kin_reg(KinParams kp, DevChain dc, const double *__restrict__ q,
                                                  double *__restrict__ T, double *__restrict__ J)
{
    extern __shared__ __attribute__((aligned(16))) double buf[];
    // (the packed form writes ~30 KB runs per wave: like the Hessian tile it may prefer the identity mapping -- RTB_PACKED_XCD, A/B'd in profiles/r05_*)
    reg_tile<NJ, WANT_T, WANT_J, PACKED>(kp, const_view(dc), q, T, J, buf, threadIdx.x, (PACKED && !RTB_PACKED_XCD) ? blockIdx.x : xcd_tile());
}

static int g_hess_mode = 0;   // A/B knob (rtbhip_tune "hess_mode")

// ---------------------------------------------------------------- Hessian, register-resident (n <= 8)
// H is 48 n^2 bytes per configuration (2352 B for the Panda): the kernel is a pure HBM writer.  The
// first version wrote each lane's block with 8-byte stores at a 2352-byte stride (3.27 ms per 1e6
// Panda configurations, 9 % of the HBM peak); here the wave stages its 64 Jacobians in LDS once and
// every lane generates the 16-byte pieces of the tile's contiguous output run on the fly.
template <int NJ>
__global__ __launch_bounds__(kWave, (NJ <= kRegMaxJoints ? 2 : 1)) void k_kin_hess(KinParams kp, DevChain dc, const double *__restrict__ q,
                                                       double *__restrict__ H)
{
    extern __shared__ __attribute__((aligned(16))) double buf[];
    const ConstChain cv = const_view(dc);
    const int lane = threadIdx.x;
    const int64_t cfg0 = (int64_t)blockIdx.x * kWave;
    const int64_t left = kp.N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    constexpr int W = 6 * NJ;
    {
        Pose P;
        double jac[6 * NJ];
        reg_compute<NJ, true>(kp, cv, q, cfg0 + lane, P, jac);
        double *mine = buf + lane * (W + 1);
#pragma unroll
        for (int k = 0; k < W; ++k) mine[k] = jac[k];
    }
    __syncthreads();
    double *dst = H + cfg0 * (int64_t)(NJ * W);
    hessian_run<NJ>(buf, W + 1, ncfg, lane, [&](int f, double a, double b, bool both) {
        if (both) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d w = {a, b};
            __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + f));
        } else {
            __builtin_nontemporal_store(a, dst + f);
        }
    });
}

// The tile's Hessians from the lanes' finished Jacobians (registers): R rounds, in round r the 64/R lanes of group r
// expand theirs (compile-time indices) into whole (n,6,n) rows of the LDS tile, which the full wave writes as one
// contiguous run.  Shared by k_kin_hess_tile (J from the chain walk) and k_hess_from_jac (J supplied by the caller).
template <int NJ, int R>
__device__ __forceinline__ void hess_tile_emit(double (&jac)[6 * NJ], double *buf, int lane, int ncfg, double *__restrict__ Hrun)
{
    constexpr int HW = NJ * 6 * NJ, S = HW | 1, G = kWave / R;
    const int grp = lane / G;
    double *mine = buf + (lane - grp * G) * S;
    for (int r = 0; r < R; ++r) {
        const int cnt = ncfg - r * G < G ? ncfg - r * G : G;
        if (cnt <= 0) break;                                         // wave-uniform
        if (grp == r) {
            // the expansion must stay inside its round: hoisted out of the loop it would hold all 6 n^2 entries in registers
#pragma unroll
            for (int k = 0; k < 6 * NJ; ++k) asm volatile("" : "+v"(jac[k]));
            hessian_from_jacobian(NJ, [&](int k) { return jac[k]; }, [&](int idx, double v) { mine[idx] = v; });
        }
        __syncthreads();
        double *dst = Hrun + (int64_t)(r * G) * HW;
        flush_rows<HW>(buf, S, cnt, lane, [&](int f, double a, double b) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d w = {a, b};
            __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + f));
        });
        __syncthreads();
    }
}

// Variant B (A/B knob hess_mode = 1): every lane forms the (6,n) block H[j] of its OWN configuration in
// registers (static indexing, no per-entry LDS gathers), the wave transposes the 64 blocks through LDS and
// writes them as 64 segments of 48n bytes (2352-byte stride between configurations), one round per j.
// Hessian, tiled form (default): every lane keeps its finished Jacobian in registers; in R rounds a group of 64/R lanes
// expands its Hessians (compile-time indices: ~9 fp64 ops per (j, i) block instead of ~50 integer + select
// instructions per entry in hessian_run) into an LDS tile of whole (n,6,n) rows, which the full wave then writes as one
// contiguous run.  LDS per wave (64/R) x (6 n^2 + 1) doubles: n = 7, R = 8 -> 18.9 KB.
template <int NJ, int R>
__global__ __launch_bounds__(kWave, (NJ <= kRegMaxJoints && R >= 8 ? 2 : 1)) void k_kin_hess_tile(KinParams kp, DevChain dc, const double *__restrict__ q,
                                                                                               double *__restrict__ H)
{
    extern __shared__ __attribute__((aligned(16))) double buf[];
    const ConstChain cv = const_view(dc);
    const int lane = threadIdx.x;
    const int64_t cfg0 = (int64_t)blockIdx.x * kWave;     // (the XCD-contiguous mapping of k_kin_reg costs this kernel 3 %)
    const int64_t left = kp.N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    Pose P;
    double jac[6 * NJ];
    reg_compute<NJ, true>(kp, cv, q, cfg0 + lane, P, jac);
    hess_tile_emit<NJ, R>(jac, buf, lane, ncfg, H + cfg0 * (int64_t)(NJ * 6 * NJ));
}
template <int NJ, int R>
static hipError_t launch_hess_tile(dim3 grid, hipStream_t s, const KinParams &kp, const DevChain &dc, const double *q, double *H)
{
    const size_t lds = (size_t)(kWave / R) * ((NJ * 6 * NJ) | 1) * sizeof(double);
    auto k = k_kin_hess_tile<NJ, R>;
    if (lds > 48 * 1024) { hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k, grid, dim3(kWave), lds, s, kp, dc, q, H);
    note_launch((int)grid.x, kWave, (int)lds);
    return hipGetLastError();
}
template <int NJ>
static hipError_t launch_hess_nj(dim3 grid, hipStream_t s, const KinParams &kp, const DevChain &dc, const double *q, double *H)
{
    // measured (Panda, 1e6, min of 10): R = 2 0.59 ms, R = 4 0.44 ms, R = 8 0.51 ms, R = 16 0.88 ms, hessian_run (mode 1) 0.61-0.64 ms
    if (g_hess_mode == 0) return launch_hess_tile<NJ, 4>(grid, s, kp, dc, q, H);
    if (g_hess_mode == 2) return launch_hess_tile<NJ, 8>(grid, s, kp, dc, q, H);
    const size_t lds = (size_t)kWave * (6 * NJ + 1) * sizeof(double);
    hipLaunchKernelGGL((k_kin_hess<NJ>), grid, dim3(kWave), lds, s, kp, dc, q, H);
    note_launch((int)grid.x, kWave, (int)lds);
    return hipGetLastError();
}

// ---------------------------------------------------------------- Hessian from a SUPPLIED Jacobian
// ETS_hessian0 / ETS_hessiane as the reference binds them take (ets, q, J, tool) and, when J is given, only run
// _ETS_hessian on it (core/fknm.cpp:583-783 -> core/methods.cpp:16-32): a pure function of J.  The tile's 64 Jacobians
// arrive as one contiguous run through LDS, every lane picks its own up into registers, then the same tile emission as
// k_kin_hess_tile.  Bytes: 48 n read + 48 n^2 written per Jacobian; HBM-bound.
template <int NJ, int R>
__global__ __launch_bounds__(kWave, 1) void k_hess_from_jac(const double *__restrict__ J, int64_t N, double *__restrict__ H)
{
    extern __shared__ __attribute__((aligned(16))) double buf[];
    constexpr int W = 6 * NJ;
    const int lane = threadIdx.x;
    const int64_t cfg0 = (int64_t)blockIdx.x * kWave;
    const int64_t left = N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    hj_load_tile(J + cfg0 * W, W, ncfg, buf, lane);
    __syncthreads();
    double jac[W];
    const double *mine = buf + lane * (W + 1);
#pragma unroll
    for (int k = 0; k < W; ++k) jac[k] = lane < ncfg ? mine[k] : 0.0;
    __syncthreads();
    hess_tile_emit<NJ, R>(jac, buf, lane, ncfg, H + cfg0 * (int64_t)(NJ * W));
}

// run-time n (chains of more than 10 joints): one lane per Jacobian, straight from and to global memory
__global__ __launch_bounds__(kWave) void k_hess_from_jac_any(int n, const double *__restrict__ J, int64_t N, double *__restrict__ H)
{
    const int64_t cfg = (int64_t)blockIdx.x * kWave + threadIdx.x;
    if (cfg >= N) return;
    const double *Jr = J + cfg * (int64_t)(6 * n);
    double *Hr = H + cfg * (int64_t)(6 * n * n);
    hessian_from_jacobian(n, [&](int k) { return Jr[k]; }, [&](int idx, double v) { Hr[idx] = v; });
}

template <int NJ>
static hipError_t launch_hess_from_jac_nj(dim3 grid, hipStream_t s, const double *J, int64_t N, double *H)
{
    constexpr int R = 4;
    const size_t a = (size_t)(kWave / R) * ((NJ * 6 * NJ) | 1), b = (size_t)kWave * (6 * NJ + 1);
    const size_t lds = (a > b ? a : b) * sizeof(double);
    auto k = k_hess_from_jac<NJ, R>;
    if (lds > 48 * 1024) { hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k, grid, dim3(kWave), lds, s, J, N, H);
    note_launch((int)grid.x, kWave, (int)lds);
    return hipGetLastError();
}

int launch_hess_from_jac(int n, const double *J, int64_t N, double *H, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (n < 1 || n > RTBHIP_MAX_JOINTS) { set_error("hessian_from_jacobian: n must be 1..32"); return RTBHIP_ELIMIT; }
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("hessian_from_jacobian: batch too large for one launch"); return RTBHIP_ELIMIT; }
    dim3 grid((unsigned)tiles);
    hipError_t e = hipSuccess;
    switch (n) {
    case 1: e = launch_hess_from_jac_nj<1>(grid, s, J, N, H); break;
    case 2: e = launch_hess_from_jac_nj<2>(grid, s, J, N, H); break;
    case 3: e = launch_hess_from_jac_nj<3>(grid, s, J, N, H); break;
    case 4: e = launch_hess_from_jac_nj<4>(grid, s, J, N, H); break;
    case 5: e = launch_hess_from_jac_nj<5>(grid, s, J, N, H); break;
    case 6: e = launch_hess_from_jac_nj<6>(grid, s, J, N, H); break;
    case 7: e = launch_hess_from_jac_nj<7>(grid, s, J, N, H); break;
    case 8: e = launch_hess_from_jac_nj<8>(grid, s, J, N, H); break;
    case 9: e = launch_hess_from_jac_nj<9>(grid, s, J, N, H); break;
    case 10: e = launch_hess_from_jac_nj<10>(grid, s, J, N, H); break;
    default:
        hipLaunchKernelGGL(k_hess_from_jac_any, grid, dim3(kWave), 0, s, n, J, N, H);
        note_launch((int)grid.x, kWave, 0);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return hip_fail(e, "k_hess_from_jac launch");
    return RTBHIP_OK;
}

// ---------------------------------------------------------------- fknm.Angle_Axis, batched
// e[i] = angle_axis(Te[i or 0], Tep[i or 0]) (core/fknm.cpp:112-162 -> core/ik.cpp:241-286).  256 B in, 48 B out per
// pair; both tiles through LDS as contiguous runs; a broadcast operand (count 1) is read by every lane from the same
// 128 bytes.  LDS: 64 x 17 doubles per operand, the first re-used for the 64 x 7 staging of e.
// RPY = true: the same staging around servo_rpy_lane (p_servo's method "rpy", servo_device.h).
// SERVO (compile-time): the whole of tools/p_servo.py:46-117 in this launch -- after the error vector, arrived = sum|e| < threshold (one byte per
// pair, read before the gain touches e) and v = gain * e leaves in its place.  Before round 5 the wrapper ran two more elementwise kernels for that.
struct ServoTail { double gain[6]; double threshold; unsigned char *arrived; };
template <bool RPY, bool SERVO = false>
__global__ __launch_bounds__(kWave) void k_angle_axis(const double *__restrict__ Te, int te_each, const double *__restrict__ Tep, int tep_each,
                                                     int64_t N, double *__restrict__ e, ServoTail tail = ServoTail())
{
    __shared__ __attribute__((aligned(16))) double a[kWave * kAaStride];
    __shared__ __attribute__((aligned(16))) double b[kWave * kAaStride];
    const int lane = threadIdx.x;
    const int64_t cfg0 = (int64_t)xcd_tile() * kWave;
    const int64_t left = N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    {
        // both operand tiles' loads in flight together (16 x 16 B per lane), then the LDS writes (servo_device.h: aa_fetch_tile)
        AaTile ta, tb;
        aa_fetch_tile(te_each ? Te + cfg0 * 16 : Te, te_each ? ncfg : 1, lane, ta);
        aa_fetch_tile(tep_each ? Tep + cfg0 * 16 : Tep, tep_each ? ncfg : 1, lane, tb);
        aa_store_tile(ta, te_each ? ncfg : 1, a, lane);
        aa_store_tile(tb, tep_each ? ncfg : 1, b, lane);
    }
    __syncthreads();
    double te16[12], tep16[12];
    const int la = te_each ? (lane < ncfg ? lane : 0) : 0, lb = tep_each ? (lane < ncfg ? lane : 0) : 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) { te16[k] = a[la * kAaStride + k]; tep16[k] = b[lb * kAaStride + k]; }
    __syncthreads();
    if constexpr (RPY) servo_rpy_lane(te16, tep16, a + lane * 7);
    else aa_lane(te16, tep16, a + lane * 7);
    if constexpr (SERVO) {
        double *mine = a + lane * 7;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) sum += fabs(mine[k]);                    // p_servo.py:113 `np.sum(np.abs(e)) < threshold`, on e itself
#pragma unroll
        for (int k = 0; k < 6; ++k) mine[k] = tail.gain[k] * mine[k];        // :108-111 v = k @ e (k = gain * eye(6) or diag(gain))
        if (lane < ncfg) tail.arrived[cfg0 + lane] = sum < tail.threshold ? 1 : 0;
    }
    __syncthreads();
    kin_flush(a, 7, 6, ncfg, e + cfg0 * 6, lane);
}

int launch_p_servo(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int64_t N, int method, const double *gain6, double threshold, double *v,
                   unsigned char *arrived, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("p_servo: batch too large for one launch"); return RTBHIP_ELIMIT; }
    ServoTail t;
    for (int k = 0; k < 6; ++k) t.gain[k] = gain6[k];
    t.threshold = threshold; t.arrived = arrived;
    if (method == 1) hipLaunchKernelGGL((k_angle_axis<true, true>), dim3((unsigned)tiles), dim3(kWave), 0, s, Te, nTe == N ? 1 : 0, Tep, nTep == N ? 1 : 0, N, v, t);
    else hipLaunchKernelGGL((k_angle_axis<false, true>), dim3((unsigned)tiles), dim3(kWave), 0, s, Te, nTe == N ? 1 : 0, Tep, nTep == N ? 1 : 0, N, v, t);
    note_launch((int)tiles, kWave, (int)(2 * kWave * kAaStride * sizeof(double)));
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return hip_fail(err, "k_angle_axis (servo) launch");
    return RTBHIP_OK;
}

int launch_angle_axis(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int64_t N, double *e, hipStream_t s, int method)
{
    if (N == 0) return RTBHIP_OK;
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("angle_axis: batch too large for one launch"); return RTBHIP_ELIMIT; }
    if (method == 1)
        hipLaunchKernelGGL((k_angle_axis<true, false>), dim3((unsigned)tiles), dim3(kWave), 0, s, Te, nTe == N ? 1 : 0, Tep,
                           nTep == N ? 1 : 0, N, e, ServoTail());
    else
        hipLaunchKernelGGL((k_angle_axis<false, false>), dim3((unsigned)tiles), dim3(kWave), 0, s, Te, nTe == N ? 1 : 0, Tep,
                           nTep == N ? 1 : 0, N, e, ServoTail());
    note_launch((int)tiles, kWave, (int)(2 * kWave * kAaStride * sizeof(double)));
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return hip_fail(err, "k_angle_axis launch");
    return RTBHIP_OK;
}

// ---------------------------------------------------------------- differential-kinematics consumers (n <= 8)
static int g_diff_sig = 1;     // rtbhip_tune("diff_sig", 0): k_kin_diff never takes a robot's structure instantiation (A/B, tests)
template <int NJ>
static hipError_t launch_diff_nj(int mode, dim3 grid, size_t lds, hipStream_t s, const KinParams &kp, const DevChain &dc,
                                 int axes, const double *q, const double *qd, double *out)
{
    if (mode == kDiffJdot) hipLaunchKernelGGL((k_kin_diff<NJ, kDiffJdot>), grid, dim3(kWave), lds, s, kp, dc, axes, q, qd, out);
    else if (mode == kDiffManip) hipLaunchKernelGGL((k_kin_diff<NJ, kDiffManip>), grid, dim3(kWave), lds, s, kp, dc, axes, q, qd, out);
    else if (mode == kDiffAnalytical) hipLaunchKernelGGL((k_kin_diff<NJ, kDiffAnalytical>), grid, dim3(kWave), lds, s, kp, dc, axes, q, qd, out);
    else if (mode == kDiffAnalyticalDot) hipLaunchKernelGGL((k_kin_diff<NJ, kDiffAnalyticalDot>), grid, dim3(kWave), lds, s, kp, dc, axes, q, qd, out);
    else hipLaunchKernelGGL((k_kin_diff<NJ, kDiffJacobm>), grid, dim3(kWave), lds, s, kp, dc, axes, q, qd, out);
    return hipGetLastError();
}

int launch_kin_diff(const Chain *c, const DevChain &dc, int mode, int axes, const double *q, const double *qd, int64_t N,
                    const Affine &tool, int frame, double *out, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (c->n < 1 || c->n > RTBHIP_MAX_JOINTS) { set_error("jacob_dot/manipulability/jacobm/jacob0_analytical: chains of 1..RTBHIP_MAX_JOINTS joints"); return RTBHIP_ELIMIT; }
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("jacob_dot/manipulability/jacobm: batch too large for one launch"); return RTBHIP_ELIMIT; }
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n); kp.frame = frame; kp.has_base = 0; kp.pad = 0; kp.N = N;
    for (int i = 0; i < 12; i++) kp.base[i] = 0.0;
    chain_tail(c, tool, kp.tail);
    const size_t lds = (size_t)reg_lds_doubles(c->n) * sizeof(double);
    dim3 grid((unsigned)tiles);
    hipError_t e = hipSuccess;
    // a structure instantiation of the walk for THIS robot (diff_kernel.h: k_kin_diff<NJ, MODE, SIG>), compiled at run time on first use; the general
    // kernel below serves until the code object is there and returns the same bits.  rtbhip_tune("diff_sig", 0): never.
    if (g_diff_sig && c->n <= kRegMaxJoints && !tool.used && mode != kDiffAnalyticalDot && jit_enabled()) {
        bool plain = true;
        for (int j = 0; j < c->n; ++j) plain = plain && !jm_prismatic(c->jmeta[j]) && !jm_flip(c->jmeta[j]);
        const SegSig sig = plain ? chain_signature(c->jmeta.data(), c->n) : 0;
        if (sig) {
            const int m = mode == kDiffJdot ? kDiffJdot : (mode == kDiffManip ? kDiffManip : (mode == kDiffAnalytical ? kDiffAnalytical : kDiffJacobm));
            if (hipFunction_t f = c->jit.get("diff_kernel.h", 48 + m, [&] { return "rtbhip::k_kin_diff<" + std::to_string(c->n) + ", " + std::to_string(m) + ", " + jit_hex(sig) + ">"; })) {
                void *args[] = {(void *)&kp, (void *)&dc, (void *)&axes, (void *)&q, (void *)&qd, (void *)&out};
                const int rc = jit_launch(f, grid, dim3(kWave), lds, s, args);
                if (rc != RTBHIP_OK) return rc;
                note_launch((int)grid.x, kWave, (int)lds);
                return RTBHIP_OK;
            }
        }
    }
    if (c->n > kDiffMax) {
        // a joint count without a built-in instantiation: the same k_kin_diff template, instantiated at run time (jit.cpp; seconds to a minute on first
        // use, then a file read).  The reference's loops take any n (robot/Robot.py:964-1235, robot/ETS.py:1687-1819).
        const int m = mode == kDiffJdot ? kDiffJdot : (mode == kDiffManip ? kDiffManip : (mode == kDiffAnalytical ? kDiffAnalytical : (mode == kDiffAnalyticalDot ? kDiffAnalyticalDot : kDiffJacobm)));
        hipFunction_t f = c->jit.get_wait("diff_kernel.h", 32 + m, [&] { return "rtbhip::k_kin_diff<" + std::to_string(c->n) + ", " + std::to_string(m) + ">"; });
        if (!f) return RTBHIP_ELIMIT;                         // (rtbhip_last_error says why: no hipRTC on this box, or the compiler's message)
        void *args[] = {(void *)&kp, (void *)&dc, (void *)&axes, (void *)&q, (void *)&qd, (void *)&out};
        const int rc = jit_launch(f, grid, dim3(kWave), lds, s, args);
        if (rc != RTBHIP_OK) return rc;
        note_launch((int)grid.x, kWave, (int)lds);
        return RTBHIP_OK;
    }
    switch (c->n) {
    case 1: e = launch_diff_nj<1>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 2: e = launch_diff_nj<2>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 3: e = launch_diff_nj<3>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 4: e = launch_diff_nj<4>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 5: e = launch_diff_nj<5>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 6: e = launch_diff_nj<6>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 7: e = launch_diff_nj<7>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 8: e = launch_diff_nj<8>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 9: e = launch_diff_nj<9>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 10: e = launch_diff_nj<10>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    // 11..16: one wave per SIMD, the longer ones with some scratch -- served, not fast
    case 11: e = launch_diff_nj<11>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 12: e = launch_diff_nj<12>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 13: e = launch_diff_nj<13>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 14: e = launch_diff_nj<14>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    case 15: e = launch_diff_nj<15>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    default: e = launch_diff_nj<16>(mode, grid, lds, s, kp, dc, axes, q, qd, out); break;
    }
    note_launch((int)grid.x, kWave, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "k_kin_diff launch");
    return RTBHIP_OK;
}

namespace {
int g_coalesced = 1;   // tuning knobs (rtbhip_tune)
int g_tiles_per_wave = 1;
int g_use_reg = 1;
}  // namespace

void kin_tune(const char *key, int value)
{
    std::string k(key);
    if (k == "coalesced") g_coalesced = value;
    if (k == "tiles_per_wave") g_tiles_per_wave = value < 1 ? 1 : value;
    if (k == "reg") g_use_reg = value;
    if (k == "hess_mode") g_hess_mode = value;
    if (k == "diff_sig") g_diff_sig = value != 0;
}

template <bool WT, bool WJ, bool WH>
static hipError_t launch_variant(bool coalesced, dim3 grid, size_t lds, hipStream_t s, const KinParams &kp,
                                 const DevChain &ops, const double *q, double *T, double *J, double *H)
{
    if (coalesced) {
        auto k = k_kin<WT, WJ, WH, true>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, grid, dim3(kWave), lds, s, kp, ops, q, T, J, H);
    } else {
        auto k = k_kin<WT, WJ, WH, false>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, grid, dim3(kWave), lds, s, kp, ops, q, T, J, H);
    }
    return hipGetLastError();
}

template <int NJ>
static hipError_t launch_reg(dim3 grid, size_t lds, hipStream_t s, const KinParams &kp, const DevChain &dc,
                             const double *q, double *T, double *J)
{
    if (T && J) hipLaunchKernelGGL((k_kin_reg<NJ, true, true>), grid, dim3(kWave), lds, s, kp, dc, q, T, J);
    else if (T) hipLaunchKernelGGL((k_kin_reg<NJ, true, false>), grid, dim3(kWave), lds, s, kp, dc, q, T, J);
    else hipLaunchKernelGGL((k_kin_reg<NJ, false, true>), grid, dim3(kWave), lds, s, kp, dc, q, T, J);
    return hipGetLastError();
}

template <int NJ>
static hipError_t launch_reg_packed(dim3 grid, size_t lds, hipStream_t s, const KinParams &kp, const DevChain &dc, const double *q, double *TJ)
{
    hipLaunchKernelGGL((k_kin_reg<NJ, true, true, true>), grid, dim3(kWave), lds, s, kp, dc, q, TJ, (double *)nullptr);
    return hipGetLastError();
}

// rtbhip_fkine_jacob_packed: rows [T (16, base applied) | J (6n)] of ONE (N, 16 + 6n) array -- a single write stream (SURVEY 8e's T||J message)
int launch_kin_packed(const Chain *c, const DevChain &ops, const double *q, int64_t N, const Affine &base,
                      const Affine &tool, int frame, double *TJ, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n); kp.frame = frame; kp.has_base = base.used;
    kp.pad = kKinPacked; kp.N = N;
    for (int i = 0; i < 12; i++) kp.base[i] = base.v[i];
    chain_tail(c, tool, kp.tail);
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("fkine_jacob_packed: batch too large for one launch"); return RTBHIP_ELIMIT; }
    dim3 grid((unsigned)tiles);
    if (g_use_reg && c->n >= 1 && c->n <= kKinRegMax) {
        const size_t rl = (size_t)reg_lds_doubles_packed(c->n) * sizeof(double);
        hipError_t e = hipSuccess;
        switch (c->n) {
        case 1: e = launch_reg_packed<1>(grid, rl, s, kp, ops, q, TJ); break;
        case 2: e = launch_reg_packed<2>(grid, rl, s, kp, ops, q, TJ); break;
        case 3: e = launch_reg_packed<3>(grid, rl, s, kp, ops, q, TJ); break;
        case 4: e = launch_reg_packed<4>(grid, rl, s, kp, ops, q, TJ); break;
        case 5: e = launch_reg_packed<5>(grid, rl, s, kp, ops, q, TJ); break;
        case 6: e = launch_reg_packed<6>(grid, rl, s, kp, ops, q, TJ); break;
        case 7: e = launch_reg_packed<7>(grid, rl, s, kp, ops, q, TJ); break;
        case 8: e = launch_reg_packed<8>(grid, rl, s, kp, ops, q, TJ); break;
        case 9: e = launch_reg_packed<9>(grid, rl, s, kp, ops, q, TJ); break;
        default: e = launch_reg_packed<10>(grid, rl, s, kp, ops, q, TJ); break;
        }
        note_launch((int)grid.x, kWave, (int)rl);
        if (e != hipSuccess) return hip_fail(e, "k_kin_reg (packed) launch");
        return RTBHIP_OK;
    }
    const size_t lds = kin_lds_bytes(c->n, c->q_width) + (size_t)kWave * 17 * sizeof(double);
    if (lds > 160 * 1024) { set_error("chain too large for the per-wave LDS staging"); return RTBHIP_ELIMIT; }
    hipError_t e = launch_variant<true, true, false>(true, grid, lds, s, kp, ops, q, TJ, TJ, nullptr);
    note_launch((int)grid.x, kWave, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "k_kin (packed) launch");
    return RTBHIP_OK;
}

int launch_kin(const Chain *c, const DevChain &ops, const double *q, int64_t N, const Affine &base,
               const Affine &tool, int frame, double *T, double *J, double *H, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    KinParams kp;
    kp.n = c->n;
    kp.qw = c->q_width;
    kp.stride = kin_stride(c->n);
    kp.frame = frame;
    kp.has_base = base.used;
    kp.pad = (T && J && !H && (long long)N * 128 <= kPoseCacheBytes) ? kKinPosePlain : 0;      // (see the store policy at the top of this file)
    kp.N = N;
    for (int i = 0; i < 12; i++) kp.base[i] = base.v[i];
    chain_tail(c, tool, kp.tail);
    const int64_t tiles = (N + kWave - 1) / kWave;
    int64_t g = (tiles + g_tiles_per_wave - 1) / g_tiles_per_wave;
    if (g > 0x7fffffff) g = 0x7fffffff;
    dim3 grid((unsigned)g);
    if (g_use_reg && H && !T && !J && c->n >= 1 && c->n <= kKinRegMax && tiles <= 0x7fffffff) {
        grid = dim3((unsigned)tiles);
        hipError_t e = hipSuccess;
        switch (c->n) {
        case 1: e = launch_hess_nj<1>(grid, s, kp, ops, q, H); break;
        case 2: e = launch_hess_nj<2>(grid, s, kp, ops, q, H); break;
        case 3: e = launch_hess_nj<3>(grid, s, kp, ops, q, H); break;
        case 4: e = launch_hess_nj<4>(grid, s, kp, ops, q, H); break;
        case 5: e = launch_hess_nj<5>(grid, s, kp, ops, q, H); break;
        case 6: e = launch_hess_nj<6>(grid, s, kp, ops, q, H); break;
        case 7: e = launch_hess_nj<7>(grid, s, kp, ops, q, H); break;
        case 8: e = launch_hess_nj<8>(grid, s, kp, ops, q, H); break;
        case 9: e = launch_hess_nj<9>(grid, s, kp, ops, q, H); break;
        default: e = launch_hess_nj<10>(grid, s, kp, ops, q, H); break;
        }
        if (e != hipSuccess) return hip_fail(e, "k_kin_hess launch");
        return RTBHIP_OK;
    }
    if (g_use_reg && !H && c->n >= 1 && c->n <= kKinRegMax && tiles <= 0x7fffffff) {
        grid = dim3((unsigned)tiles);
        // fkine alone stages only the 64 x 17 pose tile (8.7 KB): with the Jacobian's 32 x (6n + 1) rounds left out of the request more waves fit a CU
        const size_t rl = (size_t)(J ? reg_lds_doubles(c->n) : kWave * 17) * sizeof(double);
        hipError_t e = hipSuccess;
        switch (c->n) {
        case 1: e = launch_reg<1>(grid, rl, s, kp, ops, q, T, J); break;
        case 2: e = launch_reg<2>(grid, rl, s, kp, ops, q, T, J); break;
        case 3: e = launch_reg<3>(grid, rl, s, kp, ops, q, T, J); break;
        case 4: e = launch_reg<4>(grid, rl, s, kp, ops, q, T, J); break;
        case 5: e = launch_reg<5>(grid, rl, s, kp, ops, q, T, J); break;
        case 6: e = launch_reg<6>(grid, rl, s, kp, ops, q, T, J); break;
        case 7: e = launch_reg<7>(grid, rl, s, kp, ops, q, T, J); break;
        case 8: e = launch_reg<8>(grid, rl, s, kp, ops, q, T, J); break;
        case 9: e = launch_reg<9>(grid, rl, s, kp, ops, q, T, J); break;
        default: e = launch_reg<10>(grid, rl, s, kp, ops, q, T, J); break;
        }
        note_launch((int)grid.x, kWave, (int)rl);
        if (e != hipSuccess) return hip_fail(e, "k_kin_reg launch");
        return RTBHIP_OK;
    }
    const size_t lds = kin_lds_bytes(c->n, c->q_width);
    if (lds > 160 * 1024) { set_error("chain too large for the per-wave LDS staging"); return RTBHIP_ELIMIT; }
    const bool co = g_coalesced != 0;
    hipError_t e;
    const bool wt = T != nullptr, wj = J != nullptr, wh = H != nullptr;
    if (wt && wj && !wh) e = launch_variant<true, true, false>(co, grid, lds, s, kp, ops, q, T, J, H);
    else if (wt && !wj && !wh) e = launch_variant<true, false, false>(co, grid, lds, s, kp, ops, q, T, J, H);
    else if (!wt && wj && !wh) e = launch_variant<false, true, false>(co, grid, lds, s, kp, ops, q, T, J, H);
    else if (!wt && !wj && wh) e = launch_variant<false, false, true>(co, grid, lds, s, kp, ops, q, T, J, H);
    else if (wt && wj && wh) e = launch_variant<true, true, true>(co, grid, lds, s, kp, ops, q, T, J, H);
    else { set_error("launch_kin: unsupported output combination"); return RTBHIP_EINVAL; }
    note_launch((int)grid.x, kWave, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "k_kin launch");
    return RTBHIP_OK;
}

// ---------------------------------------------------------------- mixed fleet (BASELINE config 5)
// One launch walks up to kFleetMax different chains, each with its own batch: the global tile index
// is mapped to (chain, local tile) by a scan over the per-chain first-tile table (wave-uniform, in
// kernarg => SGPRs); chain length, joint count and LDS row stride are then run-time values of that
// block.  Same phases as k_kin.
constexpr int kFleetMax = 32;
struct FleetArgs {
    FleetEntry e[kFleetMax];
    int32_t count, frame;
    int64_t tiles;
    int64_t tile_base;   // first global tile of this launch (launches are chunked at 2^31-1 workgroups)
};

// One tile per workgroup (no grid-stride loop, as k_kin_reg).  Three launch classes, because registers and
// dynamic LDS are per-kernel / per-launch quantities: CLS 0 chains of 1..8 joints (register-resident tile, 3
// waves per SIMD), CLS 1 chains of 9..10 joints (register-resident, 2 waves per SIMD), CLS 2 longer chains
// (LDS tile).  Inside a class the joint count of a tile is wave-uniform: a switch picks the tile body.
template <int CLS, bool PACKED = false>
__global__ __launch_bounds__(kWave, (CLS == 0 ? RTB_REG_WAVES : 2)) void k_fleet(FleetArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int64_t gt = (int64_t)blockIdx.x + fa.tile_base;
    int ci = 0;
    for (int i = 1; i < fa.count; ++i)
        if (gt >= fa.e[i].tile0) ci = i;
    const FleetEntry &fe = fa.e[ci];
    KinParams kp;
    kp.n = fe.n; kp.qw = fe.q_width; kp.stride = fe.stride;
    kp.frame = fa.frame; kp.has_base = 0; kp.pad = PACKED ? kKinPacked : 0; kp.N = fe.N;
    const ConstChain ops = const_view(fe.dc);
    for (int k = 0; k < 9; ++k) kp.tail[k] = ops.seg[fe.n].r[k];
    for (int k = 0; k < 3; ++k) kp.tail[9 + k] = ops.seg[fe.n].t[k];
    // The chain's tiles are dealt to the XCDs as a launch of its own would deal them (trig.h: xcd_tile_of -- XCD x walks the x-th contiguous
    // eighth of THIS chain's tiles; the hardware's round-robin is by global workgroup id, a fixed rotation of the XCD labels per chain): 1 % on the
    // 16-arm fleet (1.391 -> 1.378 ms, profiles/r05_s_fleet_xcd_ab.txt).  (Its chains launched one by one in a loop take 1.196 ms -- but only because
    // each call then rewrites the SAME 128 MB pose array, which the memory-side cache absorbs (store policy above); a fleet step writes 2 GB of poses
    // once: it streams, at 0.72-0.73 of the HBM roof against 0.745 for a plain fill of the same bytes: profiles/r05_r_fleet_probe.jsonl.)
    const int64_t local = gt - fe.tile0, ctiles = (fe.N + kWave - 1) / kWave;
#ifndef RTB_FLEET_XCD
#define RTB_FLEET_XCD 1
#endif
    const int64_t tile = (RTB_FLEET_XCD && ctiles <= 0x7fffffff) ? (int64_t)xcd_tile_of((unsigned)ctiles, (unsigned)local) : local;
    if (CLS == 0) {
        switch (fe.n) {
        case 1: reg_tile<1, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        case 2: reg_tile<2, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        case 3: reg_tile<3, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        case 4: reg_tile<4, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        case 5: reg_tile<5, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        case 6: reg_tile<6, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        case 7: reg_tile<7, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        default: reg_tile<8, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile); return;
        }
    } else if (CLS == 1) {
        if (fe.n == 9) reg_tile<9, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile);
        else reg_tile<10, true, true, PACKED>(kp, ops, fe.q, fe.T, fe.J, lds, lane, tile);
        return;
    }
    double *rows = lds;
    double *qs = lds + kWave * kp.stride;
    const int64_t cfg0 = tile * kWave, cfg = cfg0 + lane;
    const int64_t left = kp.N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    const int W = 6 * kp.n;
    kin_load_q(kp, fe.q, cfg, lane, qs);
    Pose P;
    kin_walk<true>(kp, ops, lane, qs, rows, P);
    if (PACKED) {
        double *rowsT = qs + kWave * kp.qw;
        pose_store16(P, [&](int k, double v) { rowsT[lane * 17 + k] = v; });
        __syncthreads();
        kin_flush_packed(rowsT, rows, kp.stride, W, ncfg, fe.T + cfg0 * (16 + W), lane);
        return;
    }
    __syncthreads();
    kin_flush(rows, kp.stride, W, ncfg, fe.J + cfg0 * W, lane);
    __syncthreads();
    kin_stage_T(kp, lane, rows, P);
    __syncthreads();
    kin_flush(rows, kp.stride, 16, ncfg, fe.T + cfg0 * 16, lane);
}

// Dynamic LDS and the register budget are per-launch quantities, so a mixed fleet is walked as (at most)
// three launches, one per class -- a single launch capped every workgroup at 4 per CU (2.32 vs 1.57 ms for the
// 16-arm fleet of BASELINE config 5).
static int fleet_class_of(int n) { return n <= kRegMaxJoints ? 0 : (n <= kKinRegMax ? 1 : 2); }
static int launch_fleet_class(int cls, const std::vector<FleetEntry> &entries, int frame, hipStream_t s, bool packed);
int launch_fleet(const std::vector<FleetEntry> &all, int frame, hipStream_t s, bool packed)
{
    std::vector<FleetEntry> by[3];
    for (const FleetEntry &e : all) by[fleet_class_of(e.n)].push_back(e);
    for (int c = 0; c < 3; ++c)
        if (!by[c].empty()) { int rc = launch_fleet_class(c, by[c], frame, s, packed); if (rc != RTBHIP_OK) return rc; }
    return RTBHIP_OK;
}

template <bool PACKED>
static void fleet_go(int cls, unsigned g, size_t lds, hipStream_t s, const FleetArgs &fa)
{
    if (cls == 0) hipLaunchKernelGGL((k_fleet<0, PACKED>), dim3(g), dim3(kWave), lds, s, fa);
    else if (cls == 1) hipLaunchKernelGGL((k_fleet<1, PACKED>), dim3(g), dim3(kWave), lds, s, fa);
    else hipLaunchKernelGGL((k_fleet<2, PACKED>), dim3(g), dim3(kWave), lds, s, fa);
}
template <bool PACKED>
static const void *fleet_fn(int cls)
{
    return cls == 0 ? (const void *)k_fleet<0, PACKED> : (cls == 1 ? (const void *)k_fleet<1, PACKED> : (const void *)k_fleet<2, PACKED>);
}

static int launch_fleet_class(int cls, const std::vector<FleetEntry> &entries, int frame, hipStream_t s, bool packed)
{
    for (size_t first = 0; first < entries.size(); first += kFleetMax) {
        FleetArgs fa;
        std::memset(&fa, 0, sizeof fa);
        fa.count = (int)std::min<size_t>(kFleetMax, entries.size() - first);
        fa.frame = frame;
        size_t lds = 0;
        int64_t tiles = 0;
        for (int i = 0; i < fa.count; i++) {
            fa.e[i] = entries[first + i];
            fa.e[i].stride = kin_stride(fa.e[i].n);
            fa.e[i].tile0 = tiles;
            tiles += (fa.e[i].N + kWave - 1) / kWave;
            lds = std::max(lds, cls < 2 ? (size_t)(packed ? reg_lds_doubles_packed(fa.e[i].n) : reg_lds_doubles(fa.e[i].n)) * sizeof(double)
                                        : kin_lds_bytes(fa.e[i].n, fa.e[i].q_width) + (packed ? (size_t)kWave * 17 * sizeof(double) : 0));
        }
        fa.tiles = tiles;
        if (lds > 160 * 1024) { set_error("fleet: chain too large for LDS staging"); return RTBHIP_ELIMIT; }
        const void *kfn = packed ? fleet_fn<true>(cls) : fleet_fn<false>(cls);
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return hip_fail(e, "k_fleet attr");
        }
        for (int64_t t0 = 0; t0 < tiles; t0 += 0x7fffffff) {
            const int64_t g = std::min<int64_t>(0x7fffffff, tiles - t0);
            fa.tile_base = t0;
            if (packed) fleet_go<true>(cls, (unsigned)g, lds, s, fa); else fleet_go<false>(cls, (unsigned)g, lds, s, fa);
            note_launch((int)g, kWave, (int)lds);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "k_fleet launch");
        }
    }
    return RTBHIP_OK;
}

}  // namespace rtbhip
