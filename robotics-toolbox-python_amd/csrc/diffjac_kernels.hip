// diffjac_kernels.hip -- k_diff_from_jac: manipulability / manipulability Jacobian of SUPPLIED Jacobians (rtbhip_manipulability_from_jacobian,
// rtbhip_jacobm_from_jacobian; reference robot/Robot.py:701-905 with J=, :1101-1235 with J= / H=).
// One lane per Jacobian.  The tile's 64 Jacobians (48 n bytes each, contiguous) come in as one coalesced run through LDS (hj_load_tile),
// every lane picks its own up into registers and runs the same body the chain-walking kernel runs (diff_device.h).  HBM-bound for the
// measure (48 n B in, 8 B out); a supplied Hessian (48 n^2 B per row) is read by its own lane in 16-byte pieces, whole cache lines per lane.
#include <hip/hip_runtime.h>
#include "rtbhip_internal.h"
#include "kin_tile.h"
#include "servo_device.h"
#include "diffjac_device.h"

namespace rtbhip {

enum { kFromJacManip = 0, kFromJacJacobm = 1, kFromJacJacobmH = 2 };
constexpr int kFromJacMax = 16;

template <int NJ, int MODE>
__global__ __launch_bounds__(kWave, 1) void k_diff_from_jac(const double *__restrict__ J, const double *__restrict__ H, int64_t N, int axes,
                                                            double *__restrict__ out)
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
    // a lane beyond the batch works on a full-row-rank filler (row r has a one in column r mod NJ and, for NJ < 6, J J^T is made
    // non-singular by the axes the measure keeps only when NJ >= their count -- so its results may still be inf / NaN for NJ < 6): nothing is
    // stored for it, the arithmetic only has to stay in lock-step with the wave
#pragma unroll
    for (int k = 0; k < W; ++k) jac[k] = lane < ncfg ? mine[k] : ((k / NJ) % NJ == (k % NJ) ? 1.0 : 0.0);
    __syncthreads();
    if (MODE == kFromJacManip) {
        const int method = (axes >> 8) & 3;
        const double m = method == 0 ? manipulability_yoshikawa<NJ>(jac, axes & 63) : manipulability_singular<NJ>(jac, axes & 63, method);
        if (lane < ncfg) out[cfg0 + lane] = m;
        return;
    }
    double jm[NJ];
    if (MODE == kFromJacJacobm) {
        jacobm<NJ>(jac, axes & 63, jm);                       // H = hessian(J) formed on the fly (Robot.py:1206)
    } else {
        const double *Hr = H + (cfg0 + (lane < ncfg ? lane : 0)) * (int64_t)(NJ * W);
        jacobm_with_hessian<NJ>(jac, axes & 63, [&](int i, int b, int k) { return Hr[(i * 6 + b) * NJ + k]; }, jm);
    }
    constexpr int S = NJ | 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) buf[lane * S + j] = jm[j];
    __syncthreads();
    flush_run(buf, S, NJ, ncfg, out + cfg0 * NJ, lane);
}

template <int NJ>
static hipError_t launch_from_jac_nj(int mode, dim3 grid, hipStream_t s, const double *J, const double *H, int64_t N, int axes, double *out)
{
    const size_t lds = (size_t)kWave * (6 * NJ + 1) * sizeof(double);
    if (mode == kFromJacManip) hipLaunchKernelGGL((k_diff_from_jac<NJ, kFromJacManip>), grid, dim3(kWave), lds, s, J, H, N, axes, out);
    else if (mode == kFromJacJacobm) hipLaunchKernelGGL((k_diff_from_jac<NJ, kFromJacJacobm>), grid, dim3(kWave), lds, s, J, H, N, axes, out);
    else hipLaunchKernelGGL((k_diff_from_jac<NJ, kFromJacJacobmH>), grid, dim3(kWave), lds, s, J, H, N, axes, out);
    note_launch((int)grid.x, kWave, (int)lds);
    return hipGetLastError();
}

// mode 0: out (N) = measure, axes bits 8..9 = method; mode 1 / 2: out (N,n) = manipulability Jacobian, H NULL / supplied
int launch_diff_from_jac(int mode, int n, const double *J, const double *H, int64_t N, int axes, double *out, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (n < 1 || n > kFromJacMax) { set_error("manipulability/jacobm from a Jacobian: 1..16 joints on the device"); return RTBHIP_ELIMIT; }
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("manipulability/jacobm from a Jacobian: batch too large for one launch"); return RTBHIP_ELIMIT; }
    dim3 grid((unsigned)tiles);
    hipError_t e = hipSuccess;
    switch (n) {
    case 1: e = launch_from_jac_nj<1>(mode, grid, s, J, H, N, axes, out); break;
    case 2: e = launch_from_jac_nj<2>(mode, grid, s, J, H, N, axes, out); break;
    case 3: e = launch_from_jac_nj<3>(mode, grid, s, J, H, N, axes, out); break;
    case 4: e = launch_from_jac_nj<4>(mode, grid, s, J, H, N, axes, out); break;
    case 5: e = launch_from_jac_nj<5>(mode, grid, s, J, H, N, axes, out); break;
    case 6: e = launch_from_jac_nj<6>(mode, grid, s, J, H, N, axes, out); break;
    case 7: e = launch_from_jac_nj<7>(mode, grid, s, J, H, N, axes, out); break;
    case 8: e = launch_from_jac_nj<8>(mode, grid, s, J, H, N, axes, out); break;
    case 9: e = launch_from_jac_nj<9>(mode, grid, s, J, H, N, axes, out); break;
    case 10: e = launch_from_jac_nj<10>(mode, grid, s, J, H, N, axes, out); break;
    case 11: e = launch_from_jac_nj<11>(mode, grid, s, J, H, N, axes, out); break;
    case 12: e = launch_from_jac_nj<12>(mode, grid, s, J, H, N, axes, out); break;
    case 13: e = launch_from_jac_nj<13>(mode, grid, s, J, H, N, axes, out); break;
    case 14: e = launch_from_jac_nj<14>(mode, grid, s, J, H, N, axes, out); break;
    case 15: e = launch_from_jac_nj<15>(mode, grid, s, J, H, N, axes, out); break;
    default: e = launch_from_jac_nj<16>(mode, grid, s, J, H, N, axes, out); break;
    }
    if (e != hipSuccess) return hip_fail(e, "k_diff_from_jac launch");
    return RTBHIP_OK;
}

}  // namespace rtbhip
