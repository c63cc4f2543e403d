// kin_tile.h -- the per-tile phases of the fused fkine + Jacobian (+ Hessian) kernel.
//
// A tile is 64 consecutive configurations = one wavefront; lane l owns configuration cfg0 + l.
// LDS (per wave, private -- no inter-wave traffic):
//     rows : 64 x stride doubles, lane-major ("AoS"): rows[l*stride + slot].  stride is odd, so the
//            lane stride in dwords is 2 (mod 4): ds_write_b64 (16-lane groups, 32 banks) and
//            ds_read_b64 (32-lane groups, 64 banks) are conflict-free for per-lane access.
//            Holds the Jacobian scratch (p_j, z_j), then the finished J, then (re-used) the 4x4.
//     qs   : qw x 64 doubles, column-major ("SoA"): qs[c*64 + l].
// Why LDS at all: the outputs are (N,4,4) and (N,6,n) row-major, i.e. 128 B and 48n B contiguous
// PER CONFIGURATION.  A lane-per-configuration store would scatter 16 B pieces at a 336 B stride;
// staging the wave's outputs in LDS lets the wave write its 64 x 336 B = 21 KB of J (and 8 KB of T)
// as one contiguous run, 16 B per lane per instruction -- the fully coalesced pattern.
//
// Every phase is a __host__ __device__ function of (lane, LDS pointers): the GPU kernel calls them
// with lane = threadIdx.x between s_barriers; tests/emu calls them in a loop over lanes on the CPU.
#pragma once
#include "kin_device.h"

namespace rtbhip {

struct KinParams {
    int32_t n, qw, stride, frame;
    int32_t has_base, pad;
    int64_t N;
    double base[12];  // row-major 3x4, applied to T only
    double tail[12];  // C_n * tool as {R row-major (9), t (3)}
};

RTB_HD int kin_stride(int n)
{
    int s = 6 * n;
    if (s < 16) s = 16;
    return s + 1;  // odd
}
RTB_HD size_t kin_lds_bytes(int n, int qw) { return (size_t)kWave * (kin_stride(n) + qw) * sizeof(double); }

// phase A: this lane's joint coordinates -> qs (zero for lanes past the end of the batch)
RTB_HD void kin_load_q(const KinParams &kp, const double *__restrict__ q, int64_t cfg, int lane,
                       double *qs)
{
    const bool live = cfg < kp.N;
    const double *src = q + cfg * kp.qw;
    int c = 0;
    for (; c + 4 <= kp.qw; c += 4) {  // 4 independent loads in flight before the first LDS write
        double a0 = live ? src[c] : 0.0, a1 = live ? src[c + 1] : 0.0;
        double a2 = live ? src[c + 2] : 0.0, a3 = live ? src[c + 3] : 0.0;
        qs[(c + 0) * kWave + lane] = a0;
        qs[(c + 1) * kWave + lane] = a1;
        qs[(c + 2) * kWave + lane] = a2;
        qs[(c + 3) * kWave + lane] = a3;
    }
    for (; c < kp.qw; ++c) qs[c * kWave + lane] = live ? src[c] : 0.0;
}

// phase B: walk the chain, leave the pose in P (tool applied) and the finished J in rows.
template <bool WANT_J, class CV>
RTB_HD void kin_walk(const KinParams &kp, const CV &cv, int lane, const double *qs, double *rows, Pose &P)
{
    double *mine = rows + lane * kp.stride;
    chain_walk<WANT_J>(cv, kp.n, kp.tail, P,
                       [&](int c) { return qs[c * kWave + lane]; },
                       [&](int slot, double v) { mine[slot] = v; });
    if (WANT_J)
        jacobian_close(cv, kp.n, P, kp.frame, [&](int s) { return mine[s]; },
                       [&](int s, double v) { mine[s] = v; });
}

// phase D: this lane's 4x4 into rows (after the J flush; the region is re-used)
RTB_HD void kin_stage_T(const KinParams &kp, int lane, double *rows, Pose P)
{
    if (kp.has_base) pose_premul(P, kp.base);
    double *mine = rows + lane * kp.stride;
    pose_store16(P, [&](int k, double v) { mine[k] = v; });
}

// Output rows are written once and never re-read by this kernel: non-temporal 16-byte stores
// (global_store_dwordx4 ... nt) measured 0.0935 vs 0.105 ms per 1e6 Panda configurations on MI355X
// (the bare access-pattern probe, scripts/roofline_probe.hip: 0.0868 vs 0.1013 ms).
#ifndef RTB_NT_STORE
#define RTB_NT_STORE 1
#endif

// phases C / E: the wave writes `ncfg` staged rows of W doubles (W even) as one contiguous run.
// Lane l writes the 16-byte pieces l, l+64, l+128, ... of the run; (cfg, e) tracks which staged
// row / element piece f falls in without a division per piece.
template <bool NT = true>
RTB_HD void kin_flush(const double *rows, int stride, int W, int ncfg, double *__restrict__ dst,
                      int lane)
{
    const int total = ncfg * W;
    int f = 2 * lane;
    int cfg = f / W, e = f - cfg * W;
    const int da = 128 / W, db = 128 - da * W;
    for (; f < total; f += 128) {
        const double *src = rows + cfg * stride + e;
        double2 v;
        v.x = src[0];
        v.y = src[1];
#if RTB_NT_STORE && defined(__HIP_DEVICE_COMPILE__)
        if (NT) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d w = {v.x, v.y};
            __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + f));   // global_store_dwordx4 ... nt
        } else {
            *reinterpret_cast<double2 *>(dst + f) = v;
        }
#else
        *reinterpret_cast<double2 *>(dst + f) = v;
#endif
        e += db;
        cfg += da;
        if (e >= W) { e -= W; cfg += 1; }
    }
}

// PACKED output (SURVEY 8e's gather message: one (N, 16 + 6n) array, row = [T row-major 4x4 | J (6,n) C-order]): the wave writes `ncfg` staged
// rows of 16 + W doubles as ONE contiguous run -- a single write stream per launch, where the two-array form has two.  T and J are staged in
// separate LDS areas (rowsT: 17-double rows; rowsJ: `strideJ`-double rows); 16 and W are even, so a 16-byte piece never straddles the two.
template <bool NT = true>
RTB_HD void kin_flush_packed(const double *rowsT, const double *rowsJ, int strideJ, int W, int ncfg, double *__restrict__ dst, int lane)
{
    const int PW = 16 + W;
    const int total = ncfg * PW;
    int f = 2 * lane;
    int cfg = f / PW, e = f - cfg * PW;
    const int da = 128 / PW, db = 128 - da * PW;
    for (; f < total; f += 128) {
        const double *src = e < 16 ? rowsT + cfg * 17 + e : rowsJ + cfg * strideJ + (e - 16);
        double2 v;
        v.x = src[0];
        v.y = src[1];
#if RTB_NT_STORE && defined(__HIP_DEVICE_COMPILE__)
        if (NT) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d w = {v.x, v.y};
            __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + f));
        } else {
            *reinterpret_cast<double2 *>(dst + f) = v;
        }
#else
        *reinterpret_cast<double2 *>(dst + f) = v;
#endif
        e += db;
        cfg += da;
        if (e >= PW) { e -= PW; cfg += 1; }
    }
}

#if defined(__HIPCC__)
// contiguous run of ncfg rows of W doubles (row stride `stride` in LDS) -> global, 16 bytes per lane per
// piece; W may be odd (a piece may then straddle two rows, and the run may end on a single double)
__device__ __forceinline__ void flush_run(const double *rows, int stride, int W, int ncfg, double *__restrict__ dst, int lane)
{
    const int total = ncfg * W;
    for (int f = 2 * lane; f < total; f += 2 * kWave) {
        const int r = f / W, e = f - r * W;
        const double a = rows[r * stride + e];
        if (f + 1 < total) {
            const double b = (e + 1 < W) ? rows[r * stride + e + 1] : rows[(r + 1) * stride];
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d w = {a, b};
            __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(dst + f));
        } else {
            __builtin_nontemporal_store(a, dst + f);
        }
    }
}
#endif

// un-coalesced alternative (A/B baseline): every lane stores its own row straight from LDS
RTB_HD void kin_store_own(const double *rows, int stride, int W, bool live, double *__restrict__ dst_row,
                          int lane)
{
    if (!live) return;
    const double *mine = rows + lane * stride;
    for (int e = 0; e < W; e += 2) {
        double2 v;
        v.x = mine[e];
        v.y = mine[e + 1];
        *reinterpret_cast<double2 *>(dst_row + e) = v;
    }
}

// Hessian epilogue: straight from the finished J in rows to this lane's (n,6,n) block.
RTB_HD void kin_hessian(const KinParams &kp, int lane, const double *rows, bool live,
                        double *__restrict__ Hrow)
{
    if (!live) return;
    const double *mine = rows + lane * kp.stride;
    hessian_from_jacobian(kp.n, [&](int s) { return mine[s]; }, [&](int idx, double v) { Hrow[idx] = v; });
}

}  // namespace rtbhip
