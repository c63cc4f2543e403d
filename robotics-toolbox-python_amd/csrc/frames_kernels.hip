// frames_kernels.hip -- gfx950 kernel for the intermediate frames of a chain (fkine_all).  One lane owns one
// configuration and walks the chain once (frames_device.h).  Every frame is one 128-byte line of the (N, nmarks, 4, 4)
// output; a lane storing its own line 16 bytes at a time leaves the memory system with 64 partial lines per
// instruction (measured: 2.57 ms per 1e6 configurations x 8 frames, 0.42 TB/s).  So each frame goes through a
// wave-private LDS tile and is written eight configurations per instruction: eight adjacent lanes cover one whole line.
// Bound: HBM writes, 128 nmarks bytes per configuration against 8 n bytes in.
#include "frames_device.h"
#include "rtbhip_internal.h"

namespace rtbhip {

#define RTB_CONST __attribute__((address_space(4)))
struct ConstChainF {
    const RTB_CONST DevSeg *seg;
    const RTB_CONST int32_t *jmeta;
};

constexpr int kFW = 64, kFStride = 17;

__global__ __launch_bounds__(kFW) void k_frames(FrameTable ft, DevChain dc, int n, int qw, int64_t N, const double *__restrict__ q,
                                               double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double tile[kFW * kFStride];
    const int lane = threadIdx.x;
    const int64_t cfg0 = (int64_t)xcd_tile() * kFW;
    const int64_t left = N - cfg0;
    const int ncfg = left < kFW ? (int)left : kFW;
    ConstChainF cv;
    cv.seg = (const RTB_CONST DevSeg *)dc.seg;
    cv.jmeta = (const RTB_CONST int32_t *)dc.jmeta;
    const double *row = q + (cfg0 + (lane < ncfg ? lane : 0)) * qw;      // lanes past the end redo configuration cfg0: the walk is wave-uniform
    double *mine = tile + lane * kFStride;
    const int64_t stride = (int64_t)ft.nmarks * 16;
    frames_walk(cv, n, ft, [&](int c) { return row[c]; }, [&](int m, const Pose &P) {
        pose_store16(P, [&](int k, double v) { mine[k] = v; });
        __syncthreads();
        const int piece = lane & 7;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int c = it * 8 + (lane >> 3);
            if (c < ncfg) {
                typedef double v2d __attribute__((ext_vector_type(2)));
                const v2d w = {tile[c * kFStride + 2 * piece], tile[c * kFStride + 2 * piece + 1]};
                __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(out + (cfg0 + c) * stride + m * 16 + 2 * piece));
            }
        }
        __syncthreads();
    });
}

int launch_frames(const Chain *c, const DevChain &dc, const FrameTable &ft, const double *q, int64_t N, double *out, hipStream_t s)
{
    if (N == 0 || ft.nmarks == 0) return RTBHIP_OK;
    const int64_t blocks = (N + kFW - 1) / kFW;
    if (blocks > 0x7fffffff) { set_error("link_frames: batch too large for one launch"); return RTBHIP_ELIMIT; }
    hipLaunchKernelGGL(k_frames, dim3((unsigned)blocks), dim3(kFW), 0, s, ft, dc, c->n, c->q_width, N, q, out);
    note_launch((int)blocks, kFW, kFW * kFStride * 8);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_frames launch");
    return RTBHIP_OK;
}

// ---- measurement aid (bench.py): a plain streaming kernel with a given read / write mix -- what this GPU delivers, on this box, today, for the
// headline kernel's traffic (56 MB read, 464 MB written per launch), to put next to the 8 TB/s of the data sheet.  The pattern is the best
// one of the round-1 write probe (scripts/write_probe.hip, profiles/r01_k_write_probe.txt: the rate of hipMemsetAsync): single-wave workgroups,
// each storing exactly one aligned 4 KiB page with 16-byte non-temporal stores; every `rstep`-th workgroup also loads one 4 KiB page of `src`.
__global__ __launch_bounds__(64) void k_stream_probe(const double *__restrict__ src, int64_t rpages, int rstep, double *__restrict__ dst, int64_t wpages)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    v2d acc = {0.0, 0.0};
    if (b % rstep == 0 && b / rstep < rpages) {
        const v2d *p = reinterpret_cast<const v2d *>(src) + (b / rstep) * 256;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += p[k * 64 + lane];
    }
    if (b < wpages) {
        const v2d w = {(double)b, acc.x + acc.y == -1.2345e300 ? 1.0 : 0.0};      // the loaded values stay live; in practice they do not change what is stored
        v2d *o = reinterpret_cast<v2d *>(dst) + b * 256;
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(w, o + k * 64 + lane);
    }
}

int launch_stream_probe(const double *src, int64_t read_doubles, double *dst, int64_t write_doubles, hipStream_t s)
{
    const int64_t rpages = read_doubles / 512, wpages = write_doubles / 512;      // whole 4 KiB pages only (the tails are not touched)
    const int64_t grid = wpages > 0 ? wpages : rpages;
    if (grid <= 0) return RTBHIP_OK;
    if (grid > 0x7fffffff) { set_error("stream_probe: too large for one launch"); return RTBHIP_ELIMIT; }
    int rstep = rpages > 0 && wpages > 0 ? (int)(wpages / rpages) : 1;
    if (rstep < 1) rstep = 1;
    hipLaunchKernelGGL(k_stream_probe, dim3((unsigned)grid), dim3(64), 0, s, src, rpages, rstep, dst, wpages);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_stream_probe launch");
    return RTBHIP_OK;
}

}  // namespace rtbhip
