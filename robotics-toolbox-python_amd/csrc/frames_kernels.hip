// frames_kernels.hip -- gfx950 kernel for the intermediate frames of a chain (fkine_all).  One lane owns one
// configuration and walks the chain once (frames_device.h); every frame is one 128-byte line of the (N, nmarks, 4, 4)
// output, written by its lane as eight 16-byte non-temporal stores -- whole lines, so no LDS transposition is needed.
// Bound: HBM writes, 128 nmarks bytes per configuration against 8 n bytes in.
#include "frames_device.h"
#include "rtbhip_internal.h"

namespace rtbhip {

#define RTB_CONST __attribute__((address_space(4)))
struct ConstChainF {
    const RTB_CONST DevSeg *seg;
    const RTB_CONST int32_t *jmeta;
};

__global__ __launch_bounds__(256) void k_frames(FrameTable ft, DevChain dc, int n, int qw, int64_t N, const double *__restrict__ q,
                                               double *__restrict__ out)
{
    const int64_t cfg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cfg >= N) return;
    ConstChainF cv;
    cv.seg = (const RTB_CONST DevSeg *)dc.seg;
    cv.jmeta = (const RTB_CONST int32_t *)dc.jmeta;
    const double *row = q + cfg * qw;
    double *dst = out + cfg * (int64_t)ft.nmarks * 16;
    frames_walk(cv, n, ft, [&](int c) { return row[c]; }, [&](int m, const Pose &P) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d *o = reinterpret_cast<v2d *>(dst + m * 16);
        const v2d a = {P.r00, P.r01}, b = {P.r02, P.tx}, c = {P.r10, P.r11}, d = {P.r12, P.ty};
        const v2d e = {P.r20, P.r21}, f = {P.r22, P.tz}, g = {0.0, 0.0}, h = {0.0, 1.0};
        __builtin_nontemporal_store(a, o); __builtin_nontemporal_store(b, o + 1);
        __builtin_nontemporal_store(c, o + 2); __builtin_nontemporal_store(d, o + 3);
        __builtin_nontemporal_store(e, o + 4); __builtin_nontemporal_store(f, o + 5);
        __builtin_nontemporal_store(g, o + 6); __builtin_nontemporal_store(h, o + 7);
    });
}

int launch_frames(const Chain *c, const DevChain &dc, const FrameTable &ft, const double *q, int64_t N, double *out, hipStream_t s)
{
    if (N == 0 || ft.nmarks == 0) return RTBHIP_OK;
    const int64_t blocks = (N + 255) / 256;
    if (blocks > 0x7fffffff) { set_error("link_frames: batch too large for one launch"); return RTBHIP_ELIMIT; }
    hipLaunchKernelGGL(k_frames, dim3((unsigned)blocks), dim3(256), 0, s, ft, dc, c->n, c->q_width, N, q, out);
    note_launch((int)blocks, 256, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_frames launch");
    return RTBHIP_OK;
}

}  // namespace rtbhip
