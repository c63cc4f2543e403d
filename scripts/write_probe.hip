// write_probe.hip -- how fast can 464 MB (the headline kernel's output volume) be WRITTEN on this chip, by pattern?
// Reference points for the practical ceiling of k_kin_reg's store phase (the kernel is write-dominated).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT> __global__ void fill_stride(v2d *out, long n16)   // grid-stride, 256-thread blocks
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
    v2d w = {1.0, 2.0};
    for (; i < n16; i += st) { if (NT) __builtin_nontemporal_store(w, out + i); else out[i] = w; }
}
template <bool NT, int RUN> __global__ __launch_bounds__(64) void fill_runs(v2d *out, long n16)   // one wave writes RUN KiB contiguous
{
    const long base = (long)blockIdx.x * (RUN * 64);
    v2d w = {1.0, 2.0};
    for (int k = 0; k < RUN; ++k) {
        long i = base + k * 64 + threadIdx.x;
        if (i < n16) { if (NT) __builtin_nontemporal_store(w, out + i); else out[i] = w; }
    }
}
template <int RUN, int MODE> __global__ __launch_bounds__(64) void fill_runs_rot(v2d *out, long n16)   // rotated / permuted piece order
{
    const long base = (long)blockIdx.x * (RUN * 64);
    v2d w = {1.0, 2.0};
    const int k0 = MODE == 0 ? (int)(blockIdx.x % RUN) : (MODE == 1 ? (int)((blockIdx.x * 7) % RUN) : 0);
    for (int kk = 0; kk < RUN; ++kk) {
        int k = kk + k0; if (k >= RUN) k -= RUN;
        if (MODE == 2) k = (int)((kk * 5 + blockIdx.x) % RUN);      // stride-5 permutation, block-dependent start
        long i = base + k * 64 + threadIdx.x;
        if (i < n16) __builtin_nontemporal_store(w, out + i);
    }
}
// the headline kernel's real shape: per tile a 21 KiB run (J) and an 8 KiB run (T) in two different arrays
template <int MODE> __global__ __launch_bounds__(64) void fill_TJ(v2d *outT, v2d *outJ, long tiles)
{
    const long t = blockIdx.x;
    v2d w = {1.0, 2.0};
    v2d *dJ = outJ + t * 1344, *dT = outT + t * 512;
    if (MODE == 0) {
        for (int k = 0; k < 21; ++k) __builtin_nontemporal_store(w, dJ + k * 64 + threadIdx.x);
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(w, dT + k * 64 + threadIdx.x);
    } else {
        const int k0 = (int)(t % 21), j0 = (int)(t % 8);
        for (int kk = 0; kk < 21; ++kk) { int k = kk + k0; if (k >= 21) k -= 21; __builtin_nontemporal_store(w, dJ + k * 64 + threadIdx.x); }
        for (int kk = 0; kk < 8; ++kk) { int k = kk + j0; if (k >= 8) k -= 8; __builtin_nontemporal_store(w, dT + k * 64 + threadIdx.x); }
    }
}
// (b) 4 KiB runs, but each single-wave workgroup writes REP of them, far apart (stride = n16 / REP)
template <int REP> __global__ __launch_bounds__(64) void fill_4k_far(v2d *out, long n16)
{
    const long per = n16 / REP;
    v2d w = {1.0, 2.0};
    for (int r = 0; r < REP; ++r) {
        const long base = r * per + (long)blockIdx.x * 256;
        for (int k = 0; k < 4; ++k) { long i = base + k * 64 + threadIdx.x; if (i < (r + 1) * per) __builtin_nontemporal_store(w, out + i); }
    }
}
// (c) 4 KiB per wave, 4 waves per workgroup (16 KiB per workgroup)
__global__ __launch_bounds__(256) void fill_4k_wg256(v2d *out, long n16)
{
    const long base = (long)blockIdx.x * 1024 + (threadIdx.x >> 6) * 256;
    v2d w = {1.0, 2.0};
    for (int k = 0; k < 4; ++k) { long i = base + k * 64 + (threadIdx.x & 63); if (i < n16) __builtin_nontemporal_store(w, out + i); }
}
// (d) 84 KiB per 4-wave workgroup, waves interleaved: wave v writes the 1 KiB chunks v, v+4, v+8, ...
__global__ __launch_bounds__(256) void fill_84k_interleaved(v2d *out, long n16)
{
    const long base = (long)blockIdx.x * (84 * 64);
    const int v = threadIdx.x >> 6, l = threadIdx.x & 63;
    v2d w = {1.0, 2.0};
    for (int k = v; k < 84; k += 4) { long i = base + k * 64 + l; if (i < n16) __builtin_nontemporal_store(w, out + i); }
}
// (e) 21 KiB per wave but written as 1 KiB pieces in bit-reversed-ish order (5 passes of stride 5)
template <class F> float timeit(F f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; w++) f();
    CK(hipEventRecord(a));
    for (int r = 0; r < 20; r++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 20;
}
int main()
{
    for (long bytes : {464000000L, 1856000000L}) {
        long n16 = bytes / 16;
        v2d *out; CK(hipMalloc(&out, bytes));
        auto rep = [&](const char *name, float ms) { printf("%ld MB %-28s %.4f ms  %.0f GB/s\n", bytes / 1000000, name, ms, bytes / ms / 1e6); };
        rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(out, 0, bytes, 0)); }));
        rep("grid-stride 2048x256", timeit([&] { fill_stride<false><<<2048, 256>>>(out, n16); }));
        rep("grid-stride 2048x256 nt", timeit([&] { fill_stride<true><<<2048, 256>>>(out, n16); }));
        rep("grid-stride 8192x256 nt", timeit([&] { fill_stride<true><<<8192, 256>>>(out, n16); }));
        rep("wave runs of 8 KiB nt", timeit([&] { fill_runs<true, 8><<<(unsigned)((n16 + 511) / 512), 64>>>(out, n16); }));
        rep("wave runs of 21 KiB nt", timeit([&] { fill_runs<true, 21><<<(unsigned)((n16 + 1343) / 1344), 64>>>(out, n16); }));
        rep("wave runs of 29 KiB nt", timeit([&] { fill_runs<true, 29><<<(unsigned)((n16 + 1855) / 1856), 64>>>(out, n16); }));
        rep("wave runs of 29 KiB", timeit([&] { fill_runs<false, 29><<<(unsigned)((n16 + 1855) / 1856), 64>>>(out, n16); }));
        rep("21 KiB nt, start = blk % 21", timeit([&] { fill_runs_rot<21, 0><<<(unsigned)((n16 + 1343) / 1344), 64>>>(out, n16); }));
        rep("21 KiB nt, start = 7 blk % 21", timeit([&] { fill_runs_rot<21, 1><<<(unsigned)((n16 + 1343) / 1344), 64>>>(out, n16); }));
        rep("21 KiB nt, stride-5 permuted", timeit([&] { fill_runs_rot<21, 2><<<(unsigned)((n16 + 1343) / 1344), 64>>>(out, n16); }));
        {
            long tiles = bytes / (29 * 1024);
            v2d *oT = out, *oJ = out + tiles * 512;
            rep("T(8K)+J(21K) per tile nt", timeit([&] { fill_TJ<0><<<(unsigned)tiles, 64>>>(oT, oJ, tiles); }));
            rep("T+J per tile nt, rotated", timeit([&] { fill_TJ<1><<<(unsigned)tiles, 64>>>(oT, oJ, tiles); }));
        }
        rep("4 KiB x5 far apart per wave", timeit([&] { fill_4k_far<5><<<(unsigned)((n16 / 5 + 255) / 256), 64>>>(out, n16); }));
        rep("4 KiB per wave, 256-thr WGs", timeit([&] { fill_4k_wg256<<<(unsigned)((n16 + 1023) / 1024), 256>>>(out, n16); }));
        rep("84 KiB per WG, 4 waves interl.", timeit([&] { fill_84k_interleaved<<<(unsigned)((n16 + 5375) / 5376), 256>>>(out, n16); }));
        rep("wave runs of 2 KiB nt", timeit([&] { fill_runs<true, 2><<<(unsigned)((n16 + 127) / 128), 64>>>(out, n16); }));
        rep("wave runs of 4 KiB nt", timeit([&] { fill_runs<true, 4><<<(unsigned)((n16 + 255) / 256), 64>>>(out, n16); }));
        CK(hipFree(out));
    }
    return 0;
}
