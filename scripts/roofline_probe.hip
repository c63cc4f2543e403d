// roofline_probe.hip -- practical HBM ceiling for the headline kernel's access mix on this chip:
// per 64-configuration tile a wave reads 64*56 B and writes 64*(128+336) B, both fully coalesced
// (16 B per lane per instruction), with no arithmetic.  Whatever this reaches is what a perfect
// fkine+jacob0 kernel could reach; the spec-peak roofline fraction is reported against 8 TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE> __device__ __forceinline__ void st(double2 *p, double2 v)
{
    if (MODE >= 2) { v2d w = {v.x, v.y}; __builtin_nontemporal_store(w, (v2d *)p); }
    else *p = v;
}
template <int MODE>  // 0: read 56 + write 464 per cfg ; 1: write only 464 ; 2: as 0 with nt stores ; 3: as 2 with nt loads
__global__ __launch_bounds__(64) void probe(const double2 *__restrict__ in, double2 *__restrict__ outT,
                                            double2 *__restrict__ outJ, long ncfg)
{
    const int lane = threadIdx.x;
    const long t = blockIdx.x;
    double2 acc = {1.0, 2.0};
    if (MODE != 1) {
        const double2 *src = in + t * (64 * 56 / 16);
        for (int k = lane; k < 64 * 56 / 16; k += 64) {
            double2 v;
            if (MODE == 3) { v2d w = __builtin_nontemporal_load((const v2d *)(src + k)); v.x = w.x; v.y = w.y; }
            else v = src[k];
            acc.x += v.x; acc.y += v.y;
        }
    }
    double2 *dT = outT + t * (64 * 128 / 16);
    for (int k = lane; k < 64 * 128 / 16; k += 64) st<MODE>(dT + k, acc);
    double2 *dJ = outJ + t * (64 * 336 / 16);
    for (int k = lane; k < 64 * 336 / 16; k += 64) st<MODE>(dJ + k, acc);
}

__global__ void copyk(const double2 *__restrict__ in, double2 *__restrict__ out, long n16)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) out[i] = in[i];
}

int main()
{
    for (long N : {1000000L, 4000000L, 16000000L}) {
        long tiles = N / 64;
        double2 *in, *oT, *oJ;
        CK(hipMalloc(&in, N * 56)); CK(hipMalloc(&oT, N * 128)); CK(hipMalloc(&oJ, N * 336));
        CK(hipMemset(in, 0, N * 56));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const char *names[4] = {"read56+write464", "write464", "read56+write464nt", "read56nt+write464nt"};
        for (int mode = 0; mode < 4; mode++) {
            auto go = [&]() {
                if (mode == 0) probe<0><<<tiles, 64>>>(in, oT, oJ, N);
                else if (mode == 1) probe<1><<<tiles, 64>>>(in, oT, oJ, N);
                else if (mode == 2) probe<2><<<tiles, 64>>>(in, oT, oJ, N);
                else probe<3><<<tiles, 64>>>(in, oT, oJ, N);
            };
            for (int w = 0; w < 3; w++) go();
            CK(hipEventRecord(a));
            for (int r = 0; r < 20; r++) go();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
            double bytes = mode == 1 ? N * 464.0 : N * 520.0;
            printf("N=%ld mode=%s ms=%.4f GB/s=%.1f\n", N, names[mode], ms, bytes / ms / 1e6);
        }
        // plain grid-stride copy of the same volume for reference
        long n16 = N * 260 / 16;
        double2 *ci, *co; CK(hipMalloc(&ci, n16 * 16)); CK(hipMalloc(&co, n16 * 16)); CK(hipMemset(ci, 0, n16 * 16));
        for (int w = 0; w < 3; w++) copyk<<<2048, 256>>>(ci, co, n16);
        CK(hipEventRecord(a));
        for (int r = 0; r < 20; r++) copyk<<<2048, 256>>>(ci, co, n16);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
        printf("N=%ld mode=copy260+260 ms=%.4f GB/s=%.1f\n", N, ms, N * 520.0 / ms / 1e6);
        CK(hipFree(in)); CK(hipFree(oT)); CK(hipFree(oJ)); CK(hipFree(ci)); CK(hipFree(co));
    }
    return 0;
}
