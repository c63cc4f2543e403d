// scripts/contraction_probe.hip -- why the kernels' sums of two products are written out (csrc/kin_device.h: mix_pp / mix_pm / dot3x).
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only scripts/contraction_probe.hip -o - | grep -E "^_Z1kILi|v_(fma|mul|add|fmac)_f64"
// The SAME source statement  o[0] = a * c + b * s  (fp contract fast, hipcc's default) compiles to
//   V = 0, 1, 3:  v_mul(s, b) ; v_fmac(c, a)        = fma(a, c, round(b s))
//   V = 2:        v_mul(c, a) ; v_fmac(s, b)        = fma(b, s, round(a c))
// depending only on how a and b were PRODUCED (V = 0: both end a chain of three fused multiply-adds -- the general constant product; V = 1 .. 3:
// the forms a structure instantiation leaves -- a bare copy, one product plus an addend).  LLVM orders the operands of a commutative fadd by the
// depth of their expression trees (Reassociate's ranks) before the DAG combiner fuses the left product.  Both results are correctly rounded
// evaluations of a c + b s; they are not the same number.  A kernel instantiated for a robot's structure therefore must not leave any
// `x y + u v` to the compiler if it is to return the general kernel's bits (round 6: k_ik's Panda instantiation drifted by up to 7e-6 in q on
// rows whose searches took another path, 1e-10 elsewhere).  The difference b * c - a * s (fsub: not commutative) is stable.
#include <hip/hip_runtime.h>
__device__ __forceinline__ double kterm(double c, double x)
{
#pragma clang fp contract(off)
    return c * x;
}
__device__ __forceinline__ double kaddp(double acc, double x)
{
#pragma clang fp contract(off)
    return acc + x;
}
__device__ __forceinline__ double kfma(double c, double x, double acc)
{
#pragma clang fp contract(off)
    return __builtin_fma(c, x, acc);
}
__device__ __forceinline__ double ecopy(double x) { asm("" : "+v"(x)); return x; }
template <int V>
__global__ void k(double *o, const double *i)
{
    double x = i[0], y = i[1], z = i[2], c0 = i[3], c3 = i[4], c6 = i[5], c = i[6], s = i[7];
    double c1 = i[8], c4 = i[9], c7 = i[10];
    double a, b;
    if (V == 0) { a = kfma(c6, z, kfma(c0, x, kterm(c3, y))); b = kfma(c7, z, kfma(c1, x, kterm(c4, y))); }
    if (V == 1) { a = ecopy(x); b = kaddp(kterm(c4, y), z); }
    if (V == 2) { a = kaddp(kterm(c4, y), z); b = ecopy(x); }
    if (V == 3) { a = ecopy(x); b = kfma(c7, z, -y); }
    o[0] = a * c + b * s;
    o[1] = b * c - a * s;
}
template __global__ void k<0>(double *, const double *);
template __global__ void k<1>(double *, const double *);
template __global__ void k<2>(double *, const double *);
template __global__ void k<3>(double *, const double *);
