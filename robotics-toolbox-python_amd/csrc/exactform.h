// exactform.h -- the exact-form algebra: structured products that equal the general one BY CONSTRUCTION.
//
// The kernels multiply by robot constants -- the folded segments of a chain (kin_device.h), the rotations between link groups (tree_device.h),
// sin / cos of a DH twist (rne_device.h).  Most of those constants are not general: exact zeros, exact +-1 (a quarter turn, an axis permutation,
// a pure translation ...).  A kernel instantiated for a robot's STRUCTURE (ahead of time for the robots of the benchmarks, at run time by jit.cpp for
// every other robot) drops the work those entries do not need.  For that to be free of any parity cost the structured product has to return the
// general product's bits -- not "to rounding", and not "as observed with this compiler":
//
//   * a constant known at compile time to be exactly 0, +1 or -1 has a KIND (kC0, kCP, kCN); everything else is kCA;
//   * dotk<K0, K1, K2>(c0, x0, c1, x1, c2, x2) is  c0 x0 + c1 x1 + c2 x2  evaluated as ONE fixed sequence of correctly rounded operations,
//         fma(c2, x2, fma(c1, x1, round(c0 x0))),
//     written with explicit fused multiply-adds under `fp contract(off)`: the operations carry no `contract` flag, so the compiler neither fuses
//     them with their neighbours nor splits them (hipcc's default is -ffp-contract=fast-honor-pragmas: only flagged pairs are fused);
//   * a structured instance is the SAME sequence after, mechanically,   fma(0, x, a) -> a,   fma(+-1, x, a) -> a +- x,   round(0 x) innermost ->
//     the next term starts the chain,   round(+-1 x) -> +-x.   Each rewrite returns the same floating-point number for finite operands (only the
//     sign of a zero can differ);
//   * a result that is a bare copy of an input is made opaque to the optimiser (exact_copy), so that a consumer cannot fuse with whatever
//     produced the input -- in the general kernel the same value comes out of a fused multiply-add, which nothing fuses with.
//
// Parity evidence gathered on the general kernel therefore transfers to every instantiation; tests/test_jit_gpu.py and the signature tests check
// "the same bits" on the device all the same.  (Round 5's forms were written to land where the compiler's own contraction of the general expression
// landed: equal as OBSERVED -- and the UR signature's IK came out 6e-9 away, the compiler having fused across a pure-permutation constant.)
#pragma once
#ifndef RTB_HD
#define RTB_HD __host__ __device__ __forceinline__
#endif

namespace rtbhip {

constexpr int kC0 = 0, kCP = 1, kCN = 2, kCA = 3;         // coefficient kinds: exact 0, exact +1, exact -1, anything

RTB_HD double exact_copy(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));                                    // no instruction: the value's provenance ends here
#endif
    return x;
}
// kinds as (compile-time constant) ARGUMENTS: the `if` chains fold after inlining -- the callers pass constants of an unrolled loop body
RTB_HD double kterm(int K, double c, double x)             // round(c x): starts a chain
{
#pragma clang fp contract(off)
    if (K == kCP) return x;
    if (K == kCN) return -x;
    return c * x;
}
RTB_HD double kacc(int K, double c, double x, double acc)  // fma(c, x, acc): continues it
{
#pragma clang fp contract(off)
    if (K == kC0) return acc;
    if (K == kCP) return acc + x;
    if (K == kCN) return acc - x;
    return __builtin_fma(c, x, acc);
}
RTB_HD double dotk_rt(int K0, int K1, int K2, double c0, double x0, double c1, double x1, double c2, double x2)
{
    const int nz = (K0 != kC0) + (K1 != kC0) + (K2 != kC0);
    if (nz == 0) return 0.0;
    if (nz == 1) {                                         // one term: a rounded product, or a bare (negated) copy
        const int K = K0 != kC0 ? K0 : (K1 != kC0 ? K1 : K2);
        const double c = K0 != kC0 ? c0 : (K1 != kC0 ? c1 : c2), x = K0 != kC0 ? x0 : (K1 != kC0 ? x1 : x2);
        return K == kCA ? kterm(kCA, c, x) : exact_copy(kterm(K, c, x));
    }
    if (K0 != kC0) return kacc(K2, c2, x2, kacc(K1, c1, x1, kterm(K0, c0, x0)));
    return kacc(K2, c2, x2, kterm(K1, c1, x1));            // K0 absent, K1 and K2 present
}
template <int K0, int K1, int K2>
RTB_HD double dotk(double c0, double x0, double c1, double x1, double c2, double x2) { return dotk_rt(K0, K1, K2, c0, x0, c1, x1, c2, x2); }

// acc + c1 x1 - c2 x2 as  fma(c1, x1, fma(-c2, x2, acc))  with the terms of an absent (exactly zero) factor dropped: cross-product components
RTB_HD double fm2k(bool has1, bool has2, double u1, double v1, double u2, double v2, double acc)
{
#pragma clang fp contract(off)
    if (has1 && has2) return __builtin_fma(u1, v1, __builtin_fma(-u2, v2, acc));
    if (has1) return __builtin_fma(u1, v1, acc);
    if (has2) return __builtin_fma(-u2, v2, acc);
    return acc;
}

// kind of entry k (row-major) of a constant rotation of structure class cls (rtbhip_internal.h: kSeg*; chain.cpp: seg_class_bits decides the class from
// the exact zeros and ones -- the diagonal of a quarter turn keeps its cos(pi/2) = 6.1e-17 and is kCA).  One 18-bit word per class (2 bits per
// entry) behind a switch: with a constant class the whole lookup folds to a constant, no table lives in memory.
RTB_HD constexpr unsigned seg_kind_word(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8)
{
    return (unsigned)a0 | ((unsigned)a1 << 2) | ((unsigned)a2 << 4) | ((unsigned)a3 << 6) | ((unsigned)a4 << 8) | ((unsigned)a5 << 10) | ((unsigned)a6 << 12) |
           ((unsigned)a7 << 14) | ((unsigned)a8 << 16);
}
RTB_HD constexpr unsigned seg_kinds(int cls)
{
    constexpr int A = kCA, P = kCP, N = kCN, O = kC0;
    switch (cls) {
    case 1: return seg_kind_word(P, O, O, O, P, O, O, O, P);       // identity
    case 2: return seg_kind_word(P, O, O, O, A, N, O, P, A);       // RxP  (r5, r7) = (-1, +1)
    case 3: return seg_kind_word(P, O, O, O, A, P, O, N, A);       // RxN
    case 4: return seg_kind_word(P, O, O, O, A, A, O, A, A);       // Rx
    case 5: return seg_kind_word(A, O, N, O, P, O, P, O, A);       // RyP  (r2, r6) = (-1, +1)
    case 6: return seg_kind_word(A, O, P, O, P, O, N, O, A);       // RyN
    case 7: return seg_kind_word(A, O, A, O, P, O, A, O, A);       // Ry
    case 8: return seg_kind_word(A, N, O, P, A, O, O, O, P);       // RzP  (r1, r3) = (-1, +1)
    case 9: return seg_kind_word(A, P, O, N, A, O, O, O, P);       // RzN
    case 10: return seg_kind_word(A, A, O, A, A, O, O, O, P);      // Rz
    case 11: return seg_kind_word(O, O, P, P, O, O, O, P, O);      // permA: r3 = r7 = r2 = 1 (new columns = old (1, 2, 0))
    case 12: return seg_kind_word(O, P, O, O, O, P, P, O, O);      // permB: r6 = r1 = r5 = 1 (new columns = old (2, 0, 1))
    default: return seg_kind_word(A, A, A, A, A, A, A, A, A);      // general (0)
    }
}
RTB_HD constexpr int seg_kind(int cls, int k) { return (int)((seg_kinds(cls) >> (2 * k)) & 3u); }

}  // namespace rtbhip
