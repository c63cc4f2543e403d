// partial_device.h -- n-th partial derivative of the forward kinematics (ETS.partial_fkine0,
// robot/ETS.py:1821-2013).  The reference builds dT[c] (c >= 3) from the Jacobian dT[1] and the Hessian
// dT[2] by repeated application of the product rule to  H[k,:,j] = J_w[:,k] x J[:,j]:  a column of the
// order-c tensor is the sum of 2^(c-2) cross products  rot(A) x B  where A and B are columns of two
// lower-order tensors whose orders add up to c.  The host expands that rule once per call into a plan
// (which tensor, which of the c derivative indices select the column); one lane then owns one column.
// Tensor of order a (a = 1 Jacobian) is stored (N, n^(a-1), 6, n), C order, as the reference returns it.
#pragma once
#include <stdint.h>
#include "kin_device.h"

namespace rtbhip {

constexpr int kPartialMaxOrder = 6;                    // highest derivative order served on the device
constexpr int kPartialMaxTerms = 1 << (kPartialMaxOrder - 2);

struct PartialTerm {
    int32_t oa, ob;                                    // orders of the two factors (>= 1), oa + ob = c  (dwords: scalar loads)
    int8_t ia[kPartialMaxOrder];                       // oa positions in the digit vector: ia[0] picks the column,
    int8_t ib[kPartialMaxOrder];                       // ia[1..] the leading (slice) indices, least significant first
    // the same selection as a linear form: offset of row 0 of the factor's column inside its tensor
    // = sum_i digit[i] * ca[i]  (ca[ia[0]] = 1, ca[ia[t]] = 6 n * n^(t-1), 0 for digits the factor does not use)
    int32_t ca[kPartialMaxOrder], cb[kPartialMaxOrder];
};

struct PartialPlan {
    int32_t n, c, nterms, cols;                        // cols = n^c columns per configuration (< 2^31: n <= 32, c <= 6)
    int64_t N;
    int64_t size[kPartialMaxOrder + 1];                // size[a] = doubles per configuration of the order-a tensor
    PartialTerm t[kPartialMaxTerms];
};

// doubles per configuration of the order-a tensor
RTB_HD int64_t partial_size(int n, int a)
{
    int64_t s = 6 * n;
    for (int i = 1; i < a; ++i) s *= n;
    return s;
}

// Expands the product rule for order c (ETS.py:1888-1927 `add_indices` / `add_pdi`): every term of order
// c-1 splits in two, the new derivative index going to the first or to the second factor.
inline void partial_plan(int n, int c, PartialPlan *p)
{
    p->n = n; p->c = c;
    p->nterms = 1;
    p->t[0].oa = 1; p->t[0].ob = 1;
    p->t[0].ia[0] = 1; p->t[0].ib[0] = 0;              // Hessian: H[k, :, j] = J_w[:, k] x J[:, j], digits (j, k)
    for (int order = 3; order <= c; ++order) {
        PartialTerm next[kPartialMaxTerms];
        for (int i = 0; i < p->nterms; ++i) {
            PartialTerm first = p->t[i], second = p->t[i];
            first.ia[first.oa++] = (int8_t)(order - 1);
            second.ib[second.ob++] = (int8_t)(order - 1);
            next[2 * i] = first;
            next[2 * i + 1] = second;
        }
        p->nterms *= 2;
        for (int i = 0; i < p->nterms; ++i) p->t[i] = next[i];
    }
    for (int k = 0; k < p->nterms; ++k) {
        PartialTerm &t = p->t[k];
        for (int i = 0; i < kPartialMaxOrder; ++i) t.ca[i] = t.cb[i] = 0;
        int32_t w = 6 * n;
        t.ca[t.ia[0]] = 1;
        for (int u = 1; u < t.oa; ++u, w *= n) t.ca[t.ia[u]] = w;
        w = 6 * n;
        t.cb[t.ib[0]] = 1;
        for (int u = 1; u < t.ob; ++u, w *= n) t.cb[t.ib[u]] = w;
    }
    p->cols = 1;
    for (int i = 0; i < c; ++i) p->cols *= n;
    p->size[0] = 0;
    for (int a = 1; a <= kPartialMaxOrder; ++a) p->size[a] = partial_size(n, a);
}

// Index arithmetic at full VALU rate: 32-bit integer multiplies and the mul_hi of a division by a run-time n issue at
// quarter rate on CDNA and were measured to dominate this kernel; every quantity here is below 2^24 (the launcher
// checks), so 24-bit multiply-adds and a float reciprocal with a one-step correction are exact.
RTB_HD uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b) + c;
#else
    return a * b + c;
#endif
}
// r = q * n + rem for r < 2^24, inv = 1.0f / n
RTB_HD uint32_t divmod24(uint32_t r, uint32_t n, float inv, uint32_t *rem)
{
    int32_t q = (int32_t)((float)r * inv);
    int32_t m = (int32_t)r - (int32_t)mad24((uint32_t)q, n, 0);
    if (m < 0) { --q; m += (int32_t)n; }
    else if (m >= (int32_t)n) { ++q; m -= (int32_t)n; }
    *rem = (uint32_t)m;
    return (uint32_t)q;
}

// One column (cfg, col) of the order-c tensor.  src(a, cfg, off) = element `off` of configuration cfg's order-a tensor.
// `put(r, v)` receives the six rows of the column.
template <int C, class Src, class Put>
RTB_HD void partial_column(const PartialPlan &p, const Src &src, int64_t cfg, uint32_t col, Put put)
{
    const int n = p.n;
    const float inv = 1.0f / (float)n;
    uint32_t digit[C];
    uint32_t r = col;
#pragma unroll
    for (int i = 0; i < C; ++i) r = divmod24(r, (uint32_t)n, inv, &digit[i]);
    double trn[3] = {0, 0, 0}, rot[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < (1 << (C - 2)); ++k) {
        const PartialTerm &t = p.t[k];
        uint32_t ua = 0, ub = 0;
#pragma unroll
        for (int i = 0; i < C; ++i) { ua = mad24(digit[i], (uint32_t)t.ca[i], ua); ub = mad24(digit[i], (uint32_t)t.cb[i], ub); }
        const int offa = (int)ua, offb = (int)ub;
        const int oa = t.oa, ob = t.ob;
        const double w0 = src(oa, cfg, offa + 3 * n), w1 = src(oa, cfg, offa + 4 * n), w2 = src(oa, cfg, offa + 5 * n);
        const double v0 = src(ob, cfg, offb), v1 = src(ob, cfg, offb + n), v2 = src(ob, cfg, offb + 2 * n);
        const double u0 = src(ob, cfg, offb + 3 * n), u1 = src(ob, cfg, offb + 4 * n), u2 = src(ob, cfg, offb + 5 * n);
        trn[0] += w1 * v2 - w2 * v1; trn[1] += w2 * v0 - w0 * v2; trn[2] += w0 * v1 - w1 * v0;
        rot[0] += w1 * u2 - w2 * u1; rot[1] += w2 * u0 - w0 * u2; rot[2] += w0 * u1 - w1 * u0;
    }
    put(0, trn[0]); put(1, trn[1]); put(2, trn[2]);
    put(3, rot[0]); put(4, rot[1]); put(5, rot[2]);
}

// Order 3, written out (the plan of partial_plan(n, 3): two terms, in this order, so the sums round as partial_column<3>'s):
//     dT3[d2, d1, :, d0] = rot(H[d2][:, d1]) x J[:, d0]  +  rot(J[:, d1]) x H[d2][:, d0]
// jac(off) / hes(off): element `off` of this configuration's (6, n) Jacobian / (n, 6, n) Hessian.  No plan look-ups, no offset
// arithmetic beyond three multiply-adds: the digit -> offset linear forms of the general routine are constants here.
template <class Jac, class Hes, class Put>
RTB_HD void partial3_column(int n, const Jac &jac, const Hes &hes, uint32_t d0, uint32_t d1, uint32_t d2, Put put)
{
    const uint32_t hs = mad24(d2, 6u * (uint32_t)n, 0);             // slice d2 of the Hessian
    const uint32_t a1 = hs + d1, b2 = hs + d0;
    double trn[3], rot[3];
    {
        const double w0 = hes(a1 + 3 * n), w1 = hes(a1 + 4 * n), w2 = hes(a1 + 5 * n);
        const double v0 = jac(d0), v1 = jac(d0 + n), v2 = jac(d0 + 2 * n);
        const double u0 = jac(d0 + 3 * n), u1 = jac(d0 + 4 * n), u2 = jac(d0 + 5 * n);
        trn[0] = 0.0 + (w1 * v2 - w2 * v1); trn[1] = 0.0 + (w2 * v0 - w0 * v2); trn[2] = 0.0 + (w0 * v1 - w1 * v0);
        rot[0] = 0.0 + (w1 * u2 - w2 * u1); rot[1] = 0.0 + (w2 * u0 - w0 * u2); rot[2] = 0.0 + (w0 * u1 - w1 * u0);
    }
    {
        const double w0 = jac(d1 + 3 * n), w1 = jac(d1 + 4 * n), w2 = jac(d1 + 5 * n);
        const double v0 = hes(b2), v1 = hes(b2 + n), v2 = hes(b2 + 2 * n);
        const double u0 = hes(b2 + 3 * n), u1 = hes(b2 + 4 * n), u2 = hes(b2 + 5 * n);
        trn[0] += w1 * v2 - w2 * v1; trn[1] += w2 * v0 - w0 * v2; trn[2] += w0 * v1 - w1 * v0;
        rot[0] += w1 * u2 - w2 * u1; rot[1] += w2 * u0 - w0 * u2; rot[2] += w0 * u1 - w1 * u0;
    }
    put(0, trn[0]); put(1, trn[1]); put(2, trn[2]);
    put(3, rot[0]); put(4, rot[1]); put(5, rot[2]);
}

// How the order-3 kernel cuts the batch: G whole configurations per workgroup (so every Jacobian and Hessian is staged exactly
// once), U columns per lane.  G is the most whose output run (48 n^3 bytes each) fits the tile budget; 0 = not served (the general
// kernel takes over).
constexpr int kPartial3TileBytes = 36 * 1024;
constexpr int kPartial3MaxU = 4;
RTB_HD void partial3_geometry(int n, int block, int *G, int *U)
{
    const int cols = n * n * n;
    int g = kPartial3TileBytes / (48 * cols);
    const int cap = block * kPartial3MaxU / cols;                   // columns the lanes can hold
    if (g > cap) g = cap;
    *G = g < 1 ? 0 : g;
    *U = g < 1 ? 0 : (g * cols + block - 1) / block;
}

}  // namespace rtbhip
