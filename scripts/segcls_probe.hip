// scripts/segcls_probe.hip -- on the DEVICE: every structured form of P * C (kin_device.h: pose_mul_seg_sig<class, mask>) against the general product
// on segments of that class; prints the largest difference per (class, mask).  Bit-identity is the claim (0 everywhere).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude scripts/segcls_probe.hip -o build/probe/segcls_probe && build/probe/segcls_probe
#include "../robotics-toolbox-python_amd/csrc/kin_reg.h"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace rtbhip;
struct CV { const DevSeg *seg; };
template <int CLS, int TM>
__global__ void k(CV cv, int j, const double *Pin, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pose A, B;
    const double *p = Pin + 12 * i;
    A.r00 = p[0]; A.r01 = p[1]; A.r02 = p[2]; A.r10 = p[3]; A.r11 = p[4]; A.r12 = p[5]; A.r20 = p[6]; A.r21 = p[7]; A.r22 = p[8]; A.tx = p[9]; A.ty = p[10]; A.tz = p[11];
    B = A;
    pose_mul_seg_sig<CLS, TM>(A, cv, j);
    pose_mul_seg<true>(B, cv, j);
    double d = 0;
    d = fmax(d, fabs(A.r00 - B.r00)); d = fmax(d, fabs(A.r01 - B.r01)); d = fmax(d, fabs(A.r02 - B.r02));
    d = fmax(d, fabs(A.r10 - B.r10)); d = fmax(d, fabs(A.r11 - B.r11)); d = fmax(d, fabs(A.r12 - B.r12));
    d = fmax(d, fabs(A.r20 - B.r20)); d = fmax(d, fabs(A.r21 - B.r21)); d = fmax(d, fabs(A.r22 - B.r22));
    d = fmax(d, fabs(A.tx - B.tx)); d = fmax(d, fabs(A.ty - B.ty)); d = fmax(d, fabs(A.tz - B.tz));
    out[i] = d;
}
// the whole FK + Jacobian walk of k_ik (reg_core) on the UR5 chain: signature form against general form
struct CVT { const DevSeg *seg; const int32_t *jmeta; const double *trig; };
constexpr SegSig kSigUR = kSegSigPresent | seg_sig_of(0, kSegIdentity, 4) | seg_sig_of(1, kSegGeneral, 2) | seg_sig_of(2, kSegIdentity, 5) | seg_sig_of(3, kSegRzP, 1) |
                          seg_sig_of(4, kSegPermA, 4) | seg_sig_of(5, kSegPermB, 4) | seg_sig_of(6, kSegGeneral, 4);
__constant__ double kTab[kSincosTableLen] = RTB_SINCOS_TABLE_INIT;
template <SegSig SIG>
__global__ void k_walk_sig(CVT cv, const double *q, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cv.trig = kTab;
    double qv[6];
    for (int j = 0; j < 6; ++j) qv[j] = q[6 * i + j];
    Pose A, B;
    double ja[36], jb[36];
    reg_core<6, true, true, SIG>(cv, &cv.seg[6].r[0], 0, qv, A, ja);
    reg_core<6, true, true, 0>(cv, &cv.seg[6].r[0], 0, qv, B, jb);
    double d = 0;
    d = fmax(d, fabs(A.r00 - B.r00)); d = fmax(d, fabs(A.r01 - B.r01)); d = fmax(d, fabs(A.r02 - B.r02));
    d = fmax(d, fabs(A.r10 - B.r10)); d = fmax(d, fabs(A.r11 - B.r11)); d = fmax(d, fabs(A.r12 - B.r12));
    d = fmax(d, fabs(A.r20 - B.r20)); d = fmax(d, fabs(A.r21 - B.r21)); d = fmax(d, fabs(A.r22 - B.r22));
    d = fmax(d, fabs(A.tx - B.tx)); d = fmax(d, fabs(A.ty - B.ty)); d = fmax(d, fabs(A.tz - B.tz));
    out[2 * i] = d; out[2 * i + 1] = 0;
}
constexpr SegSig kAllGeneral = kSegSigPresent;      // every segment class 0 (general)
__global__ void k_walk(CVT cv, const double *q, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cv.trig = kTab;
    double qv[6];
    for (int j = 0; j < 6; ++j) qv[j] = q[6 * i + j];
    Pose A, B;
    double ja[36], jb[36];
    reg_core<6, true, true, kSigUR>(cv, &cv.seg[6].r[0], 0, qv, A, ja);
    reg_core<6, true, true, 0>(cv, &cv.seg[6].r[0], 0, qv, B, jb);
    double d = 0, dj = 0;
    d = fmax(d, fabs(A.r00 - B.r00)); d = fmax(d, fabs(A.r01 - B.r01)); d = fmax(d, fabs(A.r02 - B.r02));
    d = fmax(d, fabs(A.r10 - B.r10)); d = fmax(d, fabs(A.r11 - B.r11)); d = fmax(d, fabs(A.r12 - B.r12));
    d = fmax(d, fabs(A.r20 - B.r20)); d = fmax(d, fabs(A.r21 - B.r21)); d = fmax(d, fabs(A.r22 - B.r22));
    d = fmax(d, fabs(A.tx - B.tx)); d = fmax(d, fabs(A.ty - B.ty)); d = fmax(d, fabs(A.tz - B.tz));
    for (int k = 0; k < 36; ++k) dj = fmax(dj, fabs(ja[k] - jb[k]));
    out[2 * i] = d; out[2 * i + 1] = dj;
}
static DevSeg make(int cls, int tm)
{
    DevSeg s;
    const double e = 6.123233995736766e-17, ca = cos(0.3), sa = sin(0.3);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    auto set = [&](std::initializer_list<double> v) { int k = 0; for (double x : v) R[k++] = x; };
    switch (cls) {
    case kSegGeneral: set({0.36, 0.48, -0.8, -0.8, 0.6, 0.0, 0.48, 0.64, 0.6}); break;
    case kSegRxP: set({1, 0, 0, 0, e, -1, 0, 1, e}); break;
    case kSegRxN: set({1, 0, 0, 0, e, 1, 0, -1, e}); break;
    case kSegRx: set({1, 0, 0, 0, ca, -sa, 0, sa, ca}); break;
    case kSegRyP: set({e, 0, -1, 0, 1, 0, 1, 0, e}); break;
    case kSegRyN: set({e, 0, 1, 0, 1, 0, -1, 0, e}); break;
    case kSegRy: set({ca, 0, sa, 0, 1, 0, -sa, 0, ca}); break;
    case kSegRzP: set({e, -1, 0, 1, e, 0, 0, 0, 1}); break;
    case kSegRzN: set({e, 1, 0, -1, e, 0, 0, 0, 1}); break;
    case kSegRz: set({ca, -sa, 0, sa, ca, 0, 0, 0, 1}); break;
    case kSegPermA: set({0, 0, 1, 1, 0, 0, 0, 1, 0}); break;
    case kSegPermB: set({0, 1, 0, 0, 0, 1, 1, 0, 0}); break;
    default: break;
    }
    for (int k = 0; k < 9; ++k) s.r[k] = R[k];
    s.t[0] = (tm & 1) ? 0.0825 : 0.0; s.t[1] = (tm & 2) ? -0.316 : 0.0; s.t[2] = (tm & 4) ? 1.9e-17 + 0.107 : 0.0;
    return s;
}
template <int CLS, int TM>
static void run(const double *dP, double *dout, std::vector<double> &host, int n, DevSeg *dseg)
{
    DevSeg s = make(CLS, TM);
    hipMemcpy(dseg, &s, sizeof s, hipMemcpyHostToDevice);
    CV cv{dseg};
    hipLaunchKernelGGL((k<CLS, TM>), dim3((n + 255) / 256), dim3(256), 0, 0, cv, 0, dP, dout, n);
    hipMemcpy(host.data(), dout, n * sizeof(double), hipMemcpyDeviceToHost);
    double m = 0; int bad = 0;
    for (double x : host) { m = fmax(m, x); bad += x != 0.0; }
    printf("class %2d mask %d  max |structured - general| = %.3e  (%d of %d poses differ)\n", CLS, TM, m, bad, n);
}
template <int CLS> static void run_cls(const double *dP, double *dout, std::vector<double> &h, int n, DevSeg *dseg)
{
    run<CLS, 0>(dP, dout, h, n, dseg); run<CLS, 1>(dP, dout, h, n, dseg); run<CLS, 5>(dP, dout, h, n, dseg); run<CLS, 7>(dP, dout, h, n, dseg);
}
int main()
{
    const int n = 100000;
    std::vector<double> P(12 * n), h(n);
    unsigned long long z = 88172645463325252ull;
    auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (double)(z >> 11) / 9007199254740992.0 * 2 - 1; };
    for (int i = 0; i < n; ++i) {                 // random rotations (Gram-Schmidt) and translations
        double a[3] = {rnd(), rnd(), rnd()}, b[3] = {rnd(), rnd(), rnd()};
        double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        for (double &x : a) x /= na;
        double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        for (int k = 0; k < 3; ++k) b[k] -= d * a[k];
        double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        for (double &x : b) x /= nb;
        double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        double *p = &P[12 * i];
        p[0] = a[0]; p[1] = b[0]; p[2] = c[0]; p[3] = a[1]; p[4] = b[1]; p[5] = c[1]; p[6] = a[2]; p[7] = b[2]; p[8] = c[2];
        p[9] = rnd(); p[10] = rnd(); p[11] = rnd();
    }
    double *dP, *dout; DevSeg *dseg;
    hipMalloc(&dP, P.size() * 8); hipMalloc(&dout, n * 8); hipMalloc(&dseg, sizeof(DevSeg));
    hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    run_cls<1>(dP, dout, h, n, dseg); run_cls<2>(dP, dout, h, n, dseg); run_cls<3>(dP, dout, h, n, dseg); run_cls<4>(dP, dout, h, n, dseg);
    run_cls<5>(dP, dout, h, n, dseg); run_cls<6>(dP, dout, h, n, dseg); run_cls<7>(dP, dout, h, n, dseg); run_cls<8>(dP, dout, h, n, dseg);
    run_cls<9>(dP, dout, h, n, dseg); run_cls<10>(dP, dout, h, n, dseg); run_cls<11>(dP, dout, h, n, dseg); run_cls<12>(dP, dout, h, n, dseg);
    // the UR5's own folded segments (scripts/ur5_segs.bin, written by tests/emu: emu_chain_segments), each in the form its signature names
    if (FILE *f = fopen("scripts/ur5_segs.bin", "rb")) {
        double seg[7][12];
        if (fread(seg, sizeof seg, 1, f) == 1) {
            DevSeg hs[7];
            for (int j = 0; j < 7; ++j) { for (int k = 0; k < 9; ++k) hs[j].r[k] = seg[j][k]; for (int k = 0; k < 3; ++k) hs[j].t[k] = seg[j][9 + k]; }
            DevSeg *d7; hipMalloc(&d7, sizeof hs); hipMemcpy(d7, hs, sizeof hs, hipMemcpyHostToDevice);
            CV cv{d7};
            auto go = [&](auto kern, int j, const char *what) {
                hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, 0, cv, j, dP, dout, n);
                hipMemcpy(h.data(), dout, n * sizeof(double), hipMemcpyDeviceToHost);
                double m = 0; int bad = 0;
                for (double x : h) { m = fmax(m, x); bad += x != 0.0; }
                printf("UR5 segment %d as %s: max diff %.3e (%d poses differ)\n", j, what, m, bad);
            };
            {
                std::vector<double> q(6 * n), o(2 * n);
                for (double &x : q) x = 3.0 * rnd();
                double *dq, *dw; int32_t *djm; int32_t jm[7] = {0, 0, 0, 0, 0, 0, 0};
                hipMalloc(&dq, q.size() * 8); hipMalloc(&dw, o.size() * 8); hipMalloc(&djm, sizeof jm);
                hipMemcpy(dq, q.data(), q.size() * 8, hipMemcpyHostToDevice); hipMemcpy(djm, jm, sizeof jm, hipMemcpyHostToDevice);
                CVT cvt{d7, djm, nullptr};
                hipLaunchKernelGGL(k_walk, dim3((n + 255) / 256), dim3(256), 0, 0, cvt, dq, dw, n);
                hipMemcpy(o.data(), dw, o.size() * 8, hipMemcpyDeviceToHost);
                double mp = 0, mj = 0; int bad = 0;
                for (int i = 0; i < n; ++i) { mp = fmax(mp, o[2 * i]); mj = fmax(mj, o[2 * i + 1]); bad += (o[2 * i] != 0.0 || o[2 * i + 1] != 0.0); }
                printf("UR5 whole walk (reg_core<6,true,true,SIG> vs <..,0>): max pose diff %.3e, max Jacobian diff %.3e (%d of %d configurations differ)\n", mp, mj, bad, n);
                auto bis = [&](auto kern, const char *what) {
                    hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, 0, cvt, dq, dw, n);
                    hipMemcpy(o.data(), dw, o.size() * 8, hipMemcpyDeviceToHost);
                    double mp = 0; int bad = 0;
                    for (int i = 0; i < n; ++i) { mp = fmax(mp, o[2 * i]); bad += o[2 * i] != 0.0; }
                    printf("  walk with only %s structured: max pose diff %.3e (%d differ)\n", what, mp, bad);
                };
                bis(k_walk_sig<kAllGeneral>, "nothing (all general through the signature path)");
                bis(k_walk_sig<(kSegSigPresent | seg_sig_of(0, kSegIdentity, 4))>, "segment 0 identity+z");
                bis(k_walk_sig<(kSegSigPresent | seg_sig_of(2, kSegIdentity, 5))>, "segment 2 identity+xz");
                bis(k_walk_sig<(kSegSigPresent | seg_sig_of(3, kSegRzP, 1))>, "segment 3 RzP+x");
                bis(k_walk_sig<(kSegSigPresent | seg_sig_of(4, kSegPermA, 4))>, "segment 4 permA+z");
                bis(k_walk_sig<(kSegSigPresent | seg_sig_of(5, kSegPermB, 4))>, "segment 5 permB+z");
            }
            {
            go(k<1, 4>, 0, "identity+z"); go(k<0, 2>, 1, "general"); go(k<1, 5>, 2, "identity+xz"); go(k<8, 1>, 3, "RzP+x"); go(k<11, 4>, 4, "permA+z"); go(k<12, 4>, 5, "permB+z"); go(k<0, 4>, 6, "general");
            }
        }
        fclose(f);
    }
    return 0;
}
