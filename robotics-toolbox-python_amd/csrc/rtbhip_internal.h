// rtbhip_internal.h -- shared declarations of librtbhip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/rtbhip.h"

namespace rtbhip {

// ---------------------------------------------------------------- device chain program
// The host "chain compiler" (chain.cpp) lowers the user's elementary-transform list to this
// program.  It is wave-uniform data: kernels read it through the scalar cache (s_load), so the
// constants arrive in SGPRs and cost no VGPRs and no LDS bandwidth.
enum DevKind : int32_t {
    K_JRX = 0, K_JRY = 1, K_JRZ = 2,   // variable rotation about x/y/z
    K_JTX = 3, K_JTY = 4, K_JTZ = 5,   // variable translation along x/y/z
    K_CRX = 6, K_CRY = 7, K_CRZ = 8,   // constant rotation about x/y/z  p = {c, s}
    K_CTX = 9, K_CTY = 10, K_CTZ = 11, // constant translation along one axis p = {d}
    K_CT3 = 12,                        // constant translation p = {x,y,z}
    K_CGEN = 13,                       // general constant p = {R row-major (9), t (3)}
};

struct alignas(16) DevOp {
    int32_t kind;
    int32_t jq;    // q column read by a joint op
    int32_t jcol;  // Jacobian column (order of joints in the chain)
    int32_t flip;
    double p[12];
};
static_assert(sizeof(DevOp) == 112, "DevOp layout");

struct DevChainHeader {
    int32_t m;        // number of device ops
    int32_t n;        // joints
    int32_t q_width;  // columns of q
    int32_t pad;
};

// Host-side chain object behind an rtbhip_chain_t handle.
struct Chain {
    std::vector<rtbhip_et> ets;   // as given (for chain_info / debugging)
    std::vector<DevOp> ops;       // compiled program
    std::vector<double> qlim;     // 2*n (lows, highs)
    int n = 0, q_width = 0;
    std::map<int, DevOp *> dev_ops;    // per-device upload of ops
    std::map<int, double *> dev_qlim;  // per-device upload of qlim
    std::mutex mu;
};

// Host-side dynamics object behind an rtbhip_dyn_t handle.
struct alignas(16) DevLink {   // per-link constants for the Newton-Euler kernel (24 doubles, reordered)
    double sa, ca;             // sin/cos(alpha), evaluated once on the host with libm
    double a, d, theta, offset;
    double m, rx, ry, rz;
    double I[9];               // as given (DHRobot.py:1353), read column-major like vmath.c
    double Jm, G, B, Tc0, Tc1;
    int32_t sigma;             // 0 revolute, 1 prismatic
    int32_t pad;
};
struct Dyn {
    std::vector<DevLink> links;
    int n = 0, mdh = 0;
    std::map<int, DevLink *> dev_links;
    std::mutex mu;
};

// ---------------------------------------------------------------- error plumbing
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);
#define RTB_HIP(call)                                            \
    do {                                                         \
        hipError_t _e = (call);                                  \
        if (_e != hipSuccess) return ::rtbhip::hip_fail(_e, #call); \
    } while (0)

Chain *chain_from_handle(rtbhip_chain_t h);
Dyn *dyn_from_handle(rtbhip_dyn_t h);
int chain_device_ops(Chain *c, const DevOp **out, const double **qlim_out);
int dyn_device_links(Dyn *d, const DevLink **out);
int compile_chain(const rtbhip_et *ets, int m, const double *qlim, Chain *out);
void note_launch(int grid, int block, int lds);
int device_cu_count(int *cus);

// ---------------------------------------------------------------- kernel launchers (device pointers)
struct Affine { double v[12]; int used; };  // row-major 3x4, host-side small parameter

int launch_kin(const Chain *c, const DevOp *ops, const double *q, int64_t N, const Affine &base,
               const Affine &tool, int frame, double *T, double *J, double *H, hipStream_t s);

struct FleetEntry {   // device-visible descriptor of one chain of a fleet launch
    const DevOp *ops;
    const double *q;
    double *T;
    double *J;
    int64_t N;
    int64_t tile0;    // first global tile index of this chain
    int32_t m, n, q_width, stride;
};
int launch_fleet(const std::vector<FleetEntry> &entries, int frame, hipStream_t s);

int launch_rne(const Dyn *d, const DevLink *links, const double *q, const double *qd,
               const double *qdd, int64_t N, const double *grav3, const double *fext6, double *tau,
               hipStream_t s);

struct IkParams {
    int ilimit, slimit, reject_jl, method, flavour;
    double tol, lambda;
    double we[6];
    uint64_t seed;
};
int launch_ik(const Chain *c, const DevOp *ops, const double *qlim, const double *Tep, int64_t N,
              const double *q0, const IkParams &p, double *q_out, int32_t *success, int32_t *iters,
              int32_t *searches, double *residual, hipStream_t s);
void ik_restart_host(const Chain *c, uint64_t seed, int64_t target, int search, double *q_n);

}  // namespace rtbhip
