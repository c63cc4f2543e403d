// rtbhip_internal.h -- shared declarations of librtbhip.so (not part of the public ABI).
//
// Two readers: hipcc building the library, and hipRTC compiling ONE kernel instantiation for a robot's structure signature at run time (jit.cpp).
// hipRTC has the HIP device headers built in and no host standard library: under __HIPCC_RTC__ only the device-visible tables and constants
// of this file exist.  Not a second backend -- the same sources, the same gfx950 compiler, a different moment.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/rtbhip.h"
#define RTB_HOST_SIDE 1
#else
#define RTB_HOST_SIDE 0
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#define RTBHIP_MAX_JOINTS 32       /* include/rtbhip.h (static_assert'ed against it in jit.cpp) */
#endif

namespace rtbhip {

// ---------------------------------------------------------------- device chain program
// The host "chain compiler" (chain.cpp) lowers the user's elementary-transform list to the
// CANONICAL SEGMENT FORM
//        T(q) = C_0 * Z_0(q) * C_1 * Z_1(q) * ... * C_{n-1} * Z_{n-1}(q) * C_n
// where every C_j is one constant affine (all constant ETs between two joints folded together, the
// reference's ETS.compile() robot/ETS.py:857-906) and every joint acts about/along the local z axis
// (an Rx/Ry/tx/ty joint is conjugated by an exact axis permutation that is absorbed into the
// neighbouring constants).  The device code for a chain is then branch-free and identical for every
// robot: "multiply by a constant 3x4, rotate two columns / slide along one" n times.
// It is wave-uniform data: kernels read it through the scalar cache (s_load), so the constants
// arrive in SGPRs and cost no VGPRs and no LDS bandwidth.
struct DevSeg {
    double r[9];  // rotation, row-major
    double t[3];
};
static_assert(sizeof(DevSeg) == 96, "DevSeg layout");

// per joint, in chain order: bit 0 = prismatic, bits 8..15 = q column (jindex), bit 16 = flip
__host__ __device__ inline int jm_prismatic(int jm) { return jm & 1; }
__host__ __device__ inline int jm_jq(int jm) { return (jm >> 8) & 0xff; }
__host__ __device__ inline int jm_flip(int jm) { return (jm >> 16) & 1; }
// bits 20..23: STRUCTURE CLASS of the constant segment C_j that precedes joint j, bits 24..26: which components of its translation are non-zero
// (descriptor n -- one past the joints -- carries the class of the tail C_n alone).  Robot constants are mostly not general affines: a pure
// translation (URDF origins with rpy = 0), a quarter turn about one axis times a translation (DH alpha = +-pi/2; models/ETS/Panda.py:32-54:
// six of the Panda's seven inner constants), a rotation about one axis (the Panda's Rz(-pi/4) flange).  The class is decided from the EXACT
// zeros and ones of the folded matrix (chain.cpp: seg_class) -- cos(pi/2) = 6.1e-17 stays what it is, nothing is snapped -- and lets the
// kernels that opt in (k_ik) replace the 27 + 9 fused multiply-adds of a general P * C by 0 / 6 / 12 + 3 per non-zero translation component.
constexpr int kSegGeneral = 0, kSegIdentity = 1,
              kSegRxP = 2, kSegRxN = 3, kSegRx = 4,      // R = [1 0 0; 0 a b; 0 c d]: P (b, c) = (-1, +1), N (b, c) = (+1, -1), else any a b c d
              kSegRyP = 5, kSegRyN = 6, kSegRy = 7,      // R = [a 0 b; 0 1 0; c 0 d]
              kSegRzP = 8, kSegRzN = 9, kSegRz = 10,     // R = [a b 0; c d 0; 0 0 1]
              kSegPermA = 11, kSegPermB = 12;            // the cyclic column permutations of the axis conjugation (chain.cpp: axis_perm): new columns (1,2,0) / (2,0,1)
__host__ __device__ inline int jm_cls(int jm) { return (jm >> 20) & 15; }
__host__ __device__ inline int jm_tmask(int jm) { return (jm >> 24) & 7; }
// STRUCTURE SIGNATURE of a chain (kin_reg.h) or a dynamics tree (tree_device.h): 7 bits per constant (class | translation mask << 4), bit 63 = present
typedef unsigned long long SegSig;
typedef unsigned __int128 TreeTopo;      // tree_device.h
constexpr SegSig kSegSigPresent = 1ull << 63;
__host__ __device__ constexpr int seg_sig_cls(SegSig s, int j) { return (int)((s >> (7 * j)) & 15u); }
__host__ __device__ constexpr int seg_sig_tm(SegSig s, int j) { return (int)((s >> (7 * j + 4)) & 7u); }
constexpr SegSig seg_sig_of(int j, int cls, int tm) { return (SegSig)((cls & 15) | ((tm & 7) << 4)) << (7 * j); }
#if RTB_HOST_SIDE
int seg_class_bits(const DevSeg &a);      // chain.cpp: (class << 20) | (translation mask << 24) of a folded constant
#endif

// What a kernel receives: two wave-uniform tables in one device allocation.
struct DevChain {
    const DevSeg *seg;     // n + 1 constants C_0 .. C_n
    const int32_t *jmeta;  // n joint descriptors (+ one more word: the structure class of the tail C_n)
};

#if RTB_HOST_SIDE
// ---------------------------------------------------------------- run-time instantiation (jit.cpp)
bool jit_enabled();                                                       // rtbhip_tune("jit") != 0
bool jit_builtin_enabled();                                               // rtbhip_tune("sig_builtin") != 0: built-in structure instantiations are used
void jit_request(const char *unit, const std::string &expr, const std::string &preamble = std::string(), bool touch_device = false);              // at *_create: ask for it, nobody waits
hipFunction_t jit_function(const char *unit, const std::string &expr, const std::string &preamble = std::string());    // at a launch: the function on the current device, or NULL (not ready / failed / off)
hipFunction_t jit_function_wait(const char *unit, const std::string &expr, const std::string &preamble = std::string());   // sizes with no built-in kernel: waits; NULL = error set
int jit_launch(hipFunction_t f, dim3 grid, dim3 block, size_t lds, hipStream_t s, void **args);
std::string jit_hex(unsigned long long v);                                // "0x...ull"
void jit_tune(const char *key, int value);
int jit_wait(double timeout_s);
void jit_stats(rtbhip_jit_info *out);
int jit_compile_now(const char *unit, const char *expr, const char *arch, size_t *code_bytes, double *seconds, int *from_disk, const char *preamble = nullptr);
// per-handle memo of the functions a handle's launches resolved (key: device << 8 | variant), so a hot loop does not build name expressions
struct JitMemo {
    std::mutex mu;
    std::map<uint64_t, hipFunction_t> fn;
    template <class MakeExpr> hipFunction_t get(const char *unit, int variant, MakeExpr make, const std::string &preamble = std::string())
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        const uint64_t k = ((uint64_t)dev << 8) | (uint64_t)(variant & 0xff);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = fn.find(k);
            if (it != fn.end()) return it->second;
        }
        hipFunction_t f = jit_function(unit, make(), preamble);
        if (f) { std::lock_guard<std::mutex> lk(mu); fn[k] = f; }
        return f;
    }
    template <class MakeExpr> hipFunction_t get_wait(const char *unit, int variant, MakeExpr make, const std::string &preamble = std::string())
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        const uint64_t k = ((uint64_t)dev << 8) | (uint64_t)(variant & 0xff);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = fn.find(k);
            if (it != fn.end()) return it->second;
        }
        hipFunction_t f = jit_function_wait(unit, make(), preamble);
        if (f) { std::lock_guard<std::mutex> lk(mu); fn[k] = f; }
        return f;
    }
};
// the name expressions of a handle's run-time instantiations (empty: a built-in instantiation serves it, or no signature applies)
struct Chain; struct Dyn; struct Tree;
std::vector<std::string> ik_jit_names(const Chain *c);
std::vector<std::string> rne_jit_names(const Dyn *d);
std::vector<std::string> tree_jit_names(const Tree *t);

// Host-side chain object behind an rtbhip_chain_t handle.
struct Chain {
    std::vector<rtbhip_et> ets;   // as given (for chain_info / debugging)
    std::vector<DevSeg> seg;      // n + 1
    std::vector<int32_t> jmeta;   // n + 1: joint descriptors, then the tail's class word
    std::vector<double> qlim;     // 2*n (lows, highs)
    int n = 0, q_width = 0;
    std::map<int, void *> dev_ops;     // per-device upload: [seg | jmeta]
    std::map<int, double *> dev_qlim;  // per-device upload of qlim
    mutable JitMemo jit;
    std::mutex mu;
    ~Chain();                          // frees the per-device uploads (runs when the last user lets go, see *_from_handle)
};

#endif

// Host-side dynamics object behind an rtbhip_dyn_t handle.
struct alignas(16) DevLink {   // per-link constants for the Newton-Euler kernel (24 doubles, reordered)
    double sa, ca;             // sin/cos(alpha), evaluated once on the host with libm
    double a, d, theta, offset;
    double m, rx, ry, rz;
    double I[9];               // as given (DHRobot.py:1353), read column-major like vmath.c
    double Jm, G, B, Tc0, Tc1;
    double gjm, gb, ag;        // (G*G)*Jm, (G*G)*B, |G|: the products ne.c:464-492 forms per call, rounded in the same order
    int32_t sigma;             // 0 revolute, 1 prismatic
    int32_t flags;             // wave-uniform shortcuts: kLinkRZero (centre of mass at the link origin), kLinkIDiag (diagonal inertia), kLinkPsZero (a = d = 0)
};
constexpr int kLinkRZero = 1, kLinkIDiag = 2, kLinkPsZero = 4;   // kLinkPsZero: revolute link with a = d = 0 (its frame origin coincides with its predecessor's)
#if RTB_HOST_SIDE
struct Dyn {
    std::vector<DevLink> links;
    int n = 0, mdh = 0;
    std::map<int, DevLink *> dev_links;
    mutable JitMemo jit;
    std::mutex mu;
    ~Dyn();
};

// Host-side dynamics tree behind an rtbhip_tree_t handle (tree.cpp, tree_device.h).
struct DevGroup;
struct Tree {
    std::vector<DevGroup> groups;
    int n = 0, nslots = 0;
    SegSig sig = 0, sig2 = 0;          // structure signature of the group constants (tree_device.h): groups 0 .. 7 and 8 .. 15; 0 beyond 16 groups
    TreeTopo topo = 0;                 // the tree's bookkeeping as one word (tree_device.h: TreeTopo), 0 where it does not apply
    std::map<int, DevGroup *> dev_groups;
    mutable JitMemo jit;
    std::mutex mu;
    ~Tree();
};
int compile_tree(const rtbhip_tree_group *groups, int ng, Tree *out);
SegSig tree_signature(const DevGroup *groups, int ng, SegSig *sig2 = nullptr);
TreeTopo tree_topology(const DevGroup *groups, int ng, int nslots);
std::shared_ptr<Tree> tree_from_handle(rtbhip_tree_t h);
int tree_device_groups(Tree *t, const DevGroup **out);
int launch_tree_rne(const Tree *t, const DevGroup *groups, const double *q, const double *qd, const double *qdd, int64_t N,
                    const double *grav3, double *tau, hipStream_t s);
int launch_stream_probe(const double *src, int64_t read_doubles, double *dst, int64_t write_doubles, hipStream_t s);
int launch_tree_dyn(const Tree *t, const DevGroup *groups, int mode /* 0 inertia, 1 coriolis, 2 accel */, const double *q, const double *qd,
                    const double *tq, int64_t N, const double *grav3, double *out, hipStream_t s);

// ---------------------------------------------------------------- error plumbing
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);
#define RTB_HIP(call)                                            \
    do {                                                         \
        hipError_t _e = (call);                                  \
        if (_e != hipSuccess) return ::rtbhip::hip_fail(_e, #call); \
    } while (0)

// Handle lookups hand out SHARED ownership: a destroy racing with a launch on another thread only drops the registry's
// reference, the object (and its device tables) lives until the last call using it returns.
std::shared_ptr<Chain> chain_from_handle(rtbhip_chain_t h);
std::shared_ptr<Dyn> dyn_from_handle(rtbhip_dyn_t h);
int chain_device_ops(Chain *c, DevChain *out, const double **qlim_out);
DevChain chain_host_view(const Chain *c);
int dyn_device_links(Dyn *d, const DevLink **out);
int compile_chain(const rtbhip_et *ets, int m, const double *qlim, Chain *out);
int compile_poe(const double *twists, int n, const double *T0, const double *qlim, Chain *out);
struct Affine;
void chain_tail(const Chain *c, const Affine &tool, double out12[12]);
void note_launch(int grid, int block, int lds);

int device_cu_count(int *cus);
int pool_keep_cached();    // the current device's default stream-ordered pool keeps freed blocks (release threshold raised once per device)

// ---------------------------------------------------------------- the host-pointer boundary (hostpipe.cpp)
// Row-chunked, double-buffered H2D -> kernel -> D2H over two persistent slots (streams, device buffers, pinned staging).
// in[i] / out[i] are HOST arrays of N rows of in_row[i] / out_row[i] bytes (NULL entries are skipped and handed to the
// functor as NULL); the functor enqueues the kernel(s) for rows [row0, row0 + rows) on device copies of the chunk.
struct HostIO {
    static constexpr int kMax = 6;
    const void *in[kMax] = {nullptr};
    size_t in_row[kMax] = {0};
    int n_in = 0;
    void *out[kMax] = {nullptr};
    size_t out_row[kMax] = {0};
    int n_out = 0;
    void add_in(const void *p, size_t row) { in[n_in] = p; in_row[n_in] = row; ++n_in; }
    void add_out(void *p, size_t row) { out[n_out] = p; out_row[n_out] = row; ++n_out; }
};
using ChunkLaunch = std::function<int(const void *const *din, void *const *dout, int64_t row0, int64_t rows, hipStream_t s)>;
int host_pipeline(const HostIO &io, int64_t N, const ChunkLaunch &launch);
void hostpipe_release();
int host_alloc(size_t bytes, void **out);        // pinned host memory, cached between uses (rtbhip_host_alloc)
int host_free(void *p);
void host_cache_trim(size_t keep_bytes);
int dev_cache_alloc(size_t bytes, void **out);   // device staging buffers of the remaining host-path calls: cached, not per call
void dev_cache_free(void *p);
void dev_cache_release();
void dev_cache_trim(size_t keep_bytes);

// ---------------------------------------------------------------- kernel launchers (device pointers)
struct Affine { double v[12]; int used; };  // row-major 3x4, host-side small parameter

int launch_kin(const Chain *c, const DevChain &dc, const double *q, int64_t N, const Affine &base,
               const Affine &tool, int frame, double *T, double *J, double *H, hipStream_t s);

int launch_kin_packed(const Chain *c, const DevChain &dc, const double *q, int64_t N, const Affine &base,
                      const Affine &tool, int frame, double *TJ, hipStream_t s);

int launch_kin_diff(const Chain *c, const DevChain &dc, int mode, int axes, const double *q, const double *qd, int64_t N,
                    const Affine &tool, int frame, double *out, hipStream_t s);

int launch_hess_from_jac(int n, const double *J, int64_t N, double *H, hipStream_t s);
// manipulability (mode 0; axes bits 8..9 = method) / manipulability Jacobian (mode 1: H formed from J, 2: H supplied) of supplied Jacobians (diffjac_kernels.hip)
int launch_diff_from_jac(int mode, int n, const double *J, const double *H, int64_t N, int axes, double *out, hipStream_t s);
int launch_p_servo(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int64_t N, int method, const double *gain6, double threshold, double *v,
                   unsigned char *arrived, hipStream_t s);
int launch_angle_axis(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int64_t N, double *e, hipStream_t s, int method = 0);

struct FrameTable;
int compile_frames(const Chain *c, const int32_t *marks, int nmarks, FrameTable *ft);
int launch_frames(const Chain *c, const DevChain &dc, const FrameTable &ft, const double *q, int64_t N, double *out, hipStream_t s);
int launch_partial(int n, int order, const double *const *lower, int64_t N, double *out, hipStream_t s);
bool partial3_needs_no_hessian(int n);   // order 3: k_partial3 forms the Hessians itself, lower[1] may be NULL

struct FleetEntry {   // device-visible descriptor of one chain of a fleet launch
    DevChain dc;
    const double *q;
    double *T;
    double *J;
    int64_t N;
    int64_t tile0;    // first global tile index of this chain
    int32_t n, q_width, stride, pad;
};
int launch_fleet(const std::vector<FleetEntry> &entries, int frame, hipStream_t s, bool packed = false);   // packed: e.T is the (N, 16 + 6n) array, e.J unused

int launch_rne(const Dyn *d, const DevLink *links, const double *q, const double *qd,
               const double *qdd, int64_t N, const double *grav3, const double *fext6, double *tau,
               hipStream_t s, double *wbase = nullptr);   // wbase (N,6): base wrench as well (run-time-n kernel)

int launch_dyn(const Dyn *d, const DevLink *links, int mode, const double *q, const double *qd, const double *tq,
               int64_t N, const double *grav3, double *out, hipStream_t s);

struct IkParams {
    int ilimit, slimit, reject_jl, method, flavour;
    double tol, lambda;
    double we[6];
    uint64_t seed;
    double kq = 0.0, km = 0.0, ps = 0.1;             // null-space terms of the Python solvers; kq <= 0: none
    double pi[RTBHIP_MAX_JOINTS];                    // ... influence distance per joint (filled by ik_entry)
    double ks = 1.0;                                 // IK_QP (method 5): slack gain (kj travels in lambda)
    int64_t target0 = 0;                             // restart-generator key offset of row 0 (rtbhip_ik_target_base)
};
int ik_check_limits(const Chain *c, const IkParams &ip, int64_t N);      // argument-only refusals of the device build (ik_kernels.hip)
int launch_ik(const Chain *c, const DevChain &dc, const double *qlim, const double *Tep, int64_t N,
              const double *q0, const IkParams &p, double *q_out, int32_t *success, int32_t *iters,
              int32_t *searches, double *residual, hipStream_t s);
void ik_restart_host(const Chain *c, uint64_t seed, int64_t target, int search, double *q_n);

#endif  // RTB_HOST_SIDE
}  // namespace rtbhip
