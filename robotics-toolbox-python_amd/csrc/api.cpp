// api.cpp -- the extern "C" entry points of librtbhip.so (declared in include/rtbhip.h).
// Argument checking, handle registry, lazy per-device upload of the chain / link tables, and the
// host-memory convenience path (stage -> launch -> copy back).  No arithmetic lives here.
#include "rtbhip_internal.h"
#include "partial_device.h"
#include "frames_device.h"
#include "tree_device.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <memory>
#include <unordered_map>
#include <functional>

namespace rtbhip {

// ---------------------------------------------------------------- errors
static thread_local std::string g_err;
static thread_local int g_last_launch[3] = {0, 0, 0};

void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();  // clear sticky state where possible
    return RTBHIP_EHIP;
}
void note_launch(int grid, int block, int lds)
{
    g_last_launch[0] = grid; g_last_launch[1] = block; g_last_launch[2] = lds;
}

// ---------------------------------------------------------------- tracing ranges (opt-in)
// RTBHIP_ROCTX=1: every compute entry point is bracketed by a roctx range named after it, so `rocprofv3 --marker-trace`
// shows which ABI call a kernel belongs to (the reference has no tracing hooks).  The marker library is looked
// up at run time -- nothing links against it and nothing happens when the variable is unset.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *on = std::getenv("RTBHIP_ROCTX");
        if (!on || !*on || *on == '0') return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }
struct TraceRange {
    bool on;
    explicit TraceRange(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~TraceRange() { if (on) roctx().pop(); }
    TraceRange(const TraceRange &) = delete;
};
}  // namespace
#define RTB_TRACE(name) ::rtbhip::TraceRange _trace_range(name)

// ---------------------------------------------------------------- handle registry
// The registries are heap objects that are never destroyed: handles that are still registered when the process exits (module-level
// robot objects whose Python __del__ never ran) must not have their destructors -- which call hipFree -- run during static
// destruction, possibly after the HIP runtime has been torn down.  The OS reclaims device memory with the process.
static std::mutex &g_reg_mu = *new std::mutex();
static std::unordered_map<uint64_t, std::shared_ptr<Chain>> &g_chains = *new std::unordered_map<uint64_t, std::shared_ptr<Chain>>();
static std::unordered_map<uint64_t, std::shared_ptr<Dyn>> &g_dyns = *new std::unordered_map<uint64_t, std::shared_ptr<Dyn>>();
static std::unordered_map<uint64_t, std::shared_ptr<Tree>> &g_trees = *new std::unordered_map<uint64_t, std::shared_ptr<Tree>>();
static std::atomic<uint64_t> g_next{1};

std::shared_ptr<Chain> chain_from_handle(rtbhip_chain_t h)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_chains.find(h);
    return it == g_chains.end() ? nullptr : it->second;
}
std::shared_ptr<Dyn> dyn_from_handle(rtbhip_dyn_t h)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_dyns.find(h);
    return it == g_dyns.end() ? nullptr : it->second;
}

std::shared_ptr<Tree> tree_from_handle(rtbhip_tree_t h)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_trees.find(h);
    return it == g_trees.end() ? nullptr : it->second;
}

Chain::~Chain()
{
    for (auto &kv : dev_ops) (void)hipFree(kv.second);
    for (auto &kv : dev_qlim) (void)hipFree(kv.second);
}
Dyn::~Dyn() { for (auto &kv : dev_links) (void)hipFree(kv.second); }

static DevChain view_of(const Chain *c, const void *base)
{
    DevChain v;
    const char *p = (const char *)base;
    v.seg = (const DevSeg *)p;
    p += c->seg.size() * sizeof(DevSeg);
    v.jmeta = (const int32_t *)p;
    return v;
}

DevChain chain_host_view(const Chain *c)
{
    DevChain v;
    v.seg = c->seg.data();
    v.jmeta = c->jmeta.data();
    return v;
}

int chain_device_ops(Chain *c, DevChain *out, const double **qlim_out)
{
    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->dev_ops.find(dev);
    if (it == c->dev_ops.end()) {
        std::vector<char> blob(c->seg.size() * sizeof(DevSeg) + c->jmeta.size() * sizeof(int32_t) + 16, 0);
        char *w = blob.data();
        std::memcpy(w, c->seg.data(), c->seg.size() * sizeof(DevSeg));
        w += c->seg.size() * sizeof(DevSeg);
        if (!c->jmeta.empty()) std::memcpy(w, c->jmeta.data(), c->jmeta.size() * sizeof(int32_t));
        void *d = nullptr;
        double *ql = nullptr;
        hipError_t e = hipMalloc(&d, blob.size());
        if (e == hipSuccess) e = hipMemcpy(d, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&ql, (c->qlim.size() ? c->qlim.size() : 1) * sizeof(double));
        if (e == hipSuccess && !c->qlim.empty()) e = hipMemcpy(ql, c->qlim.data(), c->qlim.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {             // nothing half-uploaded stays behind
            if (d) (void)hipFree(d);
            if (ql) (void)hipFree(ql);
            return hip_fail(e, "chain table upload");
        }
        c->dev_ops[dev] = d;
        c->dev_qlim[dev] = ql;
        it = c->dev_ops.find(dev);
    }
    *out = view_of(c, it->second);
    if (qlim_out) *qlim_out = c->dev_qlim[dev];
    return RTBHIP_OK;
}

int dyn_device_links(Dyn *d, const DevLink **out)
{
    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(d->mu);
    auto it = d->dev_links.find(dev);
    if (it == d->dev_links.end()) {
        DevLink *p = nullptr;
        hipError_t e = hipMalloc((void **)&p, d->links.size() * sizeof(DevLink));
        if (e == hipSuccess) e = hipMemcpy(p, d->links.data(), d->links.size() * sizeof(DevLink), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (p) (void)hipFree(p); return hip_fail(e, "link table upload"); }
        d->dev_links[dev] = p;
        it = d->dev_links.find(dev);
    }
    *out = it->second;
    return RTBHIP_OK;
}

int tree_device_groups(Tree *t, const DevGroup **out)
{
    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(t->mu);
    auto it = t->dev_groups.find(dev);
    if (it == t->dev_groups.end()) {
        DevGroup *p = nullptr;
        hipError_t e = hipMalloc((void **)&p, t->groups.size() * sizeof(DevGroup));
        if (e == hipSuccess) e = hipMemcpy(p, t->groups.data(), t->groups.size() * sizeof(DevGroup), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (p) (void)hipFree(p); return hip_fail(e, "tree table upload"); }
        t->dev_groups[dev] = p;
        it = t->dev_groups.find(dev);
    }
    *out = it->second;
    return RTBHIP_OK;
}

// Stream-ordered temporaries (hipMallocAsync: the rows of the IK schedules, the lower-order tensors of partial_fkine0) come from the device's
// default memory pool.  Its release threshold is 0 by default: everything freed goes back to the OS at the next synchronisation, so EVERY
// call pays for fresh allocations (measured: 0.35 ms for the 99 MB of rows of a 1e5-target IK call).  Raised once per device: freed
// blocks stay in the pool; rtbhip_trim / rtbhip_shutdown hand them back.
int pool_keep_cached()
{
    static std::mutex mu;
    static std::map<int, bool> done;
    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev]) return RTBHIP_OK;
    hipMemPool_t pool;
    RTB_HIP(hipDeviceGetDefaultMemPool(&pool, dev));
    uint64_t keep = ~0ull;
    RTB_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    done[dev] = true;
    return RTBHIP_OK;
}

int device_cu_count(int *cus)
{
    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    RTB_HIP(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
    return RTBHIP_OK;
}

#define RTB_TRY_(expr)                  \
    do {                                \
        int _rc = (expr);               \
        if (_rc != RTBHIP_OK) return _rc; \
    } while (0)

// ---------------------------------------------------------------- host staging helper
// RAII device buffers of the RTBHIP_MEM_HOST path of the calls that are not row-pipelined (IK, the dynamics terms, the
// differential-kinematics consumers ...): drawn from a cache of device blocks (hostpipe.cpp), not allocated per call.
// The high-volume calls -- fkine / jacob / fkine_jacob / hessian and rne -- go through host_pipeline instead.
struct Staging {
    std::vector<void *> bufs;
    ~Staging() { for (void *p : bufs) dev_cache_free(p); }
    int in(const void *host, size_t bytes, void **dev)
    {
        *dev = nullptr;
        if (host == nullptr || bytes == 0) return RTBHIP_OK;
        RTB_TRY_(dev_cache_alloc(bytes, dev));
        bufs.push_back(*dev);
        RTB_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
        return RTBHIP_OK;
    }
    int out(size_t bytes, void **dev)
    {
        *dev = nullptr;
        if (bytes == 0) return RTBHIP_OK;
        RTB_TRY_(dev_cache_alloc(bytes, dev));
        bufs.push_back(*dev);
        return RTBHIP_OK;
    }
};
#define RTB_TRY(expr)                   \
    do {                                \
        int _rc = (expr);               \
        if (_rc != RTBHIP_OK) return _rc; \
    } while (0)

static int fetch(void *host, const void *dev, size_t bytes)
{
    if (host == nullptr || bytes == 0) return RTBHIP_OK;
    RTB_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return RTBHIP_OK;
}

static Affine affine_from16(const double *m16)
{
    Affine a;
    a.used = m16 != nullptr;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) a.v[4 * r + c] = m16 ? m16[4 * r + c] : (r == c ? 1.0 : 0.0);
    return a;
}

// The device a device-pointer call runs on.  The caller's buffers decide: when they live on a GPU other than the current one
// (a tensor on cuda:1 while device 0 is current -- a single process driving several GPUs) the call switches to that GPU for its
// duration -- table look-up / upload, launch geometry and the launch itself all happen there -- and the destructor restores the
// caller's current device.  `stream` must be a stream of the buffers' device (NULL = that device's default stream).
struct DeviceScope {
    int prev = -1;
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    int enter_for(const char *fn, const void *buf)
    {
        hipPointerAttribute_t at;
        int cur = 0;
        if (buf && hipPointerGetAttributes(&at, buf) == hipSuccess && hipGetDevice(&cur) == hipSuccess) {
            if (at.type == hipMemoryTypeDevice && at.device != cur && prev < 0) {
                hipError_t e = hipSetDevice(at.device);
                if (e != hipSuccess) return hip_fail(e, (std::string(fn) + ": hipSetDevice to the buffer's GPU").c_str());
                prev = cur;
            }
        } else {
            (void)hipGetLastError();
        }
        return RTBHIP_OK;
    }
};

static int check_batch(const char *fn, const void *q, int64_t N, int mem, DeviceScope *scope)
{
    if (N < 0) { set_error(std::string(fn) + ": negative N"); return RTBHIP_EINVAL; }
    if (N > 0 && q == nullptr) { set_error(std::string(fn) + ": NULL input with N > 0"); return RTBHIP_EINVAL; }
    if (mem != RTBHIP_MEM_HOST && mem != RTBHIP_MEM_DEVICE) { set_error(std::string(fn) + ": bad mem kind"); return RTBHIP_EINVAL; }
    if (mem == RTBHIP_MEM_DEVICE && N > 0 && scope) return scope->enter_for(fn, q);
    return RTBHIP_OK;
}

static int kin_entry(const char *fn, rtbhip_chain_t h, const double *q, int64_t N, const double *base16,
                     const double *tool16, int frame, double *T, double *J, double *H, int mem, void *stream)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    RTB_TRACE((std::string("rtbhip_") + fn).c_str());
    if (!c) { set_error(std::string(fn) + ": unknown chain handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch(fn, q, N, mem, &dscope));
    if (frame != 0 && frame != 1) { set_error(std::string(fn) + ": frame must be 0 (jacob0) or 1 (jacobe)"); return RTBHIP_EINVAL; }
    if (N > 0 && !T && !J && !H) { set_error(std::string(fn) + ": no output buffer"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    if (c->n == 0) {                  // a chain of constants: J is (N, 6, 0) and H (N, 0, 6, 0) -- nothing to write (the reference returns empty arrays)
        J = nullptr; H = nullptr;
        if (!T) return RTBHIP_OK;
    }
    DevChain ops;
    RTB_TRY(chain_device_ops(c, &ops, nullptr));
    Affine base = affine_from16(base16), tool = affine_from16(tool16);
    const size_t n = (size_t)c->n, qw = (size_t)c->q_width;
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_kin(c, ops, q, N, base, tool, frame, T, J, H, (hipStream_t)stream);
    // host arrays: rows stream through the two-slot pipeline (hostpipe.cpp); the functor sees device copies of one chunk
    HostIO io;
    io.add_in(q, qw * 8);
    io.add_out(T, 128);
    io.add_out(J, 48 * n);
    io.add_out(H, 48 * n * n);
    return host_pipeline(io, N, [&](const void *const *din, void *const *dout, int64_t, int64_t rows, hipStream_t s) {
        return launch_kin(c, ops, (const double *)din[0], rows, base, tool, frame, (double *)dout[0], (double *)dout[1], (double *)dout[2], s);
    });
}

// rtbhip_fkine_jacob_packed: as kin_entry for (T, J), but one (N, 16 + 6n) output array
static int kin_packed_entry(rtbhip_chain_t h, const double *q, int64_t N, const double *base16, const double *tool16, int frame, double *TJ,
                            int mem, void *stream)
{
    const char *fn = "fkine_jacob_packed";
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    RTB_TRACE("rtbhip_fkine_jacob_packed");
    if (!c) { set_error(std::string(fn) + ": unknown chain handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch(fn, q, N, mem, &dscope));
    if (frame != 0 && frame != 1) { set_error(std::string(fn) + ": frame must be 0 (jacob0) or 1 (jacobe)"); return RTBHIP_EINVAL; }
    if (N > 0 && !TJ) { set_error(std::string(fn) + ": no output buffer"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    DevChain ops;
    RTB_TRY(chain_device_ops(c, &ops, nullptr));
    Affine base = affine_from16(base16), tool = affine_from16(tool16);
    const size_t n = (size_t)c->n, qw = (size_t)c->q_width;
    if (c->n == 0) {                  // a chain of constants: the row is the pose alone, (N, 16) -- the plain fkine kernel writes exactly that
        if (mem == RTBHIP_MEM_DEVICE) return launch_kin(c, ops, q, N, base, tool, frame, TJ, nullptr, nullptr, (hipStream_t)stream);
        HostIO io0;
        io0.add_in(q, qw * 8);
        io0.add_out(TJ, 128);
        return host_pipeline(io0, N, [&](const void *const *din, void *const *dout, int64_t, int64_t rows, hipStream_t s) {
            return launch_kin(c, ops, (const double *)din[0], rows, base, tool, frame, (double *)dout[0], nullptr, nullptr, s);
        });
    }
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_kin_packed(c, ops, q, N, base, tool, frame, TJ, (hipStream_t)stream);
    HostIO io;
    io.add_in(q, qw * 8);
    io.add_out(TJ, 128 + 48 * n);
    return host_pipeline(io, N, [&](const void *const *din, void *const *dout, int64_t, int64_t rows, hipStream_t s) {
        return launch_kin_packed(c, ops, (const double *)din[0], rows, base, tool, frame, (double *)dout[0], s);
    });
}

void kin_tune(const char *key, int value);
void rne_tune(const char *key, int value);
void ik_tune(const char *key, int value);
void partial_tune(const char *key, int value);
void ik_release_device_state();
int ik_prepare_device();
void hostpipe_tune(const char *key, int value);
void shard_tune(const char *key, int value);
void tree_tune(const char *key, int value);
std::string tree_jit_knowledge(const Tree *t, std::string *type_name);      // tree_kernels.hip
bool tree_jit_applies(const Tree *t);
// ask for a handle's run-time instantiations (jit.cpp).  At *_create: the kernels a first call is most likely to want; all = every variant.
static void jit_request_chain(const Chain *c, bool touch)
{
    for (const std::string &e : ik_jit_names(c)) jit_request("ik_kernels.hip", e, std::string(), touch);
}
static void jit_request_dyn(const Dyn *d, bool all, bool touch)
{
    const std::vector<std::string> nm = rne_jit_names(d);          // k_rne, k_rne_atrest, k_dyn x 3
    for (size_t i = 0; i < nm.size() && (all || i < 2); ++i) jit_request(i < 2 ? "rne_kernels.hip" : "dyn_kernels.hip", nm[i], std::string(), touch);
}
static void jit_request_tree(const Tree *t, bool all, bool touch)
{
    const std::vector<std::string> nm = tree_jit_names(t);         // k_tree_rne (+ at rest), k_tree_dyn x 3
    if (nm.empty()) return;
    std::string tn;
    const std::string pre = tree_jit_knowledge(t, &tn);
    for (size_t i = 0; i < nm.size() && (all || i < 1); ++i)
        jit_request(nm[i].find("k_tree_dyn") != std::string::npos ? "tree_dyn_kernels.hip" : "tree_kernels.hip", nm[i], pre, touch);
}

}  // namespace rtbhip

using namespace rtbhip;

extern "C" {

const char *rtbhip_last_error(void) { return g_err.c_str(); }
int rtbhip_version(void) { return 100; }

int rtbhip_init(int32_t n_devices)
{
    int have = 0;
    RTB_HIP(hipGetDeviceCount(&have));
    if (have < 1) { set_error("init: no HIP device is visible"); return RTBHIP_EHIP; }
    if (n_devices > have) { set_error("init: fewer HIP devices are visible than requested"); return RTBHIP_EHIP; }
    return RTBHIP_OK;
}

void rtbhip_shutdown(void)
{
    // device copies of every table are dropped (handles stay valid: tables are re-uploaded lazily on next use)
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (auto &kv : g_chains) {
        std::lock_guard<std::mutex> l2(kv.second->mu);
        for (auto &d : kv.second->dev_ops) (void)hipFree(d.second);
        for (auto &d : kv.second->dev_qlim) (void)hipFree(d.second);
        kv.second->dev_ops.clear(); kv.second->dev_qlim.clear();
    }
    for (auto &kv : g_dyns) {
        std::lock_guard<std::mutex> l2(kv.second->mu);
        for (auto &d : kv.second->dev_links) (void)hipFree(d.second);
        kv.second->dev_links.clear();
    }
    for (auto &kv : g_trees) {
        std::lock_guard<std::mutex> l2(kv.second->mu);
        for (auto &d : kv.second->dev_groups) (void)hipFree(d.second);
        kv.second->dev_groups.clear();
    }
    ik_release_device_state();
    hostpipe_release();
    dev_cache_release();
    host_cache_trim(0);
    // the stream-ordered temporaries of partial_fkine0 stay cached in the device's default pool: hand them back
    int dev = 0;
    hipMemPool_t pool;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
}

int rtbhip_device_count(int *count)
{
    if (!count) { set_error("device_count: NULL"); return RTBHIP_EINVAL; }
    *count = 0;
    RTB_HIP(hipGetDeviceCount(count));
    return RTBHIP_OK;
}

int rtbhip_chain_create(const rtbhip_et *ets, int32_t m, const double *qlim, rtbhip_chain_t *chain)
{
    if (!chain) { set_error("chain_create: NULL out"); return RTBHIP_EINVAL; }
    std::shared_ptr<Chain> c(new Chain());
    RTB_TRY(compile_chain(ets, m, qlim, c.get()));
    jit_request_chain(c.get(), false);       // a chain without a built-in k_ik instantiation: ask for its own (worker thread; nobody waits)
    uint64_t h = g_next.fetch_add(1);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_chains[h] = std::move(c);
    *chain = h;
    return RTBHIP_OK;
}

int rtbhip_chain_create_poe(const double *twists, int32_t n, const double *T0_16, const double *qlim, rtbhip_chain_t *chain)
{
    if (!chain) { set_error("chain_create_poe: NULL out"); return RTBHIP_EINVAL; }
    std::shared_ptr<Chain> c(new Chain());
    RTB_TRY(compile_poe(twists, n, T0_16, qlim, c.get()));
    jit_request_chain(c.get(), false);
    uint64_t h = g_next.fetch_add(1);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_chains[h] = std::move(c);
    *chain = h;
    return RTBHIP_OK;
}

// Make the handle's device table resident on `device` (-1: the current one) NOW: after this returns, device-pointer calls with the
// handle on that device only enqueue kernels -- no allocation, no synchronous copy -- so they can be captured into a hipGraph
// without a warm-up call.  Also sizes the per-device scheduler state rtbhip_ik_lm needs.
static int upload_on(int32_t device, const std::function<int()> &fn)
{
    int cur = 0;
    RTB_HIP(hipGetDevice(&cur));
    int have = 0;
    RTB_HIP(hipGetDeviceCount(&have));
    if (device < -1 || device >= have) { set_error("upload: no such device"); return RTBHIP_EINVAL; }
    const bool sw = device >= 0 && device != cur;
    if (sw) RTB_HIP(hipSetDevice(device));
    const int rc = fn();
    if (sw) (void)hipSetDevice(cur);
    return rc;
}

int rtbhip_chain_upload(rtbhip_chain_t chain, int32_t device)
{
    const std::shared_ptr<Chain> c = chain_from_handle(chain);
    if (!c) { set_error("chain_upload: unknown handle"); return RTBHIP_EINVAL; }
    return upload_on(device, [&]() -> int {
        DevChain ops;
        RTB_TRY(chain_device_ops(c.get(), &ops, nullptr));
        return ik_prepare_device();
    });
}

int rtbhip_dyn_upload(rtbhip_dyn_t dyn, int32_t device)
{
    const std::shared_ptr<Dyn> d = dyn_from_handle(dyn);
    if (!d) { set_error("dyn_upload: unknown handle"); return RTBHIP_EINVAL; }
    return upload_on(device, [&]() -> int { const DevLink *l; return dyn_device_links(d.get(), &l); });
}

int rtbhip_tree_upload(rtbhip_tree_t tree, int32_t device)
{
    const std::shared_ptr<Tree> t = tree_from_handle(tree);
    if (!t) { set_error("tree_upload: unknown handle"); return RTBHIP_EINVAL; }
    return upload_on(device, [&]() -> int { const DevGroup *g; return tree_device_groups(t.get(), &g); });
}

// Hand idle cached memory back: device staging blocks of the host-pointer calls above `keep_device_bytes`, pinned host blocks above
// `keep_pinned_bytes` (0, 0 = everything that is not in use).
int rtbhip_trim(uint64_t keep_device_bytes, uint64_t keep_pinned_bytes)
{
    dev_cache_trim((size_t)keep_device_bytes);
    host_cache_trim((size_t)keep_pinned_bytes);
    int dev = 0;
    hipMemPool_t pool;      // the stream-ordered temporaries' pool (pool_keep_cached)
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) (void)hipMemPoolTrimTo(pool, (size_t)keep_device_bytes);
    else (void)hipGetLastError();
    return RTBHIP_OK;
}

int rtbhip_chain_destroy(rtbhip_chain_t chain)
{
    std::shared_ptr<Chain> c;             // the device tables go with the last reference (a launch in flight keeps one)
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_chains.find(chain);
        if (it == g_chains.end()) { set_error("chain_destroy: unknown handle"); return RTBHIP_EINVAL; }
        c = std::move(it->second);
        g_chains.erase(it);
    }
    return RTBHIP_OK;
}

int rtbhip_chain_info(rtbhip_chain_t chain, int32_t *n, int32_t *m, int32_t *q_width)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(chain);
    Chain *c = c_owner.get();
    if (!c) { set_error("chain_info: unknown handle"); return RTBHIP_EINVAL; }
    if (n) *n = c->n;
    if (m) *m = (int32_t)c->ets.size();
    if (q_width) *q_width = c->q_width;
    return RTBHIP_OK;
}

int rtbhip_chain_set_q_width(rtbhip_chain_t chain, int32_t q_width)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(chain);
    Chain *c = c_owner.get();
    if (!c) { set_error("chain_set_q_width: unknown handle"); return RTBHIP_EINVAL; }
    int need = 0;
    for (int j = 0; j < c->n; ++j) need = std::max(need, jm_jq(c->jmeta[j]) + 1);
    if (q_width < need || q_width > 256) {
        set_error("chain_set_q_width: width " + std::to_string(q_width) + " outside [" + std::to_string(need) + ", 256] for this chain");
        return RTBHIP_EINVAL;
    }
    c->q_width = q_width;     // the kernels take the row pitch of q from here; the chain tables do not depend on it
    return RTBHIP_OK;
}

int rtbhip_fkine(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16,
                 const double *tool16, double *T, int32_t mem, void *stream)
{
    if (N > 0 && !T) { set_error("fkine: NULL T"); return RTBHIP_EINVAL; }
    return kin_entry("fkine", chain, q, N, base16, tool16, 0, T, nullptr, nullptr, mem, stream);
}

int rtbhip_jacob(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16,
                 int32_t frame, double *J, int32_t mem, void *stream)
{
    if (N > 0 && !J) { set_error("jacob: NULL J"); return RTBHIP_EINVAL; }
    return kin_entry("jacob", chain, q, N, nullptr, tool16, frame, nullptr, J, nullptr, mem, stream);
}

int rtbhip_fkine_jacob(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16,
                       const double *tool16, int32_t frame, double *T, double *J, int32_t mem,
                       void *stream)
{
    if (N > 0 && (!T || !J)) { set_error("fkine_jacob: NULL T or J"); return RTBHIP_EINVAL; }
    return kin_entry("fkine_jacob", chain, q, N, base16, tool16, frame, T, J, nullptr, mem, stream);
}

int rtbhip_fkine_jacob_packed(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16,
                              const double *tool16, int32_t frame, double *TJ, int32_t mem, void *stream)
{
    return kin_packed_entry(chain, q, N, base16, tool16, frame, TJ, mem, stream);
}

int rtbhip_hessian(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16,
                   int32_t frame, double *H, int32_t mem, void *stream)
{
    if (N > 0 && !H) { set_error("hessian: NULL H"); return RTBHIP_EINVAL; }
    return kin_entry("hessian", chain, q, N, nullptr, tool16, frame, nullptr, nullptr, H, mem, stream);
}

/* ETS_hessian0 / ETS_hessiane with a supplied Jacobian (core/fknm.cpp:583-783 -> _ETS_hessian core/methods.cpp:16-32) */
int rtbhip_hessian_from_jacobian(const double *J, int64_t N, int32_t n, double *H, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_hessian_from_jacobian");
    DeviceScope dscope;
    RTB_TRY(check_batch("hessian_from_jacobian", J, N, mem, &dscope));
    if (n < 1 || n > RTBHIP_MAX_JOINTS) { set_error("hessian_from_jacobian: n must be 1..RTBHIP_MAX_JOINTS"); return RTBHIP_ELIMIT; }
    if (N > 0 && !H) { set_error("hessian_from_jacobian: NULL H"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    if (mem == RTBHIP_MEM_DEVICE) {
        if (((uintptr_t)J | (uintptr_t)H) & 15) { set_error("hessian_from_jacobian: device buffers must be 16-byte aligned"); return RTBHIP_EINVAL; }
        return launch_hess_from_jac(n, J, N, H, (hipStream_t)stream);
    }
    Staging st;
    void *dJ, *dH;
    const size_t jb = (size_t)N * 48 * n, hb = jb * n;
    RTB_TRY(st.in(J, jb, &dJ));
    RTB_TRY(st.out(hb, &dH));
    RTB_TRY(launch_hess_from_jac(n, (const double *)dJ, N, (double *)dH, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(H, dH, hb));
    return RTBHIP_OK;
}

/* Robot.manipulability(J=...) / Robot.jacobm(J=..., H=...) (robot/Robot.py:701-905, :1101-1235): pure functions of the supplied arrays */
static int diff_from_jac_entry(const char *fn, int mode, const double *J, const double *H, int64_t N, int32_t n, int32_t axes, double *out,
                               int32_t mem, void *stream)
{
    if (n < 1 || n > 16) { set_error(std::string(fn) + ": n must be 1..16"); return RTBHIP_ELIMIT; }
    if ((axes & 63) == 0) { set_error(std::string(fn) + ": empty axes mask"); return RTBHIP_EINVAL; }
    if (N > 0 && !out) { set_error(std::string(fn) + ": NULL output"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch(fn, J, N, mem, &dscope));
    if (N == 0) return RTBHIP_OK;
    if (mem == RTBHIP_MEM_DEVICE) {
        if (((uintptr_t)J | (uintptr_t)H) & 15) { set_error(std::string(fn) + ": device buffers must be 16-byte aligned"); return RTBHIP_EINVAL; }
        return launch_diff_from_jac(mode, n, J, H, N, axes, out, (hipStream_t)stream);
    }
    Staging st;
    void *dJ, *dH = nullptr, *dout;
    const size_t jb = (size_t)N * 48 * n, ob = (size_t)N * 8 * (mode == 0 ? 1 : n);
    RTB_TRY(st.in(J, jb, &dJ));
    if (H) RTB_TRY(st.in(H, jb * n, &dH));
    RTB_TRY(st.out(ob, &dout));
    RTB_TRY(launch_diff_from_jac(mode, n, (const double *)dJ, (const double *)dH, N, axes, (double *)dout, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(out, dout, ob));
    return RTBHIP_OK;
}

int rtbhip_manipulability_from_jacobian(const double *J, int64_t N, int32_t n, int32_t axes_mask, int32_t method, double *m, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_manipulability_from_jacobian");
    if (method < 0 || method > 2) { set_error("manipulability_from_jacobian: method must be 0 yoshikawa, 1 minsingular, 2 invcondition"); return RTBHIP_EINVAL; }
    return diff_from_jac_entry("manipulability_from_jacobian", 0, J, nullptr, N, n, (axes_mask & 63) | (method << 8), m, mem, stream);
}

int rtbhip_jacobm_from_jacobian(const double *J, const double *H, int64_t N, int32_t n, int32_t axes_mask, double *Jm, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_jacobm_from_jacobian");
    return diff_from_jac_entry("jacobm_from_jacobian", H ? 2 : 1, J, H, N, n, axes_mask & 63, Jm, mem, stream);
}

/* fknm.Angle_Axis (core/fknm.cpp:112-162 -> _angle_axis core/ik.cpp:241-286), batched with broadcasting */
static int pose_error_entry(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int method, double *e, int32_t mem, void *stream);

int rtbhip_angle_axis(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, double *e, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_angle_axis");
    return pose_error_entry(Te, nTe, Tep, nTep, 0, e, mem, stream);
}

/* the error vector of tools/p_servo.py:46-117: method 0 "angle-axis" (= rtbhip_angle_axis), 1 "rpy" (the reference's default) */
int rtbhip_p_servo_error(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int32_t method, double *e, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_p_servo_error");
    if (method != 0 && method != 1) { set_error("p_servo_error: method must be 0 angle-axis or 1 rpy"); return RTBHIP_EINVAL; }
    return pose_error_entry(Te, nTe, Tep, nTep, method, e, mem, stream);
}

/* tools/p_servo.py:46-117 whole: v = diag(gain) e and arrived = sum|e| < threshold in the launch that forms e */
int rtbhip_p_servo(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int32_t method, const double *gain6, double threshold, double *v,
                   uint8_t *arrived, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_p_servo");
    if (method != 0 && method != 1) { set_error("p_servo: method must be 0 angle-axis or 1 rpy"); return RTBHIP_EINVAL; }
    if (mem != RTBHIP_MEM_HOST && mem != RTBHIP_MEM_DEVICE) { set_error("p_servo: bad mem kind"); return RTBHIP_EINVAL; }
    if (nTe < 0 || nTep < 0) { set_error("p_servo: negative count"); return RTBHIP_EINVAL; }
    if (!gain6) { set_error("p_servo: NULL gain"); return RTBHIP_EINVAL; }
    if (nTe == 0 || nTep == 0) return RTBHIP_OK;
    const int64_t N = nTe > nTep ? nTe : nTep;
    if ((nTe != N && nTe != 1) || (nTep != N && nTep != 1)) { set_error("p_servo: the pose counts must be equal, or one of them 1"); return RTBHIP_EINVAL; }
    if (!Te || !Tep || !v || !arrived) { set_error("p_servo: NULL buffer"); return RTBHIP_EINVAL; }
    if (mem == RTBHIP_MEM_DEVICE) {
        if (((uintptr_t)Te | (uintptr_t)Tep | (uintptr_t)v) & 15) { set_error("p_servo: device buffers must be 16-byte aligned"); return RTBHIP_EINVAL; }
        DeviceScope dscope;
        RTB_TRY(dscope.enter_for("p_servo", v));
        return launch_p_servo(Te, nTe, Tep, nTep, N, method, gain6, threshold, v, arrived, (hipStream_t)stream);
    }
    Staging st;
    void *dA, *dB, *dV, *dF;
    RTB_TRY(st.in(Te, (size_t)nTe * 128, &dA));
    RTB_TRY(st.in(Tep, (size_t)nTep * 128, &dB));
    RTB_TRY(st.out((size_t)N * 48, &dV));
    RTB_TRY(st.out((size_t)N, &dF));
    RTB_TRY(launch_p_servo((const double *)dA, nTe, (const double *)dB, nTep, N, method, gain6, threshold, (double *)dV, (unsigned char *)dF, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(v, dV, (size_t)N * 48));
    return fetch(arrived, dF, (size_t)N);
}

static int pose_error_entry(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int method, double *e, int32_t mem, void *stream)
{
    if (mem != RTBHIP_MEM_HOST && mem != RTBHIP_MEM_DEVICE) { set_error("angle_axis: bad mem kind"); return RTBHIP_EINVAL; }
    if (nTe < 0 || nTep < 0) { set_error("angle_axis: negative count"); return RTBHIP_EINVAL; }
    if (nTe == 0 || nTep == 0) return RTBHIP_OK;
    const int64_t N = nTe > nTep ? nTe : nTep;
    if ((nTe != N && nTe != 1) || (nTep != N && nTep != 1)) { set_error("angle_axis: the pose counts must be equal, or one of them 1"); return RTBHIP_EINVAL; }
    if (!Te || !Tep || !e) { set_error("angle_axis: NULL buffer"); return RTBHIP_EINVAL; }
    if (mem == RTBHIP_MEM_DEVICE) {
        if (((uintptr_t)Te | (uintptr_t)Tep | (uintptr_t)e) & 15) { set_error("angle_axis: device buffers must be 16-byte aligned"); return RTBHIP_EINVAL; }
        DeviceScope dscope;
        RTB_TRY(dscope.enter_for("angle_axis", e));
        return launch_angle_axis(Te, nTe, Tep, nTep, N, e, (hipStream_t)stream, method);
    }
    Staging st;
    void *dA, *dB, *dE;
    RTB_TRY(st.in(Te, (size_t)nTe * 128, &dA));
    RTB_TRY(st.in(Tep, (size_t)nTep * 128, &dB));
    RTB_TRY(st.out((size_t)N * 48, &dE));
    RTB_TRY(launch_angle_axis((const double *)dA, nTe, (const double *)dB, nTep, N, (double *)dE, nullptr, method));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(e, dE, (size_t)N * 48));
    return RTBHIP_OK;
}

/* Robot.jacob0_dot / ETS.manipulability (yoshikawa) / ETS.jacobm */
static int diff_entry(const char *fn, rtbhip_chain_t h, int mode, int axes, const double *q, const double *qd, int64_t N,
                      const double *tool16, int frame, double *out, int mem, void *stream)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    RTB_TRACE((std::string("rtbhip_") + fn).c_str());
    if (!c) { set_error(std::string(fn) + ": unknown chain handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch(fn, q, N, mem, &dscope));
    if (frame != 0 && frame != 1) { set_error(std::string(fn) + ": frame must be 0 or 1"); return RTBHIP_EINVAL; }
    if (mode != 0 && mode != 3 && mode != 4 && (axes & 63) == 0) { set_error(std::string(fn) + ": empty axes mask"); return RTBHIP_EINVAL; }
    if (N > 0 && (!out || ((mode == 0 || mode == 4) && !qd))) { set_error(std::string(fn) + ": NULL qd/output"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    DevChain ops;
    RTB_TRY(chain_device_ops(c, &ops, nullptr));
    Affine tool = affine_from16(tool16);
    const size_t n = (size_t)c->n, qw = (size_t)c->q_width;
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_kin_diff(c, ops, mode, axes, q, qd, N, tool, frame, out, (hipStream_t)stream);
    Staging st;
    void *dq, *dqd = nullptr, *dout;
    const size_t obytes = (size_t)N * 8 * ((mode == 0 || mode == 3 || mode == 4) ? 6 * n : (mode == 1 ? 1 : n));
    RTB_TRY(st.in(q, (size_t)N * qw * 8, &dq));
    if (mode == 0 || mode == 4) RTB_TRY(st.in(qd, (size_t)N * qw * 8, &dqd));
    RTB_TRY(st.out(obytes, &dout));
    RTB_TRY(launch_kin_diff(c, ops, mode, axes, (const double *)dq, (const double *)dqd, N, tool, frame, (double *)dout, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(out, dout, obytes));
    return RTBHIP_OK;
}

int rtbhip_jacob_dot(rtbhip_chain_t chain, const double *q, const double *qd, int64_t N, const double *tool16,
                     int32_t frame, double *Jd, int32_t mem, void *stream)
{
    return diff_entry("jacob_dot", chain, 0, 63, q, qd, N, tool16, frame, Jd, mem, stream);
}

int rtbhip_jacob0_analytical(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t representation,
                             double *Ja, int32_t mem, void *stream)
{
    if (representation < 0 || representation > 3) { set_error("jacob0_analytical: representation must be 0 rpy/xyz, 1 rpy/zyx, 2 eul, 3 exp"); return RTBHIP_EINVAL; }
    return diff_entry("jacob0_analytical", chain, 3, representation, q, nullptr, N, tool16, 0, Ja, mem, stream);
}

int rtbhip_jacob0_dot_analytical(rtbhip_chain_t chain, const double *q, const double *qd, int64_t N, const double *tool16,
                                 int32_t representation, double *Jd, int32_t mem, void *stream)
{
    if (representation < 0 || representation > 3) { set_error("jacob0_dot_analytical: representation must be 0 rpy/xyz, 1 rpy/zyx, 2 eul, 3 exp"); return RTBHIP_EINVAL; }
    return diff_entry("jacob0_dot_analytical", chain, 4, representation, q, qd, N, tool16, 0, Jd, mem, stream);
}

int rtbhip_manipulability(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t axes_mask,
                          int32_t method, double *m, int32_t mem, void *stream)
{
    if (method < 0 || method > 2) { set_error("manipulability: method must be 0 yoshikawa, 1 minsingular, 2 invcondition"); return RTBHIP_EINVAL; }
    return diff_entry("manipulability", chain, 1, (axes_mask & 63) | (method << 8), q, nullptr, N, tool16, 0, m, mem, stream);
}

int rtbhip_jacobm(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t axes_mask, double *Jm,
                  int32_t mem, void *stream)
{
    return diff_entry("jacobm", chain, 2, axes_mask, q, nullptr, N, tool16, 0, Jm, mem, stream);
}

/* DHRobot.fkine_all / Robot.fkine_all (robot/DHRobot.py:1012-1064, robot/Robot.py:638-698), batched */
int rtbhip_link_frames(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16, const int32_t *marks,
                       int32_t nmarks, double *out, int32_t mem, void *stream)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(chain);
    Chain *c = c_owner.get();
    RTB_TRACE("rtbhip_link_frames");
    if (!c) { set_error("link_frames: unknown chain handle"); return RTBHIP_EINVAL; }
    if (mem != RTBHIP_MEM_HOST && mem != RTBHIP_MEM_DEVICE) { set_error("link_frames: bad mem kind"); return RTBHIP_EINVAL; }
    if (N < 0) { set_error("link_frames: negative N"); return RTBHIP_EINVAL; }
    if (nmarks > 0 && !marks) { set_error("link_frames: NULL marks"); return RTBHIP_EINVAL; }
    FrameTable ft;
    RTB_TRY(compile_frames(c, marks, nmarks, &ft));
    if (N == 0 || nmarks == 0) return RTBHIP_OK;
    if ((c->q_width > 0 && !q) || !out) { set_error("link_frames: NULL q / output"); return RTBHIP_EINVAL; }
    Affine b = affine_from16(base16);
    ft.has_base = b.used;
    for (int i = 0; i < 12; i++) ft.base[i] = b.v[i];
    DevChain ops;
    RTB_TRY(chain_device_ops(c, &ops, nullptr));
    if (mem == RTBHIP_MEM_DEVICE) return launch_frames(c, ops, ft, q, N, out, (hipStream_t)stream);
    Staging st;
    void *dq, *dout;
    const size_t obytes = (size_t)N * nmarks * 128;
    RTB_TRY(st.in(q, (size_t)N * c->q_width * 8, &dq));
    RTB_TRY(st.out(obytes, &dout));
    RTB_TRY(launch_frames(c, ops, ft, (const double *)dq, N, (double *)dout, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(out, dout, obytes));
    return RTBHIP_OK;
}

/* ETS.partial_fkine0 (robot/ETS.py:1821-2013): order >= 3; orders 1 and 2 are jacob0 / hessian0 */
int rtbhip_partial_fkine0(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t order,
                          double *out, int32_t mem, void *stream)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(chain);
    Chain *c = c_owner.get();
    RTB_TRACE("rtbhip_partial_fkine0");
    if (!c) { set_error("partial_fkine0: unknown chain handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch("partial_fkine0", q, N, mem, &dscope));
    if (order < 3 || order > kPartialMaxOrder) { set_error("partial_fkine0: order must be 3.." + std::to_string(kPartialMaxOrder)); return RTBHIP_EINVAL; }
    if (c->n < 1) { set_error("partial_fkine0: chain has no joints"); return RTBHIP_EINVAL; }
    if (N > 0 && !out) { set_error("partial_fkine0: NULL output"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    DevChain ops;
    RTB_TRY(chain_device_ops(c, &ops, nullptr));
    Affine base = affine_from16(nullptr), tool = affine_from16(tool16);
    const int n = c->n;
    hipStream_t s = mem == RTBHIP_MEM_DEVICE ? (hipStream_t)stream : nullptr;
    Staging st;
    void *dq = nullptr, *dout = nullptr;
    const size_t obytes = (size_t)N * (size_t)partial_size(n, order) * 8;
    if (mem == RTBHIP_MEM_HOST) {
        RTB_TRY(st.in(q, (size_t)N * c->q_width * 8, &dq));
        RTB_TRY(st.out(obytes, &dout));
    } else { dq = (void *)q; dout = out; }
    // the lower-order tensors are stream-ordered temporaries from the device's memory pool (kept cached between
    // calls): the call only enqueues work, as every other device-pointer entry point does
    RTB_TRY(pool_keep_cached());
    double *lower[kPartialMaxOrder] = {nullptr};
    int rc = RTBHIP_OK;
    const bool skip_h = order == 3 && partial3_needs_no_hessian(n);       // k_partial3 forms the Hessians from the Jacobians it stages
    for (int a = 1; a < order && rc == RTBHIP_OK; ++a) {
        if (a == 2 && skip_h) continue;
        void *p = nullptr;
        hipError_t e = hipMallocAsync(&p, (size_t)N * (size_t)partial_size(n, a) * 8, s);
        if (e != hipSuccess) rc = hip_fail(e, "hipMallocAsync (partial_fkine0 temporaries)");
        lower[a - 1] = (double *)p;
    }
    // the two specialised launches (register-resident Jacobian, staged Hessian) beat the combined generic tile
    if (rc == RTBHIP_OK) rc = launch_kin(c, ops, (const double *)dq, N, base, tool, 0, nullptr, lower[0], nullptr, s);
    if (rc == RTBHIP_OK && !skip_h) rc = launch_kin(c, ops, (const double *)dq, N, base, tool, 0, nullptr, nullptr, lower[1], s);
    for (int a = 3; a <= order && rc == RTBHIP_OK; ++a)
        rc = launch_partial(n, a, lower, N, a == order ? (double *)dout : lower[a - 1], s);
    for (int a = 1; a < order; ++a)
        if (lower[a - 1]) (void)hipFreeAsync(lower[a - 1], s);
    RTB_TRY(rc);
    if (mem == RTBHIP_MEM_HOST) {
        RTB_HIP(hipStreamSynchronize(s));
        RTB_TRY(fetch(out, dout, obytes));
    }
    return RTBHIP_OK;
}

int rtbhip_ik_lm(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0,
                 int32_t ilimit, int32_t slimit, double tol, int32_t reject_jl, const double *we6,
                 double lambda, int32_t method, int32_t flavour, uint64_t seed, double *q_out,
                 int32_t *success, int32_t *iters, int32_t *searches, double *residual,
                 int32_t mem, void *stream)
{
    return rtbhip_ik_lm_nullspace(chain, Tep, N, q0, ilimit, slimit, tol, reject_jl, we6, lambda, method, flavour, seed,
                                  0.0, 0.0, 0.1, nullptr, q_out, success, iters, searches, residual, mem, stream);
}

static int ik_entry(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0,
                           int32_t ilimit, int32_t slimit, double tol, int32_t reject_jl, const double *we6,
                           double lambda, int32_t method, int32_t flavour, uint64_t seed,
                           double kq, double km, double ps, const double *pi, double ks, double *q_out,
                           int32_t *success, int32_t *iters, int32_t *searches, double *residual,
                           int32_t mem, void *stream);

// restart-generator key of row 0 for the IK calls this thread makes from now on (rtbhip.h)
static thread_local int64_t t_ik_target_base = 0;
int rtbhip_ik_target_base(int64_t base)
{
    if (base < 0) { set_error("ik_target_base: negative base"); return RTBHIP_EINVAL; }
    t_ik_target_base = base;
    return RTBHIP_OK;
}

int rtbhip_ik_lm_nullspace(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0,
                           int32_t ilimit, int32_t slimit, double tol, int32_t reject_jl, const double *we6,
                           double lambda, int32_t method, int32_t flavour, uint64_t seed,
                           double kq, double km, double ps, const double *pi, double *q_out,
                           int32_t *success, int32_t *iters, int32_t *searches, double *residual,
                           int32_t mem, void *stream)
{
    if (method < 0 || method > 4) { set_error("ik_lm: method must be 0 chan, 1 wampler, 2 sugihara, 3 gauss-newton, 4 newton-raphson"); return RTBHIP_EINVAL; }
    return ik_entry(chain, Tep, N, q0, ilimit, slimit, tol, reject_jl, we6, lambda, method, flavour, seed, kq, km, ps, pi, 1.0, q_out, success, iters,
                    searches, residual, mem, stream);
}

int rtbhip_ik_qp(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0, int32_t ilimit, int32_t slimit, double tol,
                 int32_t reject_jl, const double *we6, uint64_t seed, double kj, double ks, double kq, double km, double ps, const double *pi,
                 double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual, int32_t mem, void *stream)
{
    if (!(kj > 0.0) || !(ks > 0.0)) { set_error("ik_qp: kj and ks must be positive (Q must be positive definite)"); return RTBHIP_EINVAL; }
    return ik_entry(chain, Tep, N, q0, ilimit, slimit, tol, reject_jl, we6, kj, 5, 1, seed, kq, km, ps, pi, ks, q_out, success, iters, searches,
                    residual, mem, stream);
}

static int ik_entry(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0,
                           int32_t ilimit, int32_t slimit, double tol, int32_t reject_jl, const double *we6,
                           double lambda, int32_t method, int32_t flavour, uint64_t seed,
                           double kq, double km, double ps, const double *pi, double ks, double *q_out,
                           int32_t *success, int32_t *iters, int32_t *searches, double *residual,
                           int32_t mem, void *stream)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(chain);
    Chain *c = c_owner.get();
    RTB_TRACE("rtbhip_ik_lm");
    if (!c) { set_error("ik_lm: unknown chain handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch("ik_lm", Tep, N, mem, &dscope));
    if (flavour < 0 || flavour > 1) { set_error("ik_lm: flavour must be 0 (ik_LM) or 1 (ikine_LM)"); return RTBHIP_EINVAL; }
    if (ilimit < 1 || slimit < 1) { set_error("ik_lm: ilimit and slimit must be >= 1"); return RTBHIP_EINVAL; }
    if (c->n < 1) { set_error("ik_lm: chain has no joints"); return RTBHIP_EINVAL; }
    if (c->q_width != c->n) { set_error("ik_lm: chain must use jindex 0..n-1 (reference ik.cpp:34-37 assumes the same)"); return RTBHIP_EINVAL; }
    if (N > 0 && (!q_out || !success || !iters || !searches || !residual)) { set_error("ik_lm: NULL output"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    IkParams p;
    p.ilimit = ilimit; p.slimit = slimit; p.reject_jl = reject_jl ? 1 : 0; p.method = method;
    p.flavour = flavour; p.tol = tol; p.lambda = lambda; p.seed = seed;
    p.kq = kq; p.km = km; p.ps = ps; p.ks = ks; p.target0 = t_ik_target_base;
    for (int j = 0; j < RTBHIP_MAX_JOINTS; ++j) p.pi[j] = pi ? pi[j < c->n ? j : (c->n > 0 ? c->n - 1 : 0)] : 0.3;      // NULL: the reference's default
    if (kq > 0.0 && flavour != 1) { set_error("ik_lm: null-space terms belong to the Python solvers (flavour 1)"); return RTBHIP_EINVAL; }
    if (kq > 0.0)
        for (int j = 0; j < c->n && j < RTBHIP_MAX_JOINTS; ++j)
            if (ps == p.pi[j]) { set_error("ik_lm: ps must differ from pi"); return RTBHIP_EINVAL; }
    for (int i = 0; i < 6; i++) p.we[i] = we6 ? we6[i] : 1.0;
    RTB_TRY(ik_check_limits(c, p, N));                 // what the device build refuses, before the device is touched
    DevChain ops;
    const double *qlim = nullptr;
    RTB_TRY(chain_device_ops(c, &ops, &qlim));
    const size_t n = (size_t)c->n;
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_ik(c, ops, qlim, Tep, N, q0, p, q_out, success, iters, searches, residual, (hipStream_t)stream);
    Staging st;
    void *dTep, *dq0, *dq, *ds, *di, *dse, *dr;
    RTB_TRY(st.in(Tep, (size_t)N * 128, &dTep));
    RTB_TRY(st.in(q0, (size_t)N * n * 8, &dq0));
    RTB_TRY(st.out((size_t)N * n * 8, &dq));
    RTB_TRY(st.out((size_t)N * 4, &ds));
    RTB_TRY(st.out((size_t)N * 4, &di));
    RTB_TRY(st.out((size_t)N * 4, &dse));
    RTB_TRY(st.out((size_t)N * 8, &dr));
    RTB_TRY(launch_ik(c, ops, qlim, (const double *)dTep, N, (const double *)dq0, p, (double *)dq, (int32_t *)ds,
                      (int32_t *)di, (int32_t *)dse, (double *)dr, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(q_out, dq, (size_t)N * n * 8));
    RTB_TRY(fetch(success, ds, (size_t)N * 4));
    RTB_TRY(fetch(iters, di, (size_t)N * 4));
    RTB_TRY(fetch(searches, dse, (size_t)N * 4));
    RTB_TRY(fetch(residual, dr, (size_t)N * 8));
    return RTBHIP_OK;
}

int rtbhip_ik_restart(rtbhip_chain_t chain, uint64_t seed, int64_t target, int32_t search, double *q_n)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(chain);
    Chain *c = c_owner.get();
    if (!c || !q_n) { set_error("ik_restart: bad argument"); return RTBHIP_EINVAL; }
    ik_restart_host(c, seed, target, search, q_n);
    return RTBHIP_OK;
}

static int g_rne_pszero = 1;      // rtbhip_tune("rne_pszero", 0): handles created afterwards do not take the p* = 0 shortcut (A/B; same values)
int rtbhip_dyn_create(const double *L24, int32_t n, int32_t mdh, rtbhip_dyn_t *dyn)
{
    if (!L24 || !dyn || n < 1) { set_error("dyn_create: bad argument"); return RTBHIP_EINVAL; }
    if (n > RTBHIP_MAX_JOINTS) { set_error("dyn_create: more than RTBHIP_MAX_JOINTS links"); return RTBHIP_ELIMIT; }
    if (mdh != 0 && mdh != 1) { set_error("dyn_create: mdh must be 0 or 1"); return RTBHIP_EINVAL; }
    std::shared_ptr<Dyn> d(new Dyn());
    d->n = n;
    d->mdh = mdh;
    d->links.resize(n);
    for (int i = 0; i < n; i++) {
        const double *l = L24 + 24 * i;  // layout: DHRobot.py:1342-1358
        DevLink &k = d->links[i];
        std::memset(&k, 0, sizeof k);
        int sigma = (int)l[4];  // frne.c:279 casts the double to the enum the same way
        if (sigma != 0 && sigma != 1) { set_error("dyn_create: sigma must be 0 (R) or 1 (P)"); return RTBHIP_EINVAL; }
        k.sa = sin(l[0]); k.ca = cos(l[0]);  // frne.c:325-326 evaluates these per call; constant per link
        k.a = l[1]; k.theta = l[2]; k.d = l[3]; k.sigma = sigma; k.offset = l[5];
        k.m = l[6]; k.rx = l[7]; k.ry = l[8]; k.rz = l[9];
        for (int j = 0; j < 9; j++) k.I[j] = l[10 + j];
        k.Jm = l[19]; k.G = l[20]; k.B = l[21]; k.Tc0 = l[22]; k.Tc1 = l[23];
        k.gjm = k.G * k.G * k.Jm; k.gb = k.G * k.G * k.B; k.ag = fabs(k.G);
        k.flags = 0;
        if (k.rx == 0.0 && k.ry == 0.0 && k.rz == 0.0) k.flags |= kLinkRZero;
        if (k.I[1] == 0.0 && k.I[2] == 0.0 && k.I[3] == 0.0 && k.I[5] == 0.0 && k.I[6] == 0.0 && k.I[7] == 0.0) k.flags |= kLinkIDiag;
        if (g_rne_pszero && k.sigma == 0 && k.a == 0.0 && k.d == 0.0) k.flags |= kLinkPsZero;      // p* = 0 (DH Panda links 2 and 6, Puma560 link 5 and 6)
    }
    uint64_t h = g_next.fetch_add(1);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    // a table without a built-in instantiation: ask for its own now (worker thread; nobody waits), the first launches take the general kernels
    jit_request_dyn(d.get(), false, false);
    g_dyns[h] = std::move(d);
    *dyn = h;
    return RTBHIP_OK;
}

int rtbhip_dyn_destroy(rtbhip_dyn_t dyn)
{
    std::shared_ptr<Dyn> d;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_dyns.find(dyn);
        if (it == g_dyns.end()) { set_error("dyn_destroy: unknown handle"); return RTBHIP_EINVAL; }
        d = std::move(it->second);
        g_dyns.erase(it);
    }
    return RTBHIP_OK;
}

static int rne_entry(const char *fn, rtbhip_dyn_t dyn, const double *q, const double *qd, const double *qdd, int64_t N,
                     const double *grav3, const double *fext6, double *tau, double *wbase, bool want_wbase, int32_t mem, void *stream)
{
    const std::shared_ptr<Dyn> d_owner = dyn_from_handle(dyn);
    Dyn *d = d_owner.get();
    if (!d) { set_error(std::string(fn) + ": unknown dyn handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch(fn, q, N, mem, &dscope));
    if (!grav3) { set_error(std::string(fn) + ": NULL gravity"); return RTBHIP_EINVAL; }
    if (N > 0 && !tau) { set_error(std::string(fn) + ": NULL tau"); return RTBHIP_EINVAL; }   // qd / qdd may be NULL (= zeros)
    if (want_wbase && N > 0 && !wbase) { set_error(std::string(fn) + ": NULL wbase"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    const DevLink *links = nullptr;
    RTB_TRY(dyn_device_links(d, &links));
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_rne(d, links, q, qd, qdd, N, grav3, fext6, tau, (hipStream_t)stream, wbase);
    HostIO io;
    const size_t row = (size_t)d->n * 8;
    io.add_in(q, row); io.add_in(qd, row); io.add_in(qdd, row);
    io.add_out(tau, row);
    if (want_wbase) io.add_out(wbase, 6 * 8);
    return host_pipeline(io, N, [&](const void *const *din, void *const *dout, int64_t, int64_t rows, hipStream_t s) {
        return launch_rne(d, links, (const double *)din[0], (const double *)din[1], (const double *)din[2], rows, grav3, fext6, (double *)dout[0], s,
                          want_wbase ? (double *)dout[1] : nullptr);
    });
}

int rtbhip_rne(rtbhip_dyn_t dyn, const double *q, const double *qd, const double *qdd, int64_t N,
               const double *grav3, const double *fext6, double *tau, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_rne");
    return rne_entry("rne", dyn, q, qd, qdd, N, grav3, fext6, tau, nullptr, false, mem, stream);
}

int rtbhip_rne_base_wrench(rtbhip_dyn_t dyn, const double *q, const double *qd, const double *qdd, int64_t N,
                           const double *grav3, const double *fext6, double *tau, double *wbase, int32_t mem, void *stream)
{
    RTB_TRACE("rtbhip_rne_base_wrench");
    return rne_entry("rne_base_wrench", dyn, q, qd, qdd, N, grav3, fext6, tau, wbase, true, mem, stream);
}

int rtbhip_tree_create(const rtbhip_tree_group *groups, int32_t ng, rtbhip_tree_t *tree)
{
    if (!tree) { set_error("tree_create: NULL handle pointer"); return RTBHIP_EINVAL; }
    std::shared_ptr<Tree> t(new Tree());
    RTB_TRY(compile_tree(groups, ng, t.get()));
    jit_request_tree(t.get(), false, false);
    const uint64_t h = g_next.fetch_add(1);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_trees[h] = std::move(t);
    *tree = h;
    return RTBHIP_OK;
}

int rtbhip_tree_destroy(rtbhip_tree_t tree)
{
    std::shared_ptr<Tree> t;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = g_trees.find(tree);
        if (it == g_trees.end()) { set_error("tree_destroy: unknown handle"); return RTBHIP_EINVAL; }
        t = std::move(it->second);
        g_trees.erase(it);
    }
    return RTBHIP_OK;
}

int rtbhip_tree_rne(rtbhip_tree_t tree, const double *q, const double *qd, const double *qdd, int64_t N,
                    const double *gravity3, double *tau, int32_t mem, void *stream)
{
    const std::shared_ptr<Tree> t_owner = tree_from_handle(tree);
    Tree *t = t_owner.get();
    RTB_TRACE("rtbhip_tree_rne");
    if (!t) { set_error("tree_rne: unknown tree handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch("tree_rne", q, N, mem, &dscope));
    if (!gravity3) { set_error("tree_rne: NULL gravity"); return RTBHIP_EINVAL; }
    if (N > 0 && !tau) { set_error("tree_rne: NULL tau"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    const DevGroup *groups = nullptr;
    RTB_TRY(tree_device_groups(t, &groups));
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_tree_rne(t, groups, q, qd, qdd, N, gravity3, tau, (hipStream_t)stream);
    Staging st;
    const size_t bytes = (size_t)N * t->n * 8;
    void *dq, *dqd, *dqdd, *dtau;
    RTB_TRY(st.in(q, bytes, &dq));
    RTB_TRY(st.in(qd, bytes, &dqd));
    RTB_TRY(st.in(qdd, bytes, &dqdd));
    RTB_TRY(st.out(bytes, &dtau));
    RTB_TRY(launch_tree_rne(t, groups, (const double *)dq, (const double *)dqd, (const double *)dqdd, N, gravity3,
                            (double *)dtau, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(tau, dtau, bytes));
    return RTBHIP_OK;
}

/* Dynamics.inertia / coriolis / accel (robot/Dynamics.py:704-861, 424-509) */
static int dyn_entry(const char *fn, rtbhip_dyn_t dyn, int mode, const double *q, const double *qd, const double *tq,
                     int64_t N, const double *grav3, double *out, int32_t mem, void *stream)
{
    const std::shared_ptr<Dyn> d_owner = dyn_from_handle(dyn);
    Dyn *d = d_owner.get();
    RTB_TRACE((std::string("rtbhip_") + fn).c_str());
    if (!d) { set_error(std::string(fn) + ": unknown dyn handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch(fn, q, N, mem, &dscope));
    if (N > 0 && !out) { set_error(std::string(fn) + ": NULL output"); return RTBHIP_EINVAL; }
    if (N > 0 && mode >= 1 && !qd) { set_error(std::string(fn) + ": NULL qd"); return RTBHIP_EINVAL; }
    if (N > 0 && mode == 2 && (!tq || !grav3)) { set_error(std::string(fn) + ": NULL torque/gravity"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    const DevLink *links = nullptr;
    RTB_TRY(dyn_device_links(d, &links));
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_dyn(d, links, mode, q, qd, tq, N, grav3, out, (hipStream_t)stream);
    Staging st;
    const size_t n = (size_t)d->n, bytes = (size_t)N * n * 8, obytes = mode == 2 ? bytes : bytes * n;
    void *dq, *dqd = nullptr, *dtq = nullptr, *dout;
    RTB_TRY(st.in(q, bytes, &dq));
    if (mode >= 1) RTB_TRY(st.in(qd, bytes, &dqd));
    if (mode == 2) RTB_TRY(st.in(tq, bytes, &dtq));
    RTB_TRY(st.out(obytes, &dout));
    RTB_TRY(launch_dyn(d, links, mode, (const double *)dq, (const double *)dqd, (const double *)dtq, N, grav3,
                       (double *)dout, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(out, dout, obytes));
    return RTBHIP_OK;
}

int rtbhip_inertia(rtbhip_dyn_t dyn, const double *q, int64_t N, double *M, int32_t mem, void *stream)
{
    return dyn_entry("inertia", dyn, 0, q, nullptr, nullptr, N, nullptr, M, mem, stream);
}

int rtbhip_coriolis(rtbhip_dyn_t dyn, const double *q, const double *qd, int64_t N, double *Cm, int32_t mem, void *stream)
{
    return dyn_entry("coriolis", dyn, 1, q, qd, nullptr, N, nullptr, Cm, mem, stream);
}

int rtbhip_accel(rtbhip_dyn_t dyn, const double *q, const double *qd, const double *torque, int64_t N,
                 const double *grav3, double *qdd, int32_t mem, void *stream)
{
    return dyn_entry("accel", dyn, 2, q, qd, torque, N, grav3, qdd, mem, stream);
}

/* the same terms for an ETS robot (link tree): Dynamics.inertia / coriolis / accel over Robot.rne */
static int tree_dyn_entry(const char *fn, rtbhip_tree_t tree, int mode, const double *q, const double *qd, const double *tq,
                          int64_t N, const double *grav3, double *out, int32_t mem, void *stream)
{
    const std::shared_ptr<Tree> t_owner = tree_from_handle(tree);
    Tree *t = t_owner.get();
    RTB_TRACE((std::string("rtbhip_") + fn).c_str());
    if (!t) { set_error(std::string(fn) + ": unknown tree handle"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch(fn, q, N, mem, &dscope));
    if (N > 0 && !out) { set_error(std::string(fn) + ": NULL output"); return RTBHIP_EINVAL; }
    if (N > 0 && mode >= 1 && !qd) { set_error(std::string(fn) + ": NULL qd"); return RTBHIP_EINVAL; }
    if (N > 0 && mode == 2 && (!tq || !grav3)) { set_error(std::string(fn) + ": NULL torque/gravity"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    const DevGroup *groups = nullptr;
    RTB_TRY(tree_device_groups(t, &groups));
    if (mem == RTBHIP_MEM_DEVICE)
        return launch_tree_dyn(t, groups, mode, q, qd, tq, N, grav3, out, (hipStream_t)stream);
    Staging st;
    const size_t n = (size_t)t->n, bytes = (size_t)N * n * 8, obytes = mode == 2 ? bytes : bytes * n;
    void *dq, *dqd = nullptr, *dtq = nullptr, *dout;
    RTB_TRY(st.in(q, bytes, &dq));
    if (mode >= 1) RTB_TRY(st.in(qd, bytes, &dqd));
    if (mode == 2) RTB_TRY(st.in(tq, bytes, &dtq));
    RTB_TRY(st.out(obytes, &dout));
    RTB_TRY(launch_tree_dyn(t, groups, mode, (const double *)dq, (const double *)dqd, (const double *)dtq, N, grav3,
                            (double *)dout, nullptr));
    RTB_HIP(hipDeviceSynchronize());
    RTB_TRY(fetch(out, dout, obytes));
    return RTBHIP_OK;
}

int rtbhip_tree_inertia(rtbhip_tree_t tree, const double *q, int64_t N, double *M, int32_t mem, void *stream)
{
    return tree_dyn_entry("tree_inertia", tree, 0, q, nullptr, nullptr, N, nullptr, M, mem, stream);
}

int rtbhip_tree_coriolis(rtbhip_tree_t tree, const double *q, const double *qd, int64_t N, double *Cm, int32_t mem, void *stream)
{
    return tree_dyn_entry("tree_coriolis", tree, 1, q, qd, nullptr, N, nullptr, Cm, mem, stream);
}

int rtbhip_tree_accel(rtbhip_tree_t tree, const double *q, const double *qd, const double *torque, int64_t N,
                      const double *gravity3, double *qdd, int32_t mem, void *stream)
{
    return tree_dyn_entry("tree_accel", tree, 2, q, qd, torque, N, gravity3, qdd, mem, stream);
}

static int fleet_entry(const rtbhip_chain_t *chains, int32_t n_chains, const double *const *q,
                       const int64_t *N, int32_t frame, double *const *T, double *const *J,
                       int32_t mem, void *stream, bool packed)
{
    // packed: T[c] is the (N[c], 16 + 6 n_c) array of [T | J] rows, J is not used
    if (n_chains < 0 || (n_chains > 0 && (!chains || !q || !N || !T || (!packed && !J)))) { set_error("fleet: bad argument"); return RTBHIP_EINVAL; }
    RTB_TRACE(packed ? "rtbhip_fleet_fkine_jacob_packed" : "rtbhip_fleet_fkine_jacob");
    if (frame != 0 && frame != 1) { set_error("fleet: frame must be 0 or 1"); return RTBHIP_EINVAL; }
    if (mem != RTBHIP_MEM_HOST && mem != RTBHIP_MEM_DEVICE) { set_error("fleet: bad mem kind"); return RTBHIP_EINVAL; }
    std::vector<FleetEntry> entries;
    Staging st;
    DeviceScope dscope;
    if (mem == RTBHIP_MEM_DEVICE)
        for (int i = 0; i < n_chains; i++)
            if (N[i] > 0 && q[i]) { RTB_TRY(dscope.enter_for("fleet", q[i])); break; }
    std::vector<void *> dT(n_chains, nullptr), dJ(n_chains, nullptr);
    int64_t tile0 = 0;
    for (int i = 0; i < n_chains; i++) {
        const std::shared_ptr<Chain> c_owner = chain_from_handle(chains[i]);
        Chain *c = c_owner.get();
        if (!c) { set_error("fleet: unknown chain handle"); return RTBHIP_EINVAL; }
        if (N[i] < 0) { set_error("fleet: negative N"); return RTBHIP_EINVAL; }
        if (N[i] == 0) continue;
        if (!q[i] || !T[i] || (!packed && !J[i])) { set_error("fleet: NULL buffer"); return RTBHIP_EINVAL; }
        FleetEntry e;
        RTB_TRY(chain_device_ops(c, &e.dc, nullptr));
        e.n = c->n; e.q_width = c->q_width; e.N = N[i]; e.tile0 = tile0;
        e.stride = 0; e.pad = 0;
        if (mem == RTBHIP_MEM_DEVICE) {
            e.q = q[i]; e.T = T[i]; e.J = packed ? nullptr : J[i];
        } else {
            void *dq;
            RTB_TRY(st.in(q[i], (size_t)N[i] * c->q_width * 8, &dq));
            RTB_TRY(st.out((size_t)N[i] * (packed ? 128 + 48 * c->n : 128), &dT[i]));
            if (!packed) RTB_TRY(st.out((size_t)N[i] * 48 * c->n, &dJ[i]));
            e.q = (const double *)dq; e.T = (double *)dT[i]; e.J = (double *)dJ[i];
        }
        tile0 += (N[i] + 63) / 64;
        entries.push_back(e);
    }
    if (entries.empty()) return RTBHIP_OK;
    RTB_TRY(launch_fleet(entries, frame, mem == RTBHIP_MEM_DEVICE ? (hipStream_t)stream : nullptr, packed));
    if (mem == RTBHIP_MEM_HOST) {
        RTB_HIP(hipDeviceSynchronize());
        for (int i = 0; i < n_chains; i++) {
            if (N[i] == 0) continue;
            const std::shared_ptr<Chain> c_owner = chain_from_handle(chains[i]);
            Chain *c = c_owner.get();
            RTB_TRY(fetch(T[i], dT[i], (size_t)N[i] * (packed ? 128 + 48 * c->n : 128)));
            if (!packed) RTB_TRY(fetch(J[i], dJ[i], (size_t)N[i] * 48 * c->n));
        }
    }
    return RTBHIP_OK;
}

int rtbhip_fleet_fkine_jacob(const rtbhip_chain_t *chains, int32_t n_chains, const double *const *q,
                             const int64_t *N, int32_t frame, double *const *T, double *const *J,
                             int32_t mem, void *stream)
{
    return fleet_entry(chains, n_chains, q, N, frame, T, J, mem, stream, false);
}

int rtbhip_fleet_fkine_jacob_packed(const rtbhip_chain_t *chains, int32_t n_chains, const double *const *q,
                                    const int64_t *N, int32_t frame, double *const *TJ, int32_t mem, void *stream)
{
    return fleet_entry(chains, n_chains, q, N, frame, TJ, nullptr, mem, stream, true);
}

int rtbhip_host_alloc(uint64_t bytes, void **ptr)
{
    if (!ptr) { set_error("host_alloc: NULL out"); return RTBHIP_EINVAL; }
    return host_alloc((size_t)bytes, ptr);
}

int rtbhip_host_free(void *ptr) { return host_free(ptr); }

int rtbhip_shard_range(int64_t N, int32_t rank, int32_t world, int64_t *begin, int64_t *count)
{
    if (N < 0 || world < 1 || rank < 0 || rank >= world || !begin || !count) { set_error("shard_range: bad argument"); return RTBHIP_EINVAL; }
    int64_t base = N / world, extra = N % world;
    *count = base + (rank < extra ? 1 : 0);
    *begin = base * rank + (rank < extra ? rank : extra);
    return RTBHIP_OK;
}

int rtbhip_last_launch(int32_t *grid, int32_t *block, int32_t *lds_bytes)
{
    if (grid) *grid = g_last_launch[0];
    if (block) *block = g_last_launch[1];
    if (lds_bytes) *lds_bytes = g_last_launch[2];
    return RTBHIP_OK;
}

int rtbhip_stream_probe(const double *src, int64_t read_doubles, double *dst, int64_t write_doubles, void *stream)
{
    if ((read_doubles > 0 && !src) || (write_doubles > 0 && !dst) || read_doubles < 0 || write_doubles < 0) { set_error("stream_probe: bad argument"); return RTBHIP_EINVAL; }
    DeviceScope dscope;
    RTB_TRY(check_batch("stream_probe", dst ? dst : src, 1, RTBHIP_MEM_DEVICE, &dscope));
    return launch_stream_probe(src, read_doubles, dst, write_doubles, (hipStream_t)stream);
}

int rtbhip_tune(const char *key, int32_t value)
{
    if (!key) { set_error("tune: NULL key"); return RTBHIP_EINVAL; }
    kin_tune(key, value);
    rne_tune(key, value);
    ik_tune(key, value);
    partial_tune(key, value);
    if (std::string(key) == "rne_pszero") g_rne_pszero = value != 0;
    hostpipe_tune(key, value);
    shard_tune(key, value);
    tree_tune(key, value);
    jit_tune(key, value);
    return RTBHIP_OK;
}

// ---- run-time instantiation (jit.cpp)
int rtbhip_jit_stats(rtbhip_jit_info *out)
{
    if (!out) { set_error("jit_stats: NULL out"); return RTBHIP_EINVAL; }
    jit_stats(out);
    return RTBHIP_OK;
}
int rtbhip_jit_wait(double timeout_s) { return jit_wait(timeout_s); }
int rtbhip_jit_compile(const char *unit, const char *expr, const char *arch, int64_t *code_bytes, double *seconds, int32_t *from_disk)
{
    if (!unit || !expr || !arch) { set_error("jit_compile: NULL argument"); return RTBHIP_EINVAL; }
    size_t cb = 0;
    double sec = 0.0;
    int fd = 0;
    // "unit" may carry generated source after a newline (a tree's knowledge type): "tree_kernels.hip\nnamespace rtbhip { struct ... }"
    const std::string u = unit;
    const size_t nl = u.find('\n');
    const std::string file = nl == std::string::npos ? u : u.substr(0, nl), pre = nl == std::string::npos ? std::string() : u.substr(nl + 1);
    const int rc = jit_compile_now(file.c_str(), expr, arch, &cb, &sec, &fd, pre.c_str());
    if (code_bytes) *code_bytes = (int64_t)cb;
    if (seconds) *seconds = sec;
    if (from_disk) *from_disk = fd;
    return rc;
}
int rtbhip_jit_prepare(int32_t kind, uint64_t handle)
{
    if (kind == 0) { auto c = chain_from_handle(handle); if (!c) return RTBHIP_EINVAL; jit_request_chain(c.get(), true); }
    else if (kind == 1) { auto d = dyn_from_handle(handle); if (!d) return RTBHIP_EINVAL; jit_request_dyn(d.get(), true, true); }
    else if (kind == 2) { auto t = tree_from_handle(handle); if (!t) return RTBHIP_EINVAL; jit_request_tree(t.get(), true, true); }
    else { set_error("jit_prepare: kind must be 0 (chain), 1 (dyn) or 2 (tree)"); return RTBHIP_EINVAL; }
    return RTBHIP_OK;
}
int rtbhip_jit_names(int32_t kind, uint64_t handle, char *buf, int64_t cap)
{
    if (!buf || cap < 1) { set_error("jit_names: bad buffer"); return RTBHIP_EINVAL; }
    std::vector<std::string> names;
    std::string pre;
    if (kind == 0) { auto c = chain_from_handle(handle); if (!c) return RTBHIP_EINVAL; names = ik_jit_names(c.get()); }
    else if (kind == 1) { auto d = dyn_from_handle(handle); if (!d) return RTBHIP_EINVAL; names = rne_jit_names(d.get()); }
    else if (kind == 2) {
        auto t = tree_from_handle(handle);
        if (!t) return RTBHIP_EINVAL;
        names = tree_jit_names(t.get());
        if (!names.empty()) { std::string tn; pre = tree_jit_knowledge(t.get(), &tn); }
    } else { set_error("jit_names: kind must be 0 (chain), 1 (dyn) or 2 (tree)"); return RTBHIP_EINVAL; }
    std::string all;
    for (const std::string &n : names) all += n + "\n";
    if (!pre.empty()) all += "\f" + pre;            // after a form feed: the generated knowledge type the tree's expressions refer to
    std::snprintf(buf, (size_t)cap, "%s", all.c_str());
    return RTBHIP_OK;
}

}  // extern "C"
