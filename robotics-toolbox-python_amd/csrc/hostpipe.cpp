// hostpipe.cpp -- the RTBHIP_MEM_HOST boundary: host arrays in, host arrays out, over PCIe.
//
// The reference's boundary hands over NumPy arrays (host memory), so this is what `panda.fkine(numpy_q)` pays for.  Round 1
// staged every call through fresh hipMalloc buffers and pageable hipMemcpy: 22-41 ms per 1e6 Panda configurations, 13-24 GB/s,
// against 0.08 ms of kernel time.  Here:
//   * rows are processed in chunks (~32 MB of traffic each) that alternate between two SLOTS, each with its own stream,
//     device buffers and pinned staging buffers -- all persistent, grown on demand, never allocated per call;
//   * per chunk: inputs -> (pinned staging, several copy threads) -> H2D -> kernel -> D2H, so the H2D / kernel of one slot
//     overlaps the D2H of the other and the PCIe link (the bound: 464 of the 520 bytes per configuration leave the device)
//     never waits for the host;
//   * memory that is ALREADY pinned (hipHostMalloc / hipHostRegister; rtbhip_host_alloc hands it out and the Python shim
//     allocates its result arrays from it) is the DMA endpoint itself: no staging copy, the results land in the caller's array;
//   * pageable results go through the slot's pinned buffer and are copied out by the copy threads while the next chunk
//     is in flight.
// Nothing here computes: kernels are launched through the caller's functor on device pointers.
#include "rtbhip_internal.h"
#include <algorithm>
#include <array>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <unistd.h>

namespace rtbhip {

// ---------------------------------------------------------------- copy threads
namespace {
class CopyPool {
public:
    static CopyPool &get() { static CopyPool p; return p; }
    // dst/src host memory; splits the range over the workers and waits
    void copy(void *dst, const void *src, size_t bytes)
    {
        // worker threads do not survive fork(): in a child the vector is non-empty but nobody would ever take a job
        if (bytes < (1u << 20) || workers_.empty() || getpid() != owner_) { std::memcpy(dst, src, bytes); return; }
        const size_t parts = workers_.size() + 1;
        const size_t step = ((bytes / parts) + 4095) & ~(size_t)4095;
        std::unique_lock<std::mutex> lk(mu_);
        pending_ = 0;
        size_t off = step;                                  // the caller copies the first part itself
        for (size_t w = 0; w < workers_.size() && off < bytes; ++w, off += step) {
            jobs_.push_back({(char *)dst + off, (const char *)src + off, std::min(step, bytes - off)});
            ++pending_;
        }
        lk.unlock();
        cv_.notify_all();
        std::memcpy(dst, src, std::min(step, bytes));
        lk.lock();
        done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    struct Job { char *dst; const char *src; size_t n; };
    CopyPool()
    {
        owner_ = getpid();
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 16 ? 7 : (hw >= 4 ? (int)hw / 2 - 1 : 0);
        if (const char *e = std::getenv("RTBHIP_COPY_THREADS")) n = std::max(0, std::atoi(e) - 1);
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool()
    {
        if (getpid() != owner_) { for (auto &t : workers_) t.detach(); return; }      // a forked child: no threads to join
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void run()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
            if (stop_) return;
            Job j = jobs_.back();
            jobs_.pop_back();
            lk.unlock();
            std::memcpy(j.dst, j.src, j.n);
            lk.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<Job> jobs_;
    std::vector<std::thread> workers_;
    int pending_ = 0;
    bool stop_ = false;
    pid_t owner_ = 0;
};

bool is_pinned_host(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

size_t g_chunk_bytes = [] {                 // traffic per chunk; rtbhip_tune("host_chunk_kb", ...) for tests and A/B runs
    const char *e = std::getenv("RTBHIP_HOST_CHUNK_MB");
    return (size_t)(e && std::atoi(e) > 0 ? std::atoi(e) : 32) << 20;
}();
constexpr size_t kAlign = 256;
size_t up(size_t x) { return (x + kAlign - 1) & ~(kAlign - 1); }

// ---------------------------------------------------------------- persistent slots
struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    void *dev = nullptr, *pin = nullptr;     // device buffer [inputs | outputs]; pinned staging of the same layout
    size_t dev_cap = 0, pin_cap = 0;
};
struct DeviceSlots {
    std::mutex mu;                           // one host-path call at a time PER DEVICE: calls on different GPUs overlap
    std::array<Slot, 2> slots;
};
struct Pipe {
    std::mutex mu;                           // guards the map only
    std::map<int, std::unique_ptr<DeviceSlots>> per_device;
};
Pipe &pipe() { static Pipe &p = *new Pipe(); return p; }                  // never destroyed

int slot_reserve(Slot &s, size_t dev_bytes, size_t pin_bytes)
{
    if (!s.stream) {
        RTB_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        RTB_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (dev_bytes > s.dev_cap) {
        if (s.dev) (void)hipFree(s.dev);
        s.dev = nullptr; s.dev_cap = 0;
        RTB_HIP(hipMalloc(&s.dev, dev_bytes));
        s.dev_cap = dev_bytes;
    }
    if (pin_bytes > s.pin_cap) {
        if (s.pin) (void)hipHostFree(s.pin);
        s.pin = nullptr; s.pin_cap = 0;
        RTB_HIP(hipHostMalloc(&s.pin, pin_bytes, hipHostMallocDefault));
        s.pin_cap = pin_bytes;
    }
    return RTBHIP_OK;
}
}  // namespace

void hostpipe_tune(const char *key, int value)
{
    if (std::string(key) == "host_chunk_kb" && value > 0) g_chunk_bytes = (size_t)value << 10;
}

void hostpipe_release()
{
    Pipe &p = pipe();
    std::lock_guard<std::mutex> lk(p.mu);
    for (auto &kv : p.per_device) {
        std::lock_guard<std::mutex> l2(kv.second->mu);
        for (Slot &s : kv.second->slots) {
            if (s.dev) (void)hipFree(s.dev);
            if (s.pin) (void)hipHostFree(s.pin);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            s = Slot();
        }
    }
}

int host_pipeline(const HostIO &io, int64_t N, const ChunkLaunch &launch)
{
    if (N <= 0) return RTBHIP_OK;
    size_t row_bytes = 0;
    for (int i = 0; i < io.n_in; ++i) if (io.in[i]) row_bytes += io.in_row[i];
    for (int i = 0; i < io.n_out; ++i) if (io.out[i]) row_bytes += io.out_row[i];
    if (row_bytes == 0) return RTBHIP_OK;
    int64_t rows_chunk = (int64_t)(g_chunk_bytes / row_bytes);
    rows_chunk = std::max<int64_t>(rows_chunk & ~(int64_t)63, 64);
    if (rows_chunk > N) rows_chunk = N;
    // layout of one slot buffer: every used input, then every used output, each rounded up to 256 bytes
    size_t off_in[HostIO::kMax], off_out[HostIO::kMax], total = 0, pin_total = 0;
    bool pin_in[HostIO::kMax], pin_out[HostIO::kMax];
    for (int i = 0; i < io.n_in; ++i) {
        off_in[i] = total;
        pin_in[i] = is_pinned_host(io.in[i]);
        if (io.in[i]) { total += up((size_t)rows_chunk * io.in_row[i]); }
    }
    for (int i = 0; i < io.n_out; ++i) {
        off_out[i] = total;
        pin_out[i] = is_pinned_host(io.out[i]);
        if (io.out[i]) { total += up((size_t)rows_chunk * io.out_row[i]); }
    }
    // the staging buffer mirrors the device layout (simple), but is only needed if something is pageable
    bool need_pin = false;
    for (int i = 0; i < io.n_in; ++i) need_pin = need_pin || (io.in[i] && !pin_in[i]);
    for (int i = 0; i < io.n_out; ++i) need_pin = need_pin || (io.out[i] && !pin_out[i]);
    pin_total = need_pin ? total : 0;

    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    Pipe &P = pipe();
    DeviceSlots *ds = nullptr;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto &up_ = P.per_device[dev];
        if (!up_) up_.reset(new DeviceSlots());
        ds = up_.get();                      // entries are never erased: the pointer stays valid
    }
    std::lock_guard<std::mutex> lk(ds->mu);
    std::array<Slot, 2> &slots = ds->slots;
    for (Slot &s : slots) { int rc = slot_reserve(s, total, pin_total); if (rc != RTBHIP_OK) return rc; }

    const int64_t nchunks = (N + rows_chunk - 1) / rows_chunk;
    CopyPool &cp = CopyPool::get();
    auto finish = [&](int64_t k) -> int {       // chunk k's D2H has been issued on its slot: wait for it, copy pageable results out
        Slot &s = slots[k & 1];
        RTB_HIP(hipEventSynchronize(s.done));
        const int64_t row0 = k * rows_chunk, rows = std::min(rows_chunk, N - row0);
        for (int i = 0; i < io.n_out; ++i)
            if (io.out[i] && !pin_out[i])
                cp.copy((char *)io.out[i] + (size_t)row0 * io.out_row[i], (char *)s.pin + off_out[i], (size_t)rows * io.out_row[i]);
        return RTBHIP_OK;
    };
    int rc = RTBHIP_OK;
    int64_t issued = 0;
    for (int64_t k = 0; k < nchunks && rc == RTBHIP_OK; ++k) {
        Slot &s = slots[k & 1];
        if (k >= 2) rc = finish(k - 2);          // the slot's previous chunk must have left its buffers
        if (rc != RTBHIP_OK) break;
        const int64_t row0 = k * rows_chunk, rows = std::min(rows_chunk, N - row0);
        const void *din[HostIO::kMax];
        void *dout[HostIO::kMax];
        for (int i = 0; i < io.n_in && rc == RTBHIP_OK; ++i) {
            din[i] = nullptr;
            if (!io.in[i]) continue;
            const size_t bytes = (size_t)rows * io.in_row[i];
            const char *src = (const char *)io.in[i] + (size_t)row0 * io.in_row[i];
            if (!pin_in[i]) {
                cp.copy((char *)s.pin + off_in[i], src, bytes);
                src = (const char *)s.pin + off_in[i];
            }
            hipError_t e = hipMemcpyAsync((char *)s.dev + off_in[i], src, bytes, hipMemcpyHostToDevice, s.stream);
            if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync (host path, H2D)");
            din[i] = (char *)s.dev + off_in[i];
        }
        for (int i = 0; i < io.n_out; ++i) dout[i] = io.out[i] ? (char *)s.dev + off_out[i] : nullptr;
        if (rc == RTBHIP_OK) rc = launch(din, dout, row0, rows, s.stream);
        for (int i = 0; i < io.n_out && rc == RTBHIP_OK; ++i) {
            if (!io.out[i]) continue;
            const size_t bytes = (size_t)rows * io.out_row[i];
            void *dst = pin_out[i] ? (void *)((char *)io.out[i] + (size_t)row0 * io.out_row[i]) : (void *)((char *)s.pin + off_out[i]);
            hipError_t e = hipMemcpyAsync(dst, (char *)s.dev + off_out[i], bytes, hipMemcpyDeviceToHost, s.stream);
            if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync (host path, D2H)");
        }
        if (rc == RTBHIP_OK) {
            hipError_t e = hipEventRecord(s.done, s.stream);
            if (e != hipSuccess) rc = hip_fail(e, "hipEventRecord (host path)");
        }
        if (rc == RTBHIP_OK) issued = k + 1;
    }
    // drain: the last one or two chunks (also after an error, so that no DMA is left writing into the caller's arrays)
    for (int64_t k = std::max<int64_t>(0, issued - 2); k < issued; ++k) {
        int r2 = finish(k);
        if (rc == RTBHIP_OK) rc = r2;
    }
    if (rc != RTBHIP_OK) { (void)hipStreamSynchronize(slots[0].stream); (void)hipStreamSynchronize(slots[1].stream); }
    return rc;
}

// ---------------------------------------------------------------- pinned host memory for callers (result arrays)
namespace {
struct HostCache {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;      // size -> block
    std::map<void *, size_t> live;                  // handed out
    size_t cached = 0;
    size_t cap = [] {
        const char *e = std::getenv("RTBHIP_PINNED_CACHE_MB");
        return (size_t)(e && std::atoll(e) >= 0 ? std::atoll(e) : 1024) << 20;
    }();
};
HostCache &hcache() { static HostCache &c = *new HostCache(); return c; }   // never destroyed (no hipHostFree at exit)
}  // namespace

int host_alloc(size_t bytes, void **out)
{
    *out = nullptr;
    if (bytes == 0) return RTBHIP_OK;
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    HostCache &c = hcache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_blocks.lower_bound(want);
        if (it != c.free_blocks.end() && it->first <= want + want / 4) {     // pinning is what costs: reuse a block up to 25 % larger
            *out = it->second;
            c.live[*out] = it->first;
            c.cached -= it->first;
            c.free_blocks.erase(it);
            return RTBHIP_OK;
        }
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        host_cache_trim(0);                                                 // give cached blocks back and try once more
        e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc (rtbhip_host_alloc)");
    }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[p] = want;
    *out = p;
    return RTBHIP_OK;
}

int host_free(void *p)
{
    if (!p) return RTBHIP_OK;
    HostCache &c = hcache();
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it == c.live.end()) { set_error("host_free: not a block of rtbhip_host_alloc"); return RTBHIP_EINVAL; }
        sz = it->second;
        c.live.erase(it);
        if (c.cached + sz <= c.cap) {
            c.free_blocks.emplace(sz, p);
            c.cached += sz;
            return RTBHIP_OK;
        }
    }
    (void)hipHostFree(p);
    return RTBHIP_OK;
}

void host_cache_trim(size_t keep_bytes)
{
    HostCache &c = hcache();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        while (c.cached > keep_bytes && !c.free_blocks.empty()) {
            auto it = std::prev(c.free_blocks.end());
            c.cached -= it->first;
            drop.push_back(it->second);
            c.free_blocks.erase(it);
        }
    }
    for (void *p : drop) (void)hipHostFree(p);
}

// ---------------------------------------------------------------- cached device buffers for the other host-path calls
// IK, the dynamics terms, hessian_from_jacobian, the fleet ... stage their host arrays in blocks drawn from here instead of a
// hipMalloc per call.  The cache is BOUNDED: idle blocks above RTBHIP_DEVICE_CACHE_MB (default 512) per device go straight back to
// the driver when they are released, so one large call (hessian_from_jacobian at N = 1e6 needs 2.3 GB) does not leave its buffers
// parked where torch's or anybody else's allocator cannot reach them; rtbhip_trim() empties it on request.
namespace {
struct DevCache {
    std::mutex mu;
    std::map<int, std::multimap<size_t, void *>> free_blocks;   // per device
    std::map<int, size_t> cached;                               // idle bytes per device
    std::map<void *, std::pair<int, size_t>> live;
    size_t cap = [] {
        const char *e = std::getenv("RTBHIP_DEVICE_CACHE_MB");
        return (size_t)(e && std::atoll(e) >= 0 ? std::atoll(e) : 512) << 20;
    }();
};
DevCache &dcache() { static DevCache &c = *new DevCache(); return c; }   // never destroyed: no hipFree during static destruction
size_t size_class(size_t b)
{
    size_t c = 4096;
    while (c < b) c <<= 1;
    return b > (64u << 20) ? ((b + (16u << 20) - 1) / (16u << 20)) * (16u << 20) : c;   // powers of two up to 64 MB, then 16 MB steps
}
}  // namespace

int dev_cache_alloc(size_t bytes, void **out)
{
    *out = nullptr;
    if (bytes == 0) return RTBHIP_OK;
    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    const size_t want = size_class(bytes);
    DevCache &c = dcache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto &fb = c.free_blocks[dev];
        auto it = fb.lower_bound(want);
        if (it != fb.end() && it->first <= 2 * want) {          // the smallest idle block that fits, up to twice the request
            *out = it->second;
            c.live[*out] = {dev, it->first};
            c.cached[dev] -= it->first;
            fb.erase(it);
            return RTBHIP_OK;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dev_cache_release();
        e = hipMalloc(&p, want);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (host-path staging)");
    }
    std::lock_guard<std::mutex> lk(c.mu);
    c.live[p] = {dev, want};
    *out = p;
    return RTBHIP_OK;
}

void dev_cache_free(void *p)
{
    if (!p) return;
    DevCache &c = dcache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it == c.live.end()) return;
        const int dev = it->second.first;
        const size_t sz = it->second.second;
        c.live.erase(it);
        if (c.cached[dev] + sz <= c.cap) {
            c.free_blocks[dev].emplace(sz, p);
            c.cached[dev] += sz;
            return;
        }
    }
    (void)hipFree(p);                                           // above the cap: back to the driver now
}

void dev_cache_trim(size_t keep_bytes)
{
    DevCache &c = dcache();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (auto &kv : c.free_blocks) {
            size_t &have = c.cached[kv.first];
            while (have > keep_bytes && !kv.second.empty()) {
                auto it = std::prev(kv.second.end());           // largest first
                have -= it->first;
                drop.push_back(it->second);
                kv.second.erase(it);
            }
        }
    }
    for (void *p : drop) (void)hipFree(p);
}

void dev_cache_release() { dev_cache_trim(0); }

}  // namespace rtbhip
