// shard.cpp -- the multi-GPU half of the boundary (SURVEY 8e / 8b `rtbhip_shard_gather`): the ONE exchange of the path, the gather of the
// ranks' output shards, on RCCL over xGMI.  The path itself has no collective: rows are independent (every configuration / target /
// (q, qd, qdd) triple), chain tables are replicated, each rank computes rtbhip_shard_range(N, rank, world) and leaves its rows in HBM.
//
// librccl.so is opened with dlopen on first use: a single-GPU consumer of librtbhip.so neither links nor loads it (and inside a PyTorch
// process the loader hands back the copy torch already mapped -- same SONAME librccl.so.1 -- so one RCCL serves both).  No torch types: a
// communicator is an opaque pointer (ncclComm_t underneath) made from a 128-byte id the caller ships to its ranks however it likes
// (MPI, a file, torch.distributed's store ...), or by ncclCommInitAll for one process driving every GPU.
//
// Gather geometry: rank r's rows land at byte offset begin_r * row_bytes of `out` (global row order), on `root` alone or -- root = -1 --
// on every rank.  Equal shards (N % world == 0) are ONE ncclGather / ncclAllGather; ragged shards (the first N % world ranks hold one row
// more) are one grouped set of ncclSend / ncclRecv straight into place -- no padding, no staging copy, no second pass.  Gather-to-root
// moves each shard over the sender's single xGMI link to the root (7 peers in parallel); the all-gather form hands every rank all N rows
// (world x the traffic, world x the receive memory): root >= 0 is the default of the Python layer for that reason.
#include "rtbhip_internal.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>

namespace rtbhip {
namespace {

typedef void *ncclComm_t;
struct NcclId { char internal[128]; };
typedef int ncclResult_t;                       // ncclSuccess = 0
enum { kNcclUint8 = 1 };                        // rccl.h: ncclUint8 = 1

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(NcclId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, NcclId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return RTBHIP_OK;
    const char *names[] = {getenv("RTBHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    std::string tried;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(" ") + n;
    }
    if (!h) { set_error("shard: cannot load RCCL (tried" + tried + "): " + (dlerror() ? dlerror() : "?")); return RTBHIP_EHIP; }
    Rccl r;
    r.h = h;
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) ok = false; return p; };
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Gather = (decltype(r.Gather))sym("ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(h); set_error("shard: the RCCL library lacks a required symbol"); return RTBHIP_EHIP; }
    g_rccl = r;
    return RTBHIP_OK;
}

int rccl_fail(ncclResult_t e, const char *what)
{
    set_error(std::string("shard: ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error") + " (" + std::to_string(e) + ")");
    return RTBHIP_EHIP;
}
#define RTB_NCCL(call, what)                                  \
    do {                                                      \
        ncclResult_t _e = (call);                             \
        if (_e != 0) return rccl_fail(_e, what);              \
    } while (0)

int g_force_p2p = 0;      // rtbhip_tune("shard_p2p", 1): take the grouped send / receive form for equal shards too (exercises it where world = 1)

// run on the GPU the shard lives on (one process driving several GPUs), restore the caller's current device on the way out
struct OnDeviceOf {
    int prev = -1;
    explicit OnDeviceOf(const void *buf)
    {
        hipPointerAttribute_t at;
        int cur = 0;
        if (buf && hipPointerGetAttributes(&at, buf) == hipSuccess && hipGetDevice(&cur) == hipSuccess) {
            if (at.type == hipMemoryTypeDevice && at.device != cur && hipSetDevice(at.device) == hipSuccess) prev = cur;
        } else {
            (void)hipGetLastError();
        }
    }
    ~OnDeviceOf() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void range(int64_t N, int rank, int world, int64_t *begin, int64_t *count)
{
    const int64_t base = N / world, extra = N % world;
    *count = base + (rank < extra ? 1 : 0);
    *begin = base * rank + (rank < extra ? rank : extra);
}

}  // namespace
void shard_tune(const char *key, int value) { if (std::string(key) == "shard_p2p") g_force_p2p = value != 0; }
}  // namespace rtbhip

using namespace rtbhip;

extern "C" {

int rtbhip_device_identity(int32_t device, char *pci_bus_id32, unsigned char *uuid16)
{
    int have = 0;
    RTB_HIP(hipGetDeviceCount(&have));
    if (device < 0 || device >= have) { set_error("device_identity: no such device"); return RTBHIP_EINVAL; }
    if (pci_bus_id32) RTB_HIP(hipDeviceGetPCIBusId(pci_bus_id32, 32, device));
    if (uuid16) {
        hipUUID u;
        RTB_HIP(hipDeviceGetUuid(&u, device));
        memcpy(uuid16, u.bytes, 16);
    }
    return RTBHIP_OK;
}

// ---- device memory and streams for consumers that have no HIP binding of their own (a C / Go / Java host above this ABI): enough to keep
// inputs and results resident and to drive one stream per GPU.  A consumer that already holds device pointers (PyTorch, its own HIP code)
// passes those and never calls these.
int rtbhip_device_alloc(int32_t device, uint64_t bytes, void **ptr)
{
    if (!ptr) { set_error("device_alloc: NULL out"); return RTBHIP_EINVAL; }
    *ptr = nullptr;
    int have = 0, cur = 0;
    RTB_HIP(hipGetDeviceCount(&have));
    if (device < 0 || device >= have) { set_error("device_alloc: no such device"); return RTBHIP_EINVAL; }
    RTB_HIP(hipGetDevice(&cur));
    if (cur != device) RTB_HIP(hipSetDevice(device));
    hipError_t e = hipMalloc(ptr, bytes ? (size_t)bytes : 1);
    if (cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return hip_fail(e, "device_alloc: hipMalloc");
    return RTBHIP_OK;
}

int rtbhip_device_free(void *ptr)
{
    if (ptr) RTB_HIP(hipFree(ptr));
    return RTBHIP_OK;
}

int rtbhip_device_copy(void *dst, const void *src, uint64_t bytes, int32_t kind, void *stream)
{
    if (bytes == 0) return RTBHIP_OK;
    if (!dst || !src || kind < 1 || kind > 3) { set_error("device_copy: bad argument (kind: 1 host to device, 2 device to host, 3 device to device)"); return RTBHIP_EINVAL; }
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    if (stream) RTB_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, k, (hipStream_t)stream));
    else RTB_HIP(hipMemcpy(dst, src, (size_t)bytes, k));
    return RTBHIP_OK;
}

int rtbhip_stream_create(int32_t device, void **stream)
{
    if (!stream) { set_error("stream_create: NULL out"); return RTBHIP_EINVAL; }
    int have = 0, cur = 0;
    RTB_HIP(hipGetDeviceCount(&have));
    if (device < 0 || device >= have) { set_error("stream_create: no such device"); return RTBHIP_EINVAL; }
    RTB_HIP(hipGetDevice(&cur));
    if (cur != device) RTB_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return hip_fail(e, "stream_create");
    *stream = s;
    return RTBHIP_OK;
}

int rtbhip_stream_destroy(void *stream)
{
    if (stream) RTB_HIP(hipStreamDestroy((hipStream_t)stream));
    return RTBHIP_OK;
}

int rtbhip_stream_sync(void *stream)
{
    RTB_HIP(hipStreamSynchronize((hipStream_t)stream));      // NULL: the current device's default stream
    return RTBHIP_OK;
}

int rtbhip_shard_comm_id(void *id128)
{
    if (!id128) { set_error("shard_comm_id: NULL id"); return RTBHIP_EINVAL; }
    int rc = rccl_load();
    if (rc != RTBHIP_OK) return rc;
    RTB_NCCL(g_rccl.GetUniqueId((NcclId *)id128), "ncclGetUniqueId");
    return RTBHIP_OK;
}

int rtbhip_shard_comm_create(const void *id128, int32_t world, int32_t rank, rtbhip_comm_t *comm)
{
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) { set_error("shard_comm_create: bad argument"); return RTBHIP_EINVAL; }
    int rc = rccl_load();
    if (rc != RTBHIP_OK) return rc;
    NcclId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    RTB_NCCL(g_rccl.CommInitRank(&c, world, id, rank), "ncclCommInitRank");      // on the calling thread's current device
    *comm = c;
    return RTBHIP_OK;
}

int rtbhip_shard_comm_create_all(int32_t ndev, const int32_t *devices, rtbhip_comm_t *comms)
{
    if (ndev < 1 || !comms) { set_error("shard_comm_create_all: bad argument"); return RTBHIP_EINVAL; }
    int have = 0;
    RTB_HIP(hipGetDeviceCount(&have));
    if (ndev > have) { set_error("shard_comm_create_all: " + std::to_string(ndev) + " communicators asked for, " + std::to_string(have) + " devices visible"); return RTBHIP_EINVAL; }
    if (devices)
        for (int i = 0; i < ndev; ++i)
            for (int j = 0; j < i; ++j)
                if (devices[i] == devices[j]) { set_error("shard_comm_create_all: a device is listed twice"); return RTBHIP_EINVAL; }
    int rc = rccl_load();
    if (rc != RTBHIP_OK) return rc;
    static_assert(sizeof(int32_t) == sizeof(int), "device list");
    RTB_NCCL(g_rccl.CommInitAll((ncclComm_t *)comms, ndev, (const int *)devices), "ncclCommInitAll");
    return RTBHIP_OK;
}

int rtbhip_shard_comm_destroy(rtbhip_comm_t comm)
{
    if (!comm) return RTBHIP_OK;
    int rc = rccl_load();
    if (rc != RTBHIP_OK) return rc;
    RTB_NCCL(g_rccl.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
    return RTBHIP_OK;
}

int rtbhip_shard_comm_info(rtbhip_comm_t comm, int32_t *world, int32_t *rank, int32_t *rccl_version)
{
    int rc = rccl_load();
    if (rc != RTBHIP_OK) return rc;
    int v = 0;
    if (comm && world) { RTB_NCCL(g_rccl.CommCount((ncclComm_t)comm, &v), "ncclCommCount"); *world = v; }
    if (comm && rank) { RTB_NCCL(g_rccl.CommUserRank((ncclComm_t)comm, &v), "ncclCommUserRank"); *rank = v; }
    if (rccl_version) { RTB_NCCL(g_rccl.GetVersion(&v), "ncclGetVersion"); *rccl_version = v; }
    return RTBHIP_OK;
}

int rtbhip_shard_group(int32_t begin)
{
    int rc = rccl_load();
    if (rc != RTBHIP_OK) return rc;
    if (begin) RTB_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    else RTB_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
    return RTBHIP_OK;
}

int rtbhip_shard_gather(rtbhip_comm_t comm, const void *local, int64_t rows, int64_t row_bytes, int64_t N, int32_t world, int32_t rank,
                        int32_t root, void *out, void *stream)
{
    if (N < 0 || row_bytes < 1 || world < 1 || rank < 0 || rank >= world || root < -1 || root >= world) { set_error("shard_gather: bad argument"); return RTBHIP_EINVAL; }
    int64_t begin, count;
    range(N, rank, world, &begin, &count);
    if (rows != count) {
        set_error("shard_gather: rank " + std::to_string(rank) + " of " + std::to_string(world) + " holds " + std::to_string(count) + " of " + std::to_string(N) +
                  " rows (rtbhip_shard_range), not " + std::to_string(rows));
        return RTBHIP_EINVAL;
    }
    const bool receives = root < 0 || root == rank;
    if (rows > 0 && !local) { set_error("shard_gather: NULL local shard"); return RTBHIP_EINVAL; }
    if (receives && N > 0 && !out) { set_error("shard_gather: NULL output on a receiving rank"); return RTBHIP_EINVAL; }
    if (!comm && world > 1) { set_error("shard_gather: a world of more than one rank needs a communicator (rtbhip_shard_comm_create)"); return RTBHIP_EINVAL; }
    if (N == 0) return RTBHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    char *dst = (char *)out;
    OnDeviceOf scope(local ? local : out);
    if (!comm) {                                    // one rank, no communicator: the shard IS the result
        if (out != local) RTB_HIP(hipMemcpyAsync(out, local, (size_t)(N * row_bytes), hipMemcpyDeviceToDevice, s));
        return RTBHIP_OK;
    }
    int rc = rccl_load();
    if (rc != RTBHIP_OK) return rc;
    int cw = 0, cr = 0;
    RTB_NCCL(g_rccl.CommCount((ncclComm_t)comm, &cw), "ncclCommCount");
    RTB_NCCL(g_rccl.CommUserRank((ncclComm_t)comm, &cr), "ncclCommUserRank");
    if (cw != world || cr != rank) {
        set_error("shard_gather: the communicator is rank " + std::to_string(cr) + " of " + std::to_string(cw) + ", the call says rank " + std::to_string(rank) + " of " + std::to_string(world));
        return RTBHIP_EINVAL;
    }
    ncclComm_t c = (ncclComm_t)comm;
    if (N % world == 0 && !g_force_p2p) {           // equal shards: ONE collective, rank r's rows at offset r * rows
        const size_t bytes = (size_t)(rows * row_bytes);
        if (root < 0) RTB_NCCL(g_rccl.AllGather(local, out, bytes, kNcclUint8, c, s), "ncclAllGather");
        else RTB_NCCL(g_rccl.Gather(local, out, bytes, kNcclUint8, root, c, s), "ncclGather");
        return RTBHIP_OK;
    }
    // ragged shards: one group of point-to-point transfers straight into place
    RTB_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    ncclResult_t e = 0;
    for (int peer = 0; peer < world && e == 0; ++peer) {
        if ((root < 0 || root == peer) && rows > 0) e = g_rccl.Send(local, (size_t)(rows * row_bytes), kNcclUint8, peer, c, s);      // (self-sends pair with the self-receive below)
        if (receives && e == 0) {
            int64_t pb, pc;
            range(N, peer, world, &pb, &pc);
            if (pc > 0) e = g_rccl.Recv(dst + pb * row_bytes, (size_t)(pc * row_bytes), kNcclUint8, peer, c, s);
        }
    }
    ncclResult_t e2 = g_rccl.GroupEnd();
    if (e != 0) return rccl_fail(e, "ncclSend / ncclRecv");
    if (e2 != 0) return rccl_fail(e2, "ncclGroupEnd");
    return RTBHIP_OK;
}

}  // extern "C"
