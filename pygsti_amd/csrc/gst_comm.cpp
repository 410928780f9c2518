// gst_comm.cpp -- the multi-GPU exchange of include/gstfwd.h (gst_comm_*): row blocks of probability / Jacobian arrays
// and the normal-equation sums travel between DEVICE buffers of one-process-per-GPU ranks.
//
// What the reference does with mpi4py on host arrays (pygsti/baseobjs/resourceallocation.py:316-348 Gatherv / Allgatherv,
// :441-508 Allreduce; called from layouts/distlayout.py:1143-1147, 1259, 1355) is done here over xGMI:
//   * RCCL transport: librccl is bound at run time (dlopen + dlsym: a single-GPU process never loads its 300 MB), row
//     blocks move as one grouped ncclSend/ncclRecv exchange (each GPU pair has its own xGMI link; a ring collective is
//     bound by one link, a fan-in / all-to-all of point-to-point transfers uses all seven), sums are ncclAllReduce;
//   * IPC transport: peer writes through HIP IPC handles exchanged in a POSIX shared-memory mailbox -- no compute units,
//     works for ranks that share a GPU (development boxes, tests), deterministic rank-order sums.
// There is no host staging in either transport.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the functions are looked up with dlsym

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gstfwd.h"
#include "gst_internal.hpp"

namespace {

using gst::set_error;

#define HIP_TRYC(expr)                                                                               \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return set_error(e_ == hipErrorOutOfMemory ? GST_ENOMEM : GST_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---- RCCL, bound at run time --------------------------------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    std::string path;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
};

std::mutex g_rccl_mutex;
RcclApi g_rccl;
std::string g_rccl_error;

int load_rccl(const RcclApi** out)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.handle) {
        if (!g_rccl_error.empty()) return set_error(GST_EUNSUPPORTED, g_rccl_error);
        std::vector<std::string> names;
        // ROCm's own RCCL first: libgstfwd links ROCm's HIP runtime, and an RCCL built against another runtime release
        // (e.g. the copy a Python framework bundles, which a bare SONAME lookup would return once that framework is
        // imported) is not a combination anyone tests
        if (const char* e = std::getenv("GST_RCCL_LIBRARY")) names.push_back(e);
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        std::string tried;
        void* h = nullptr;
        for (const auto& n : names) {
            h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) { g_rccl.path = n; break; }
            tried += "\n  " + n + ": " + (dlerror() ? "not loadable" : "?");
        }
        if (!h) {
            g_rccl_error = "RCCL is not available (tried:" + tried + ")";
            return set_error(GST_EUNSUPPORTED, g_rccl_error);
        }
        RcclApi a;
        a.handle = h; a.path = g_rccl.path;
        bool ok = true;
#define BIND(field, sym) do { a.field = (decltype(a.field))dlsym(h, #sym); ok = ok && a.field != nullptr; } while (0)
        BIND(GetVersion, ncclGetVersion); BIND(GetUniqueId, ncclGetUniqueId); BIND(CommInitRank, ncclCommInitRank);
        BIND(CommDestroy, ncclCommDestroy); BIND(GetErrorString, ncclGetErrorString); BIND(GroupStart, ncclGroupStart);
        BIND(GroupEnd, ncclGroupEnd); BIND(Send, ncclSend); BIND(Recv, ncclRecv); BIND(AllReduce, ncclAllReduce);
#undef BIND
        if (!ok) {
            dlclose(h);
            g_rccl_error = "the RCCL library found (" + a.path + ") lacks a required symbol";
            return set_error(GST_EUNSUPPORTED, g_rccl_error);
        }
        g_rccl = a;
    }
    *out = &g_rccl;
    return GST_OK;
}

#define NCCL_TRY(api, expr)                                                                          \
    do {                                                                                             \
        ncclResult_t r_ = (expr);                                                                    \
        if (r_ != ncclSuccess) return set_error(GST_EHIP, std::string(#expr) + ": " + (api)->GetErrorString(r_)); \
    } while (0)

// ---- IPC transport: shared-memory mailbox -------------------------------------------------------------------------
constexpr int IPC_MAX_RANKS = 64;
struct IpcSlot {
    hipIpcMemHandle_t handle;      // allocation that holds this rank's destination buffer
    uint64_t offset;               // of the buffer inside that allocation
    uint64_t bytes;                // size of the allocation
    uint64_t alloc_id;             // changes whenever (handle, bytes) do: peers re-open on a new id
    int32_t device;
    int32_t pad;
};
struct IpcMailbox {
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> barrier_count;
    std::atomic<uint32_t> barrier_sense;
    uint32_t size;
    IpcSlot slot[IPC_MAX_RANKS];
};

struct OpenedHandle { uint64_t alloc_id = 0; void* base = nullptr; uint64_t last_use = 0; bool pinned = false; };   // pinned: handed out by gst_comm_map_root_buffer, never evicted
struct RootDesc { hipIpcMemHandle_t handle; uint64_t offset, bytes; };        // what a root sends its peers under RCCL (80 bytes)
constexpr int IPC_MAPPINGS_PER_PEER = 8;       // a peer's destinations alternate (Jacobian, probabilities, staging ...)

}  // namespace

struct gst_comm {
    int transport = GST_TRANSPORT_RCCL, rank = 0, size = 1, device = 0;
    hipStream_t stream = nullptr;
    // RCCL
    const RcclApi* api = nullptr;
    ncclComm_t nccl = nullptr;
    int rccl_version = 0;
    // IPC
    IpcMailbox* box = nullptr;
    std::string shm_name;
    uint32_t local_sense = 0;
    uint64_t next_alloc_id = 1;
    // publisher side: allocations this rank has published before, so that a destination that comes back (Jacobian /
    // probabilities / all-reduce staging alternate) is re-published under its OLD id and the peers' mapping caches hit
    struct Published { hipIpcMemHandle_t handle; uint64_t bytes; uint64_t alloc_id; uint64_t last_use; };
    std::vector<Published> published;
    uint64_t ipc_opens = 0;                               // hipIpcOpenMemHandle calls made by this rank (tests: no re-open on A/B/A/B)
    std::vector<OpenedHandle> opened;                     // per peer rank: IPC_MAPPINGS_PER_PEER mappings of its recent destinations
    uint64_t use_clock = 0;
    double* stage = nullptr;                              // all-reduce staging [size][stage_n]
    size_t stage_n = 0;
    // gst_comm_map_root_buffer under RCCL: descriptor staging (page-locked host + device), mappings opened so far
    RootDesc* h_desc = nullptr;
    char* d_desc = nullptr;
    struct RootMap { hipIpcMemHandle_t handle; void* base; };
    bool test_self_send = false;                          // GST_TEST_FORCE comm_self=1 (tests): blocks that stay on their rank travel
                                                          // through ncclSend / ncclRecv to the rank itself instead of a local copy
    std::vector<RootMap> root_maps;
};

namespace {

int ipc_barrier(gst_comm* c, long timeout_s = 600)
{
    IpcMailbox* b = c->box;
    c->local_sense ^= 1u;
    if (b->barrier_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->size) {
        b->barrier_count.store(0, std::memory_order_relaxed);
        b->barrier_sense.store(c->local_sense, std::memory_order_release);
    } else {
        struct timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        uint64_t spins = 0;
        while (b->barrier_sense.load(std::memory_order_acquire) != c->local_sense) {
            if (++spins > 2000) sched_yield();
            if ((spins & 0xffff) == 0) {
                struct timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (t1.tv_sec - t0.tv_sec > timeout_s) return set_error(GST_EHIP, "IPC barrier timed out (a peer rank is gone?)");
            }
        }
    }
    return GST_OK;
}

// Publish the allocation that contains `ptr` as this rank's destination buffer.
int ipc_publish(gst_comm* c, const void* ptr)
{
    IpcSlot& s = c->box->slot[c->rank];
    void* base = nullptr;
    size_t bytes = 0;
    HIP_TRYC(hipMemGetAddressRange((hipDeviceptr_t*)&base, &bytes, (hipDeviceptr_t)ptr));
    // (a freed and re-allocated buffer may come back at the same address: the handle, not the address, identifies it)
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));        // (the runtime fills only part of the 64 bytes; the handles are compared bytewise below)
    HIP_TRYC(hipIpcGetMemHandle(&h, base));
    if (s.alloc_id == 0 || s.bytes != bytes || std::memcmp(&h, &s.handle, sizeof(h)) != 0) {
        gst_comm::Published* hit = nullptr;
        gst_comm::Published* lru = nullptr;
        for (auto& p : c->published) {
            if (p.bytes == bytes && std::memcmp(&h, &p.handle, sizeof(h)) == 0) hit = &p;
            if (!lru || p.last_use < lru->last_use) lru = &p;
        }
        if (!hit) {
            // (as many remembered allocations as a peer keeps mappings: an id evicted here is evicted there, too)
            if (c->published.size() < (size_t)IPC_MAPPINGS_PER_PEER) { c->published.push_back({}); hit = &c->published.back(); }
            else hit = lru;
            hit->handle = h; hit->bytes = bytes;
            hit->alloc_id = ((uint64_t)(c->rank + 1) << 40) | c->next_alloc_id++;
        }
        hit->last_use = ++c->use_clock;
        s.handle = h; s.bytes = bytes; s.device = c->device;
        s.alloc_id = hit->alloc_id;
    }
    s.offset = (uint64_t)((const char*)ptr - (const char*)base);
    return GST_OK;
}

// Device address, in THIS process, of peer `r`'s published destination buffer.
int ipc_peer_ptr(gst_comm* c, int r, char** out, bool pin = false)
{
    const IpcSlot& s = c->box->slot[r];
    OpenedHandle* set = &c->opened[(size_t)r * IPC_MAPPINGS_PER_PEER];
    OpenedHandle* hit = nullptr;
    OpenedHandle* lru = nullptr;               // victim: an empty entry if there is one, else the least recently used
    for (int k = 0; k < IPC_MAPPINGS_PER_PEER; k++) {
        if (set[k].base && set[k].alloc_id == s.alloc_id) hit = &set[k];
        if (set[k].pinned) continue;
        if (!lru) lru = &set[k];
        else if (lru->base && (!set[k].base || set[k].last_use < lru->last_use)) lru = &set[k];
    }
    if (!hit) {
        if (!lru) return set_error(GST_EUNSUPPORTED, "every IPC mapping of this peer is pinned by gst_comm_map_root_buffer");
        if (lru->base) { (void)hipIpcCloseMemHandle(lru->base); lru->base = nullptr; lru->alloc_id = 0; }
        void* p = nullptr;
        HIP_TRYC(hipIpcOpenMemHandle(&p, s.handle, hipIpcMemLazyEnablePeerAccess));
        c->ipc_opens++;
        lru->base = p; lru->alloc_id = s.alloc_id;
        hit = lru;
    }
    hit->last_use = ++c->use_clock;
    OpenedHandle& o = *hit;
    if (pin) o.pinned = true;
    *out = (char*)o.base + s.offset;
    return GST_OK;
}

struct Blocks {
    int32_t n;
    const int32_t* owner;
    const int64_t* row0;
    const int64_t* rows;
};

// the row blocks a collective writes into d_full no longer hold an exact Jacobian's zeros (gst_track.cpp); an in-place
// all-gather leaves the caller's own blocks alone
void touch_rows(double* d_full, const Blocks& b, int64_t row_doubles, int skip_owner)
{
    if (!d_full) return;
    for (int32_t k = 0; k < b.n; k++)
        if (b.owner[k] != skip_owner && b.rows[k] > 0) gst::track_touch(d_full + b.row0[k] * row_doubles, (size_t)(b.rows[k] * row_doubles) * 8);
}

int check_blocks(const gst_comm* c, const Blocks& b, int64_t row_doubles)
{
    if (b.n < 0 || row_doubles < 0 || (b.n > 0 && (!b.owner || !b.row0 || !b.rows))) return set_error(GST_EINVAL, "bad block list");
    for (int32_t k = 0; k < b.n; k++)
        if (b.owner[k] < 0 || b.owner[k] >= c->size || b.row0[k] < 0 || b.rows[k] < 0)
            return set_error(GST_EINVAL, "block " + std::to_string(k) + ": owner or row range out of range");
    return GST_OK;
}

// The one exchange both gathers are: every block travels from its owner to `root` (root < 0: to every other rank).
// Senders read block k at src_of(k), receivers write it at d_full + row0[k] * row_doubles.
int exchange_rows(gst_comm* c, hipStream_t st, const double* d_local, double* d_full, int64_t row_doubles, const Blocks& b, int root)
{
    const bool all = root < 0;
    const bool i_receive = all || c->rank == root;
    if (i_receive && !d_full && b.n > 0) return set_error(GST_EINVAL, "d_full is NULL on a receiving rank");
    // where this rank's own block k starts: in place inside d_full, or packed in d_local (gather, non-root ranks)
    std::vector<const double*> src((size_t)b.n, nullptr);
    {
        int64_t packed = 0;
        for (int32_t k = 0; k < b.n; k++) {
            if (b.owner[k] != c->rank) continue;
            if (i_receive) src[(size_t)k] = d_full + b.row0[k] * row_doubles;
            else {
                if (!d_local && b.rows[k] > 0) return set_error(GST_EINVAL, "d_local is NULL on a sending rank");
                src[(size_t)k] = d_local + packed;
                packed += b.rows[k] * row_doubles;
            }
        }
    }
    if (c->size == 1 && c->transport != GST_TRANSPORT_RCCL) return GST_OK;
    HIP_TRYC(hipSetDevice(c->device));
    if (c->transport == GST_TRANSPORT_RCCL) {          // (one rank: an empty group -- still a round trip through RCCL)
        const RcclApi* A = c->api;
        NCCL_TRY(A, A->GroupStart());
        // a failing Send / Recv must not leave the group open (every later RCCL call of this thread would join it):
        // remember the first error, close the group, then report
        ncclResult_t bad = ncclSuccess;
        const char* what = "";
        for (int32_t k = 0; k < b.n && bad == ncclSuccess; k++) {
            const size_t cnt = (size_t)(b.rows[k] * row_doubles);
            if (cnt == 0) continue;
            const int o = b.owner[k];
            if (o == c->rank) {
                for (int r = 0; r < c->size && bad == ncclSuccess; r++) {
                    if (r == c->rank || !(all || r == root)) continue;
                    bad = A->Send(src[(size_t)k], cnt, ncclDouble, r, c->nccl, st); what = "ncclSend";
                }
            } else if (i_receive) {
                bad = A->Recv(d_full + b.row0[k] * row_doubles, cnt, ncclDouble, o, c->nccl, st); what = "ncclRecv";
            }
        }
        const ncclResult_t ended = A->GroupEnd();
        if (bad != ncclSuccess) return set_error(GST_EHIP, std::string(what) + ": " + A->GetErrorString(bad));
        if (ended != ncclSuccess) return set_error(GST_EHIP, std::string("ncclGroupEnd: ") + A->GetErrorString(ended));
        return GST_OK;
    }
    // IPC: receivers publish their buffers; once every rank's fills are done (stream sync + barrier) each owner copies its
    // blocks straight into every receiver's array; a second barrier tells the receivers that the data has landed.
    HIP_TRYC(hipStreamSynchronize(st));
    int rc;
    if (i_receive && d_full && (rc = ipc_publish(c, d_full))) return rc;
    if ((rc = ipc_barrier(c))) return rc;
    for (int r = 0; r < c->size; r++) {
        if (r == c->rank || !(all || r == root)) continue;
        char* peer = nullptr;
        bool have = false;
        for (int32_t k = 0; k < b.n; k++) {
            if (b.owner[k] != c->rank || b.rows[k] == 0 || row_doubles == 0) continue;
            if (!have) { if ((rc = ipc_peer_ptr(c, r, &peer))) return rc; have = true; }
            HIP_TRYC(hipMemcpyAsync(peer + (size_t)(b.row0[k] * row_doubles) * 8, src[(size_t)k], (size_t)(b.rows[k] * row_doubles) * 8,
                                    hipMemcpyDeviceToDevice, st));
        }
    }
    HIP_TRYC(hipStreamSynchronize(st));
    return ipc_barrier(c);
}

// General block exchange (Alltoallv on device pointers): block k = cnt[k] doubles from rank src_rank[k]'s d_src + src_off[k]
// to rank dst_rank[k]'s d_dst + dst_off[k].  Every rank passes the same list and its own two base pointers.
int exchange_blocks(gst_comm* c, hipStream_t st, const double* d_src, double* d_dst, int32_t n, const int32_t* src_rank,
                    const int32_t* dst_rank, const int64_t* src_off, const int64_t* dst_off, const int64_t* cnt)
{
    bool i_send = false, i_recv = false;
    for (int32_t k = 0; k < n; k++) {
        if (src_rank[k] < 0 || src_rank[k] >= c->size || dst_rank[k] < 0 || dst_rank[k] >= c->size || src_off[k] < 0 || dst_off[k] < 0 || cnt[k] < 0)
            return set_error(GST_EINVAL, "block " + std::to_string(k) + ": rank, offset or count out of range");
        if (cnt[k] == 0) continue;
        if (src_rank[k] == c->rank) i_send = true;
        if (dst_rank[k] == c->rank) i_recv = true;
    }
    if (i_send && !d_src) return set_error(GST_EINVAL, "d_src is NULL on a sending rank");
    if (i_recv && !d_dst) return set_error(GST_EINVAL, "d_dst is NULL on a receiving rank");
    HIP_TRYC(hipSetDevice(c->device));
    // blocks that stay on their rank: a device-to-device copy in stream order
    const bool self_send = c->test_self_send && c->transport == GST_TRANSPORT_RCCL;
    for (int32_t k = 0; k < n && !self_send; k++)
        if (cnt[k] > 0 && src_rank[k] == c->rank && dst_rank[k] == c->rank)
            HIP_TRYC(hipMemcpyAsync(d_dst + dst_off[k], d_src + src_off[k], (size_t)cnt[k] * 8, hipMemcpyDeviceToDevice, st));
    if (c->size == 1 && c->transport != GST_TRANSPORT_RCCL) return GST_OK;
    if (c->transport == GST_TRANSPORT_RCCL) {
        const RcclApi* A = c->api;
        NCCL_TRY(A, A->GroupStart());
        ncclResult_t bad = ncclSuccess;
        const char* what = "";
        for (int32_t k = 0; k < n && bad == ncclSuccess; k++) {
            if (cnt[k] == 0 || (src_rank[k] == dst_rank[k] && !self_send)) continue;
            if (src_rank[k] == c->rank) { bad = A->Send(d_src + src_off[k], (size_t)cnt[k], ncclDouble, dst_rank[k], c->nccl, st); what = "ncclSend"; }
            if (bad == ncclSuccess && dst_rank[k] == c->rank) { bad = A->Recv(d_dst + dst_off[k], (size_t)cnt[k], ncclDouble, src_rank[k], c->nccl, st); what = "ncclRecv"; }
        }
        const ncclResult_t ended = A->GroupEnd();
        if (bad != ncclSuccess) return set_error(GST_EHIP, std::string(what) + ": " + A->GetErrorString(bad));
        if (ended != ncclSuccess) return set_error(GST_EHIP, std::string("ncclGroupEnd: ") + A->GetErrorString(ended));
        return GST_OK;
    }
    // IPC: receivers publish d_dst; after the barrier every sender copies its blocks straight into the receivers' arrays
    HIP_TRYC(hipStreamSynchronize(st));
    int rc;
    if (i_recv && (rc = ipc_publish(c, d_dst))) return rc;
    if ((rc = ipc_barrier(c))) return rc;
    for (int r = 0; r < c->size; r++) {
        if (r == c->rank) continue;
        char* peer = nullptr;
        bool have = false;
        for (int32_t k = 0; k < n; k++) {
            if (src_rank[k] != c->rank || dst_rank[k] != r || cnt[k] == 0) continue;
            if (!have) { if ((rc = ipc_peer_ptr(c, r, &peer))) return rc; have = true; }
            HIP_TRYC(hipMemcpyAsync(peer + (size_t)dst_off[k] * 8, d_src + src_off[k], (size_t)cnt[k] * 8, hipMemcpyDeviceToDevice, st));
        }
    }
    HIP_TRYC(hipStreamSynchronize(st));
    return ipc_barrier(c);
}

hipStream_t pick_stream(gst_comm* c, gst_plan* plan, int* rc)
{
    *rc = GST_OK;
    if (!plan) return c->stream;
    if ((*rc = gst::plan_ensure_device(plan))) return nullptr;
    if (gst::plan_device(plan) != c->device) {
        *rc = set_error(GST_EINVAL, "the plan lives on device " + std::to_string(gst::plan_device(plan)) + ", the communicator on " + std::to_string(c->device));
        return nullptr;
    }
    return gst::plan_stream(plan);
}

template <typename F>
int guarded(F&& body)
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return set_error(GST_ENOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return set_error(GST_EINVAL, std::string("internal error: ") + e.what());
    } catch (...) {
        return set_error(GST_EINVAL, "internal error (unknown exception)");
    }
}

}  // namespace

extern "C" {

int gst_comm_get_unique_id(int transport, void* id_out)
{
    return guarded([&]() -> int {
        if (!id_out) return set_error(GST_EINVAL, "id_out is NULL");
        std::memset(id_out, 0, GST_COMM_ID_BYTES);
        if (transport == GST_TRANSPORT_RCCL) {
            const RcclApi* A = nullptr;
            int rc = load_rccl(&A);
            if (rc) return rc;
            static_assert(sizeof(ncclUniqueId) <= GST_COMM_ID_BYTES, "ncclUniqueId must fit the id buffer");
            ncclUniqueId id;
            NCCL_TRY(A, A->GetUniqueId(&id));
            std::memcpy(id_out, &id, sizeof(id));
            return GST_OK;
        }
        if (transport == GST_TRANSPORT_IPC) {
            // a random name for the shared-memory mailbox
            unsigned char rnd[16];
            bool ok = false;
            if (FILE* f = std::fopen("/dev/urandom", "rb")) { ok = std::fread(rnd, 1, sizeof(rnd), f) == sizeof(rnd); std::fclose(f); }
            if (!ok) {
                struct timespec t;
                clock_gettime(CLOCK_REALTIME, &t);
                uint64_t x = (uint64_t)t.tv_nsec * 6364136223846793005ull + (uint64_t)getpid() * 1442695040888963407ull + (uint64_t)t.tv_sec;
                for (int i = 0; i < 16; i++) { x = x * 6364136223846793005ull + 1442695040888963407ull; rnd[i] = (unsigned char)(x >> 56); }
            }
            char* s = (char*)id_out;
            int n = std::snprintf(s, GST_COMM_ID_BYTES, "/gstfwd_");
            for (int i = 0; i < 16; i++) n += std::snprintf(s + n, GST_COMM_ID_BYTES - n, "%02x", rnd[i]);
            return GST_OK;
        }
        return set_error(GST_EINVAL, "unknown transport");
    });
}

int gst_comm_create(int transport, int device, int rank, int size, const void* id, gst_comm** out)
{
    return guarded([&]() -> int {
        if (!out) return set_error(GST_EINVAL, "out is NULL");
        *out = nullptr;
        if (size < 1 || rank < 0 || rank >= size || !id) return set_error(GST_EINVAL, "bad rank / size / id");
        if (transport != GST_TRANSPORT_RCCL && transport != GST_TRANSPORT_IPC) return set_error(GST_EINVAL, "unknown transport");
        int n_dev = 0;
        hipError_t e = hipGetDeviceCount(&n_dev);
        if (e != hipSuccess || n_dev <= 0)
            return set_error(GST_ENODEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + ")");
        if (device < 0) { HIP_TRYC(hipGetDevice(&device)); }
        if (device >= n_dev) return set_error(GST_ENODEVICE, "device ordinal out of range");
        HIP_TRYC(hipSetDevice(device));
        gst_comm* c = new gst_comm();
        c->transport = transport; c->rank = rank; c->size = size; c->device = device;
        if (const char* tf = std::getenv("GST_TEST_FORCE")) c->test_self_send = std::strstr(tf, "comm_self=1") != nullptr;
        hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (es != hipSuccess) { delete c; return set_error(GST_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(es)); }
        if (transport == GST_TRANSPORT_RCCL) {
            int rc = load_rccl(&c->api);
            if (rc) { (void)hipStreamDestroy(c->stream); delete c; return rc; }
            (void)c->api->GetVersion(&c->rccl_version);
            ncclUniqueId uid;
            std::memcpy(&uid, id, sizeof(uid));
            ncclResult_t r = c->api->CommInitRank(&c->nccl, size, uid, rank);
            if (r != ncclSuccess) {
                std::string msg = std::string("ncclCommInitRank: ") + c->api->GetErrorString(r) +
                                  " (ranks that share a GPU cannot form an RCCL communicator: use GST_TRANSPORT_IPC)";
                (void)hipStreamDestroy(c->stream); delete c;
                return set_error(GST_EHIP, msg);
            }
        } else {
            if (size > IPC_MAX_RANKS) { (void)hipStreamDestroy(c->stream); delete c; return set_error(GST_EUNSUPPORTED, "the IPC transport serves at most 64 ranks"); }
            char name[GST_COMM_ID_BYTES];
            std::memcpy(name, id, GST_COMM_ID_BYTES);
            name[GST_COMM_ID_BYTES - 1] = 0;
            if (name[0] != '/' || std::strlen(name) < 8) { (void)hipStreamDestroy(c->stream); delete c; return set_error(GST_EINVAL, "not an IPC transport id"); }
            c->shm_name = name;
            int fd = -1;
            if (rank == 0) {
                fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
                if (fd >= 0 && ftruncate(fd, sizeof(IpcMailbox)) != 0) { close(fd); shm_unlink(name); fd = -1; }
            } else {
                for (int tries = 0; tries < 60000 && fd < 0; tries++) {      // rank 0 may not be there yet: up to 60 s
                    fd = shm_open(name, O_RDWR, 0600);
                    if (fd >= 0) {
                        struct stat sb;
                        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(IpcMailbox)) { close(fd); fd = -1; }
                    }
                    if (fd < 0) usleep(1000);
                }
            }
            if (fd < 0) { (void)hipStreamDestroy(c->stream); delete c; return set_error(GST_EHIP, std::string("cannot open the shared-memory mailbox ") + name); }
            void* m = mmap(nullptr, sizeof(IpcMailbox), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) { (void)hipStreamDestroy(c->stream); delete c; return set_error(GST_EHIP, "mmap of the shared-memory mailbox failed"); }
            c->box = (IpcMailbox*)m;          // (a fresh segment is zero-filled: counters start at 0)
            if (rank == 0) c->box->size = (uint32_t)size;
            c->opened.assign((size_t)size * IPC_MAPPINGS_PER_PEER, OpenedHandle());
            struct timespec t0;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            // every rank must have been started with the same world size; rank 0 wrote its own into the mailbox before it
            // attached.  Checked BEFORE the attach wait: a mismatch would otherwise end in that wait's time-out (too few
            // ranks) or in barriers that release early (too many)
            for (;;) {
                const uint32_t theirs = ((volatile IpcMailbox*)c->box)->size;
                if (theirs == (uint32_t)size) break;
                struct timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (theirs != 0 || t1.tv_sec - t0.tv_sec > 120) {
                    munmap(c->box, sizeof(IpcMailbox));
                    (void)hipStreamDestroy(c->stream); delete c;
                    return set_error(GST_EINVAL, "rank " + std::to_string(rank) + " was created with size " + std::to_string(size) +
                                                     " but rank 0's mailbox says " + std::to_string(theirs));
                }
                usleep(200);
            }
            c->box->attached.fetch_add(1, std::memory_order_acq_rel);
            while (c->box->attached.load(std::memory_order_acquire) < (uint32_t)size) {
                usleep(200);
                struct timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (t1.tv_sec - t0.tv_sec > 120) {
                    munmap(c->box, sizeof(IpcMailbox));
                    if (rank == 0) shm_unlink(name);
                    (void)hipStreamDestroy(c->stream); delete c;
                    return set_error(GST_EHIP, "timed out waiting for the other ranks to attach to the mailbox");
                }
            }
            int rc = ipc_barrier(c);
            if (rank == 0) shm_unlink(name);       // everyone is attached: the name can go (no stale segments after a crash)
            if (rc) { munmap(c->box, sizeof(IpcMailbox)); (void)hipStreamDestroy(c->stream); delete c; return rc; }
        }
        *out = c;
        return GST_OK;
    });
}

int gst_comm_destroy(gst_comm* c)
{
    return guarded([&]() -> int {
        if (!c) return GST_OK;
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->transport == GST_TRANSPORT_RCCL) {
            if (c->nccl) (void)c->api->CommDestroy(c->nccl);
        } else if (c->box) {
            (void)ipc_barrier(c, 15);               // nobody unmaps a buffer a peer may still be writing (a vanished peer: move on)
            for (auto& o : c->opened) if (o.base) (void)hipIpcCloseMemHandle(o.base);
            munmap(c->box, sizeof(IpcMailbox));
        }
        if (c->stage) (void)hipFree(c->stage);
        for (auto& m : c->root_maps) if (m.base) (void)hipIpcCloseMemHandle(m.base);
        if (c->h_desc) (void)hipHostFree(c->h_desc);
        if (c->d_desc) (void)hipFree(c->d_desc);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return GST_OK;
    });
}

int gst_comm_allgather_rows(gst_comm* c, gst_plan* plan, double* d_full, int64_t row_doubles, int32_t n_blocks,
                            const int32_t* blk_owner, const int64_t* blk_row0, const int64_t* blk_rows)
{
    return guarded([&]() -> int {
        if (!c) return set_error(GST_EINVAL, "comm is NULL");
        Blocks b{n_blocks, blk_owner, blk_row0, blk_rows};
        int rc = check_blocks(c, b, row_doubles);
        if (rc) return rc;
        touch_rows(d_full, b, row_doubles, c->rank);
        hipStream_t st = pick_stream(c, plan, &rc);
        if (rc) return rc;
        return exchange_rows(c, st, nullptr, d_full, row_doubles, b, -1);
    });
}

int gst_comm_gather_rows(gst_comm* c, gst_plan* plan, const double* d_local, double* d_full, int64_t row_doubles,
                         int32_t n_blocks, const int32_t* blk_owner, const int64_t* blk_row0, const int64_t* blk_rows, int32_t root)
{
    return guarded([&]() -> int {
        if (!c) return set_error(GST_EINVAL, "comm is NULL");
        if (root < 0 || root >= c->size) return set_error(GST_EINVAL, "root out of range");
        Blocks b{n_blocks, blk_owner, blk_row0, blk_rows};
        int rc = check_blocks(c, b, row_doubles);
        if (rc) return rc;
        if (c->rank == root) touch_rows(d_full, b, row_doubles, -1);
        hipStream_t st = pick_stream(c, plan, &rc);
        if (rc) return rc;
        return exchange_rows(c, st, d_local, d_full, row_doubles, b, root);
    });
}

// The fan-in WITHOUT a copy: every rank gets a device pointer, valid in its own process, onto `root`'s buffer, so that its fill
// writes its row block straight into the assembled array -- over xGMI while the kernel runs.  The path's "gather to rank 0"
// then costs no separate pass over the data (resourceallocation.py:329-348 is a Gatherv of host arrays after the fill).
int gst_comm_map_root_buffer(gst_comm* c, int32_t root, void* d_buf, void** d_mapped)
{
    return guarded([&]() -> int {
        if (!c || !d_mapped) return set_error(GST_EINVAL, "NULL argument");
        *d_mapped = nullptr;
        if (root < 0 || root >= c->size) return set_error(GST_EINVAL, "root out of range");
        if (c->rank == root && !d_buf) return set_error(GST_EINVAL, "d_buf is NULL on the root");
        HIP_TRYC(hipSetDevice(c->device));
        if (c->size == 1) { *d_mapped = d_buf; return GST_OK; }
        if (c->transport == GST_TRANSPORT_IPC) {
            int rc;
            if (c->rank == root && (rc = ipc_publish(c, d_buf))) { (void)ipc_barrier(c); (void)ipc_barrier(c); return rc; }
            if ((rc = ipc_barrier(c))) return rc;
            int rc_open = GST_OK;
            if (c->rank == root) *d_mapped = d_buf;
            else { char* ptr = nullptr; rc_open = ipc_peer_ptr(c, root, &ptr, true); *d_mapped = ptr; }
            if ((rc = ipc_barrier(c))) return rc;          // (the root's slot may be re-published from here on)
            return rc_open;
        }
        // RCCL: the root's (IPC handle, offset, size) travels as 80 bytes through the communicator itself
        if (!c->h_desc) HIP_TRYC(hipHostMalloc((void**)&c->h_desc, sizeof(RootDesc), hipHostMallocDefault));
        if (!c->d_desc) HIP_TRYC(hipMalloc((void**)&c->d_desc, sizeof(RootDesc)));
        const RcclApi* A = c->api;
        if (c->rank == root) {
            void* base = nullptr; size_t bytes = 0;
            HIP_TRYC(hipMemGetAddressRange((hipDeviceptr_t*)&base, &bytes, (hipDeviceptr_t)d_buf));
            std::memset(c->h_desc, 0, sizeof(RootDesc));
            HIP_TRYC(hipIpcGetMemHandle(&c->h_desc->handle, base));
            c->h_desc->offset = (uint64_t)((char*)d_buf - (char*)base); c->h_desc->bytes = bytes;
            HIP_TRYC(hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RootDesc), hipMemcpyHostToDevice, c->stream));
        }
        NCCL_TRY(A, A->GroupStart());
        ncclResult_t bad = ncclSuccess;
        if (c->rank == root) { for (int r = 0; r < c->size && bad == ncclSuccess; r++) if (r != root) bad = A->Send(c->d_desc, sizeof(RootDesc), ncclChar, r, c->nccl, c->stream); }
        else bad = A->Recv(c->d_desc, sizeof(RootDesc), ncclChar, root, c->nccl, c->stream);
        const ncclResult_t ended = A->GroupEnd();
        if (bad != ncclSuccess) return set_error(GST_EHIP, std::string("gst_comm_map_root_buffer: ") + A->GetErrorString(bad));
        if (ended != ncclSuccess) return set_error(GST_EHIP, std::string("ncclGroupEnd: ") + A->GetErrorString(ended));
        if (c->rank == root) { HIP_TRYC(hipStreamSynchronize(c->stream)); *d_mapped = d_buf; return GST_OK; }
        HIP_TRYC(hipMemcpyAsync(c->h_desc, c->d_desc, sizeof(RootDesc), hipMemcpyDeviceToHost, c->stream));
        HIP_TRYC(hipStreamSynchronize(c->stream));
        for (auto& m : c->root_maps)
            if (std::memcmp(&m.handle, &c->h_desc->handle, sizeof(hipIpcMemHandle_t)) == 0) { *d_mapped = (char*)m.base + c->h_desc->offset; return GST_OK; }
        void* p = nullptr;
        HIP_TRYC(hipIpcOpenMemHandle(&p, c->h_desc->handle, hipIpcMemLazyEnablePeerAccess));
        c->root_maps.push_back({c->h_desc->handle, p});
        *d_mapped = (char*)p + c->h_desc->offset;
        return GST_OK;
    });
}

int gst_comm_exchange_blocks(gst_comm* c, gst_plan* plan, const double* d_src, double* d_dst, int32_t n_blocks, const int32_t* blk_src_rank,
                             const int32_t* blk_dst_rank, const int64_t* blk_src_off, const int64_t* blk_dst_off, const int64_t* blk_count)
{
    return guarded([&]() -> int {
        if (!c) return set_error(GST_EINVAL, "comm is NULL");
        if (n_blocks < 0 || (n_blocks > 0 && (!blk_src_rank || !blk_dst_rank || !blk_src_off || !blk_dst_off || !blk_count)))
            return set_error(GST_EINVAL, "bad block list");
        for (int32_t k = 0; k < n_blocks; k++)      // (received blocks overwrite what an exact Jacobian left in the destination)
            if (blk_dst_rank[k] == c->rank && blk_count[k] > 0 && d_dst) gst::track_touch(d_dst + blk_dst_off[k], (size_t)blk_count[k] * 8);
        int rc;
        hipStream_t st = pick_stream(c, plan, &rc);
        if (rc) return rc;
        return exchange_blocks(c, st, d_src, d_dst, n_blocks, blk_src_rank, blk_dst_rank, blk_src_off, blk_dst_off, blk_count);
    });
}

int gst_comm_allreduce_sum(gst_comm* c, gst_plan* plan, double* d_buf, int64_t n)
{
    return guarded([&]() -> int {
        if (!c || n < 0 || (n > 0 && !d_buf)) return set_error(GST_EINVAL, "bad argument");
        gst::track_touch(d_buf, (size_t)n * 8);
        int rc;
        hipStream_t st = pick_stream(c, plan, &rc);
        if (rc) return rc;
        if (n == 0 || (c->size == 1 && c->transport != GST_TRANSPORT_RCCL)) return GST_OK;
        HIP_TRYC(hipSetDevice(c->device));
        if (c->transport == GST_TRANSPORT_RCCL) {
            NCCL_TRY(c->api, c->api->AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, c->nccl, st));
            return GST_OK;
        }
        // IPC: every rank's copy lands in slot `rank` of every rank's staging array, then the slots are added in rank order
        if (c->stage_n < (size_t)n) {
            HIP_TRYC(hipStreamSynchronize(st));
            if (c->stage) { (void)hipFree(c->stage); c->stage = nullptr; c->stage_n = 0; }
            // (every rank grows its staging array in the same call, so the peers' mappings are refreshed together)
            HIP_TRYC(hipMalloc((void**)&c->stage, (size_t)c->size * (size_t)n * 8));
            c->stage_n = (size_t)n;
        }
        HIP_TRYC(hipMemcpyAsync(c->stage + (size_t)c->rank * c->stage_n, d_buf, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
        std::vector<int32_t> owner((size_t)c->size);
        std::vector<int64_t> row0((size_t)c->size), rows((size_t)c->size, 1);
        for (int r = 0; r < c->size; r++) { owner[(size_t)r] = r; row0[(size_t)r] = r; }
        // rows of stage_n doubles, of which the first n carry data: send whole rows only when n == stage_n
        if ((size_t)n == c->stage_n) {
            Blocks b{c->size, owner.data(), row0.data(), rows.data()};
            if ((rc = exchange_rows(c, st, nullptr, c->stage, (int64_t)c->stage_n, b, -1))) return rc;
        } else {
            // a shorter vector than the staging rows: address in doubles (row length 1)
            for (int r = 0; r < c->size; r++) { row0[(size_t)r] = (int64_t)r * (int64_t)c->stage_n; rows[(size_t)r] = n; }
            Blocks b{c->size, owner.data(), row0.data(), rows.data()};
            if ((rc = exchange_rows(c, st, nullptr, c->stage, 1, b, -1))) return rc;
        }
        HIP_TRYC(gst::launch_sum_slots(c->stage, c->size, (int64_t)c->stage_n, n, d_buf, st));
        HIP_TRYC(hipStreamSynchronize(st));
        return ipc_barrier(c);           // nobody overwrites a staging slot before every rank has summed
    });
}

int gst_comm_barrier(gst_comm* c)
{
    return guarded([&]() -> int {
        if (!c) return set_error(GST_EINVAL, "comm is NULL");
        HIP_TRYC(hipSetDevice(c->device));
        HIP_TRYC(hipStreamSynchronize(c->stream));
        if (c->size == 1) return GST_OK;
        if (c->transport == GST_TRANSPORT_IPC) return ipc_barrier(c);
        // RCCL: a one-element all-reduce on the comm's own stream
        if (!c->stage) { HIP_TRYC(hipMalloc((void**)&c->stage, 8)); c->stage_n = 1; HIP_TRYC(hipMemsetAsync(c->stage, 0, 8, c->stream)); }
        NCCL_TRY(c->api, c->api->AllReduce(c->stage, c->stage, 1, ncclDouble, ncclSum, c->nccl, c->stream));
        HIP_TRYC(hipStreamSynchronize(c->stream));
        return GST_OK;
    });
}

int gst_comm_sync(gst_comm* c)
{
    return guarded([&]() -> int {
        if (!c) return set_error(GST_EINVAL, "comm is NULL");
        HIP_TRYC(hipSetDevice(c->device));
        HIP_TRYC(hipStreamSynchronize(c->stream));
        return GST_OK;
    });
}

int gst_comm_get_info(const gst_comm* c, gst_comm_info* out)
{
    if (!c || !out) return set_error(GST_EINVAL, "NULL argument");
    std::memset(out, 0, sizeof(*out));
    out->transport = c->transport; out->rank = c->rank; out->size = c->size; out->device = c->device;
    out->rccl_version = c->rccl_version;
    out->ipc_opens = (int32_t)std::min<uint64_t>(c->ipc_opens, 0x7fffffffu);
    return GST_OK;
}

}  // extern "C"
