// Data-parallel gradient exchange driven directly over RCCL (SURVEY.md section 8b `agan_allreduce_bucket`, section 8e).
//
// The reference is single-GPU; this is the new component's C-ABI form: a communicator per process (one process per GPU) and an
// in-place SUM all-reduce of one contiguous gradient bucket, issued as reduce-scatter + all-gather on the caller's stream -- on the
// point-to-point xGMI fabric (7 links per GPU) each rank then owns 1/world of the bucket and the two halves move (world-1)/world of the
// bytes each, instead of whatever ring the library's heuristic picks for one all-reduce call.
// RCCL is bound at run time (dlopen of the librccl the process already carries -- torch's -- or the ROCm one): the kernels library has
// no link-time dependency on it, and a box without RCCL only loses these four entry points.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "agan_common.h"

using namespace agan;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // first the copy the process ALREADY carries (torch's): RTLD_NOLOAD never maps a second RCCL beside it; only a process
        // without one (a plain C++ host) loads the ROCm library
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.handle) break;
        }
        for (const char* name : names) {
            if (r.handle) break;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) return;
#define AGAN_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, #sym))
        AGAN_SYM(GetUniqueId, ncclGetUniqueId);
        AGAN_SYM(CommInitRank, ncclCommInitRank);
        AGAN_SYM(CommDestroy, ncclCommDestroy);
        AGAN_SYM(AllReduce, ncclAllReduce);
        AGAN_SYM(ReduceScatter, ncclReduceScatter);
        AGAN_SYM(AllGather, ncclAllGather);
        AGAN_SYM(GetErrorString, ncclGetErrorString);
#undef AGAN_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.GetErrorString;
    });
    return r;
}

struct Comm {
    ncclComm_t comm;
    int rank, world;
};

int fail(const char* what, ncclResult_t rc) {
    set_error("%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
    return AGAN_ELAUNCH;
}

}  // namespace

extern "C" {

int agan_comm_unique_id(void* id) {
    AGAN_REQUIRE(id != nullptr, "comm_unique_id: null pointer");
    AGAN_REQUIRE(rccl().ok, "comm_unique_id: librccl not found (dlopen)");
    static_assert(sizeof(ncclUniqueId) == AGAN_COMM_ID_BYTES, "AGAN_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId u;
    const ncclResult_t rc = rccl().GetUniqueId(&u);
    if (rc != ncclSuccess) return fail("comm_unique_id", rc);
    std::memcpy(id, &u, sizeof(u));
    return AGAN_OK;
}

int agan_comm_init(void** comm, int rank, int world, const void* id) {
    AGAN_REQUIRE(comm && id, "comm_init: null pointer");
    AGAN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    AGAN_REQUIRE(rccl().ok, "comm_init: librccl not found (dlopen)");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    Comm* c = new Comm{nullptr, rank, world};
    const ncclResult_t rc = rccl().CommInitRank(&c->comm, world, u, rank);     // binds the calling thread's current device
    if (rc != ncclSuccess) {
        delete c;
        return fail("comm_init", rc);
    }
    *comm = c;
    return AGAN_OK;
}

int agan_comm_destroy(void* comm) {
    if (!comm) return AGAN_OK;
    Comm* c = static_cast<Comm*>(comm);
    const ncclResult_t rc = rccl().CommDestroy(c->comm);
    delete c;
    return rc == ncclSuccess ? AGAN_OK : fail("comm_destroy", rc);
}

size_t agan_allreduce_chunk_elems(size_t n, int world) {
    // buckets are slices of FlatAdam's flat gradient buffer (16-byte aligned parameters): an even split into 16-byte aligned
    // chunks, one per rank (rank r owns [r * chunk, (r + 1) * chunk)), or 0 = no such split -> the library's all-reduce
    if (world < 1 || n == 0) return 0;
    if (n % (size_t)world != 0 || (n / (size_t)world) % 4 != 0) return 0;
    return n / (size_t)world;
}

int agan_allreduce_bucket(void* comm, float* buf, size_t n, void* stream) {
    AGAN_REQUIRE(comm && buf && n > 0, "allreduce_bucket: bad argument");
    Comm* c = static_cast<Comm*>(comm);
    hipStream_t st = as_stream(stream);
    if (const size_t chunk = agan_allreduce_chunk_elems(n, c->world)) {
        float* mine = buf + (size_t)c->rank * chunk;
        ncclResult_t rc = rccl().ReduceScatter(buf, mine, chunk, ncclFloat, ncclSum, c->comm, st);      // in place: recv = send + rank * chunk
        if (rc != ncclSuccess) return fail("allreduce_bucket/reduce_scatter", rc);
        rc = rccl().AllGather(mine, buf, chunk, ncclFloat, c->comm, st);                                 // in place: send = recv + rank * chunk
        if (rc != ncclSuccess) return fail("allreduce_bucket/all_gather", rc);
        return AGAN_OK;
    }
    const ncclResult_t rc = rccl().AllReduce(buf, buf, n, ncclFloat, ncclSum, c->comm, st);
    return rc == ncclSuccess ? AGAN_OK : fail("allreduce_bucket/all_reduce", rc);
}

}  // extern "C"
