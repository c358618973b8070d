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

#include <algorithm>
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
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
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
        AGAN_SYM(Send, ncclSend);
        AGAN_SYM(Recv, ncclRecv);
        AGAN_SYM(GroupStart, ncclGroupStart);
        AGAN_SYM(GroupEnd, ncclGroupEnd);
        AGAN_SYM(GetErrorString, ncclGetErrorString);
#undef AGAN_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.Send && r.Recv &&
               r.GroupStart && r.GroupEnd && r.GetErrorString;
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

// ---- 16-bit wire format of a gradient bucket (AGAN_DT_BF16): HBM-bound element-wise passes, 16 bytes per lane -------------------
// Every rank rounds its fp32 bucket to bf16 (pack), receives piece `rank` of every rank, adds the `world` pieces IN RANK ORDER IN FP32
// and rounds the sum to bf16 once (sum), and widens the gathered sums back to fp32 (unpack):
//     result = fp32(bf16(sum_r fp32(bf16(g_r))))      -- identical on every rank, error independent of the number of ranks.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {       // round to nearest even (v_cvt_pk_bf16_f32; keeps a NaN a NaN)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// n is a multiple of 4 (buckets are 16-byte aligned slices), n_wire >= n a multiple of 4: the tail [n, n_wire) is zero padding
__global__ __launch_bounds__(256) void exchange_pack_bf16_kernel(const float* __restrict__ g, unsigned* __restrict__ wire, size_t n4, size_t nw4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nw4; i += (size_t)gridDim.x * 256) {
        u32x2_t o = {0u, 0u};
        if (i < n4) {
            const f32x4_t v = reinterpret_cast<const f32x4_t*>(g)[i];
            o[0] = pack_bf16x2(v[0], v[1]);
            o[1] = pack_bf16x2(v[2], v[3]);
        }
        reinterpret_cast<u32x2_t*>(wire)[i] = o;
    }
}
// pieces: [world][per] bf16 (piece r came from rank r); sum: [per] bf16.  per is a multiple of 4.
__global__ __launch_bounds__(256) void exchange_sum_bf16_kernel(const unsigned* __restrict__ pieces, int world, size_t per4, unsigned* __restrict__ sum) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per4; i += (size_t)gridDim.x * 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int r = 0; r < world; ++r) {                                // rank order, fp32
            const u32x2_t v = reinterpret_cast<const u32x2_t*>(pieces)[(size_t)r * per4 + i];
            if (r == 0) { a0 = bf16_lo(v[0]); a1 = bf16_hi(v[0]); a2 = bf16_lo(v[1]); a3 = bf16_hi(v[1]); }
            else { a0 += bf16_lo(v[0]); a1 += bf16_hi(v[0]); a2 += bf16_lo(v[1]); a3 += bf16_hi(v[1]); }
        }
        reinterpret_cast<u32x2_t*>(sum)[i] = u32x2_t{pack_bf16x2(a0, a1), pack_bf16x2(a2, a3)};
    }
}
__global__ __launch_bounds__(256) void exchange_unpack_bf16_kernel(const unsigned* __restrict__ wire, float* __restrict__ g, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const u32x2_t v = reinterpret_cast<const u32x2_t*>(wire)[i];
        reinterpret_cast<f32x4_t*>(g)[i] = f32x4_t{bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])};
    }
}
int ew_grid(size_t work) { return (int)std::max<size_t>(1, std::min<size_t>((work + 255) / 256, 256 * 8)); }

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

size_t agan_exchange_wire_elems(size_t n, int world) {
    // 16-bit wire image of a bucket of n floats: `world` equal pieces of a multiple of 8 elements (16-byte aligned pieces)
    if (world < 1 || n == 0) return 0;
    const size_t per = ((n + (size_t)world - 1) / (size_t)world + 7) / 8 * 8;
    return per * (size_t)world;
}

int agan_exchange_pack_bf16(const float* g, void* wire, size_t n, size_t n_wire, void* stream) {
    AGAN_REQUIRE(g && wire && n > 0 && n_wire >= n, "exchange_pack_bf16: bad argument");
    AGAN_REQUIRE(n % 4 == 0 && n_wire % 4 == 0, "exchange_pack_bf16: element counts must be multiples of 4 (16-byte aligned buckets)");
    hipLaunchKernelGGL(exchange_pack_bf16_kernel, dim3(ew_grid(n_wire / 4)), dim3(256), 0, as_stream(stream), g, static_cast<unsigned*>(wire), n / 4, n_wire / 4);
    return check_launch("exchange_pack_bf16");
}

int agan_exchange_sum_bf16(const void* pieces, int world, size_t per, void* sum, void* stream) {
    AGAN_REQUIRE(pieces && sum && world >= 1 && per > 0 && per % 4 == 0, "exchange_sum_bf16: bad argument");
    hipLaunchKernelGGL(exchange_sum_bf16_kernel, dim3(ew_grid(per / 4)), dim3(256), 0, as_stream(stream), static_cast<const unsigned*>(pieces), world, per / 4,
                       static_cast<unsigned*>(sum));
    return check_launch("exchange_sum_bf16");
}

int agan_exchange_unpack_bf16(const void* wire, float* g, size_t n, void* stream) {
    AGAN_REQUIRE(wire && g && n > 0 && n % 4 == 0, "exchange_unpack_bf16: bad argument");
    hipLaunchKernelGGL(exchange_unpack_bf16_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), static_cast<const unsigned*>(wire), g, n / 4);
    return check_launch("exchange_unpack_bf16");
}

size_t agan_allreduce_scratch_bytes(size_t n, int world, int wire_dtype) {
    if (wire_dtype != AGAN_DT_BF16) return 0;
    // send image [world][per] + received pieces [world][per] + this rank's sum [per], bf16
    const size_t nw = agan_exchange_wire_elems(n, world);
    return nw == 0 ? 0 : (2 * nw + nw / (size_t)world) * 2;
}

int agan_allreduce_bucket_dt(void* comm, float* buf, size_t n, int wire_dtype, void* scratch, size_t scratch_bytes, void* stream) {
    if (wire_dtype == AGAN_DT_F32) return agan_allreduce_bucket(comm, buf, n, stream);
    AGAN_REQUIRE(wire_dtype == AGAN_DT_BF16, "allreduce_bucket: wire dtype %d (AGAN_DT_F32 or AGAN_DT_BF16)", wire_dtype);
    AGAN_REQUIRE(comm && buf && n > 0 && n % 4 == 0, "allreduce_bucket: bad argument");
    Comm* c = static_cast<Comm*>(comm);
    const size_t nw = agan_exchange_wire_elems(n, c->world), per = nw / (size_t)c->world;
    if (!scratch || scratch_bytes < agan_allreduce_scratch_bytes(n, c->world, wire_dtype)) {
        set_error("allreduce_bucket: scratch too small (agan_allreduce_scratch_bytes)");
        return AGAN_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    unsigned short* send = static_cast<unsigned short*>(scratch);
    unsigned short* recv = send + nw;
    unsigned short* mine = recv + nw;
    if (int e = agan_exchange_pack_bf16(buf, send, n, nw, stream)) return e;
    // all-to-all of the pieces: piece j of every rank goes to rank j (grouped point-to-point: one transfer per xGMI link)
    ncclResult_t rc = rccl().GroupStart();
    if (rc != ncclSuccess) return fail("allreduce_bucket/group_start", rc);
    for (int r = 0; r < c->world; ++r) {
        rc = rccl().Send(send + (size_t)r * per, per, ncclBfloat16, r, c->comm, st);
        if (rc != ncclSuccess) return fail("allreduce_bucket/send", rc);
        rc = rccl().Recv(recv + (size_t)r * per, per, ncclBfloat16, r, c->comm, st);
        if (rc != ncclSuccess) return fail("allreduce_bucket/recv", rc);
    }
    rc = rccl().GroupEnd();
    if (rc != ncclSuccess) return fail("allreduce_bucket/group_end", rc);
    if (int e = agan_exchange_sum_bf16(recv, c->world, per, mine, stream)) return e;
    rc = rccl().AllGather(mine, send, per, ncclBfloat16, c->comm, st);          // the send image is free again: it receives the sums
    if (rc != ncclSuccess) return fail("allreduce_bucket/all_gather", rc);
    return agan_exchange_unpack_bf16(send, buf, n, stream);
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
