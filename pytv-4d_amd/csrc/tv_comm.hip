// tv_comm.hip -- multi-GPU surface of the C-ABI (include/pytv4d.h): z-slab neighbours over RCCL, one process per GPU.
//
// The reference has no multi-GPU code; its layout remark (README.md:235: "(Nz, M, N, N) ... can be
// decomposed easily along z") is what this acts on.  Rank r owns the planes [z0, z0 + nz) and, per operator apply,
// trades ONE boundary plane with each z-neighbour: a chain, not a ring -- nearest-neighbour ncclSend / ncclRecv in one
// group, on the caller's stream, plus an fp64 all-reduce for the few scalars (TV, fidelity, CG dots).  No torch types,
// no Python: a C / C++ host can shard with these four calls (examples/cabi_demo.cpp); pytv/slab.py is one more caller.
//
// RCCL is bound at RUN time (dlopen): the library has no link-time dependency on it, single-GPU hosts never load it,
// and inside a PyTorch process the already-loaded librccl is reused (one RCCL per process).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "tv_host.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* forced = getenv("TV_RCCL_LIB");        // read once, here
        const char* names[] = {"librccl.so", "librccl.so.1"};
        if (forced && *forced) r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        for (const char* n : names)                        // already in the process (PyTorch's copy)?
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char* n : {"librccl.so.1", "librccl.so"})
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) { r.error = std::string("cannot load librccl: ") + dlerror(); return; }
#define TV_SYM(field, name)                                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));                       \
    if (!r.field && r.error.empty()) r.error = std::string("librccl lacks ") + name;
        TV_SYM(GetUniqueId, "ncclGetUniqueId") TV_SYM(CommInitRank, "ncclCommInitRank") TV_SYM(CommDestroy, "ncclCommDestroy")
        TV_SYM(GroupStart, "ncclGroupStart") TV_SYM(GroupEnd, "ncclGroupEnd") TV_SYM(Send, "ncclSend") TV_SYM(Recv, "ncclRecv")
        TV_SYM(AllReduce, "ncclAllReduce") TV_SYM(GetErrorString, "ncclGetErrorString")
#undef TV_SYM
    });
    return r;
}

int ncclfail(ncclResult_t e, const char* where) {
    Rccl& r = rccl();
    g_err = std::string(where) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error");
    return 1000 + (int)e;            // > 0 like hipError_t codes, offset so that the two ranges do not overlap
}
#define NCCL_TRY(call)                                         \
    do {                                                       \
        ncclResult_t e__ = (call);                             \
        if (e__ != ncclSuccess) return ncclfail(e__, #call);   \
    } while (0)

}  // namespace

struct tv_ctx {
    ncclComm_t comm;
    int rank, nranks, device;
};

extern "C" {

int tv_ctx_unique_id(void* id_out) {
    if (id_out == nullptr) return fail(TV_E_ARG, "NULL id buffer");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(TV_E_ARG, r.error.c_str());
    ncclUniqueId id;
    NCCL_TRY(r.GetUniqueId(&id));
    static_assert(sizeof(id) == TV_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int tv_ctx_create(tv_ctx** out, int rank, int nranks, const void* unique_id, int device) {
    if (out == nullptr || unique_id == nullptr || nranks < 1 || rank < 0 || rank >= nranks || device < 0)
        return fail(TV_E_ARG, "tv_ctx_create: bad argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(TV_E_ARG, r.error.c_str());
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm;
    NCCL_TRY(r.CommInitRank(&comm, nranks, id, rank));
    *out = new tv_ctx{comm, rank, nranks, device};
    return 0;
}

int tv_ctx_destroy(tv_ctx* ctx) {
    if (ctx == nullptr) return 0;
    Rccl& r = rccl();
    ncclResult_t e = r.CommDestroy(ctx->comm);
    delete ctx;
    if (e != ncclSuccess) return ncclfail(e, "ncclCommDestroy");
    return 0;
}

int tv_ctx_rank(const tv_ctx* ctx) { return ctx ? ctx->rank : TV_E_ARG; }
int tv_ctx_size(const tv_ctx* ctx) { return ctx ? ctx->nranks : TV_E_ARG; }

int tv_halo_exchange(tv_ctx* ctx, int32_t dtype, int64_t count, int prev_rank, int next_rank, const void* send_prev,
                     const void* send_next, void* recv_prev, void* recv_next, void* stream) {
    if (ctx == nullptr || count < 0) return fail(TV_E_ARG, "tv_halo_exchange: bad argument");
    if (dtype != TV_F32 && dtype != TV_F64) return fail(TV_E_ARG, "unknown dtype");
    if (prev_rank >= ctx->nranks || next_rank >= ctx->nranks) return fail(TV_E_ARG, "neighbour rank outside the communicator");
    Rccl& r = rccl();
    const ncclDataType_t ty = (dtype == TV_F32) ? ncclFloat32 : ncclFloat64;
    hipStream_t st = (hipStream_t)stream;
    const bool p = prev_rank >= 0, n = next_rank >= 0;
    if (count == 0 || (!(p && (send_prev || recv_prev)) && !(n && (send_next || recv_next)))) return 0;
    // one group: both directions progress together (no send / receive ordering deadlock between neighbours)
    NCCL_TRY(r.GroupStart());
    ncclResult_t e = ncclSuccess;
    if (p && recv_prev && e == ncclSuccess) e = r.Recv(recv_prev, (size_t)count, ty, prev_rank, ctx->comm, st);
    if (p && send_prev && e == ncclSuccess) e = r.Send(send_prev, (size_t)count, ty, prev_rank, ctx->comm, st);
    if (n && send_next && e == ncclSuccess) e = r.Send(send_next, (size_t)count, ty, next_rank, ctx->comm, st);
    if (n && recv_next && e == ncclSuccess) e = r.Recv(recv_next, (size_t)count, ty, next_rank, ctx->comm, st);
    ncclResult_t e2 = r.GroupEnd();
    if (e != ncclSuccess) return ncclfail(e, "ncclSend / ncclRecv");
    if (e2 != ncclSuccess) return ncclfail(e2, "ncclGroupEnd");
    return 0;
}

int tv_allreduce_f64(tv_ctx* ctx, double* buf, int64_t n, int32_t op, void* stream) {
    if (ctx == nullptr || buf == nullptr || n < 0 || (op != TV_SUM && op != TV_MAX)) return fail(TV_E_ARG, "tv_allreduce_f64: bad argument");
    if (n == 0) return 0;
    Rccl& r = rccl();
    NCCL_TRY(r.AllReduce(buf, buf, (size_t)n, ncclFloat64, op == TV_SUM ? ncclSum : ncclMax, ctx->comm, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
