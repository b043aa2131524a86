// comm.hip.h -- the communicator behind the multi-GPU search: one process
// per GPU, RCCL over xGMI bound directly (dlopen of librccl.so: a single-GPU user never loads it),
// nothing else.  The reference is single-process and has no counterpart.
//
// Everything the search exchanges is small and latency-bound, so the engine needs three host-level
// primitives only: an all-gather of one fixed-size record per rank, and a matched send / recv of a
// block of node records when a shard runs dry.  Two transports implement them:
//   * RCCL (the product): pinned host buffer -> HBM -> ncclAllGather / ncclSend / ncclRecv on the
//     communicator's own high-priority stream -> pinned host buffer.  The all-gather can be POSTED and
//     collected later, so a rank's exchange overlaps its node-LP launches.
//   * custom: three caller-supplied callbacks on host buffers (the CPU tests and the one-GPU
//     rehearsal plug a gloo process group in here; RCCL refuses two ranks on one device).
// Included by mipx.hip (needs mipx_ctx, HIP_TRY, fail).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

struct mipx_comm {
    mipx_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    bool rccl = false;
    // RCCL transport
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    char *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;   // all-gather staging
    size_t cap_send = 0, cap_recv = 0;
    char *d_msg = nullptr;   // send / recv staging (device)
    size_t cap_msg = 0;
    // custom transport
    mipx_comm_ops ops{};
    void *user = nullptr;
    std::vector<char> c_send, c_recv;
    // posted all-gather
    bool pending = false;
    size_t pending_bytes = 0;
};

namespace {

RcclApi g_rccl;

int rccl_load(mipx_ctx *ctx) {
    if (g_rccl.lib) return MIPX_OK;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(ctx, MIPX_ENODEV, "mipx_comm: librccl.so not found (needed for more than one GPU)");
    RcclApi a;
    a.lib = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.Send || !a.Recv || !a.GroupStart ||
        !a.GroupEnd)
        return fail(ctx, MIPX_ENODEV, "mipx_comm: librccl.so lacks a needed entry point");
    g_rccl = a;
    return MIPX_OK;
}

int rccl_fail(mipx_ctx *ctx, const char *what, ncclResult_t r) {
    if (ctx) {
        ctx->err = what;
        ctx->err += ": ";
        ctx->err += g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error";
    }
    return MIPX_EHIP;
}
#define RCCL_TRY(ctx, call)                                   \
    do {                                                      \
        ncclResult_t r_ = (call);                             \
        if (r_ != ncclSuccess) return rccl_fail((ctx), #call, r_); \
    } while (0)

int comm_reserve(mipx_comm *c, size_t send_bytes, size_t recv_bytes) {
    mipx_ctx *ctx = c->ctx;
    if (send_bytes > c->cap_send) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->h_send) (void)hipHostFree(c->h_send);
        c->d_send = nullptr; c->h_send = nullptr; c->cap_send = 0;
        HIP_TRY(ctx, hipMalloc((void **)&c->d_send, send_bytes));
        HIP_TRY(ctx, hipHostMalloc((void **)&c->h_send, send_bytes, hipHostMallocDefault));
        c->cap_send = send_bytes;
    }
    if (recv_bytes > c->cap_recv) {
        if (c->d_recv) (void)hipFree(c->d_recv);
        if (c->h_recv) (void)hipHostFree(c->h_recv);
        c->d_recv = nullptr; c->h_recv = nullptr; c->cap_recv = 0;
        HIP_TRY(ctx, hipMalloc((void **)&c->d_recv, recv_bytes));
        HIP_TRY(ctx, hipHostMalloc((void **)&c->h_recv, recv_bytes, hipHostMallocDefault));
        c->cap_recv = recv_bytes;
    }
    return MIPX_OK;
}

// Post the all-gather of `bytes` per rank (RCCL: queued on the communicator's stream, the host does
// not wait; custom: done right here).  comm_collect() returns the gathered block (world x bytes).
int comm_post(mipx_comm *c, const void *send, size_t bytes) {
    mipx_ctx *ctx = c->ctx;
    if (c->pending) return fail(ctx, MIPX_EINVAL, "mipx_comm: an all-gather is already posted");
    const size_t all = bytes * (size_t)c->world;
    if (c->rccl) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        int rc = comm_reserve(c, bytes, all);
        if (rc) return rc;
        std::memcpy(c->h_send, send, bytes);
        HIP_TRY(ctx, hipMemcpyAsync(c->d_send, c->h_send, bytes, hipMemcpyHostToDevice, c->stream));
        RCCL_TRY(ctx, g_rccl.AllGather(c->d_send, c->d_recv, bytes, ncclChar, c->nccl, c->stream));
        HIP_TRY(ctx, hipMemcpyAsync(c->h_recv, c->d_recv, all, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(ctx, hipEventRecord(c->done, c->stream));
    } else {
        c->c_recv.resize(all);
        if (c->world == 1) std::memcpy(c->c_recv.data(), send, bytes);
        else if (c->ops.allgather(c->user, send, c->c_recv.data(), bytes) != 0)
            return fail(ctx, MIPX_EHIP, "mipx_comm: the custom all-gather failed");
    }
    c->pending = true;
    c->pending_bytes = bytes;
    return MIPX_OK;
}

int comm_collect(mipx_comm *c, const char **gathered) {
    mipx_ctx *ctx = c->ctx;
    if (!c->pending) return fail(ctx, MIPX_EINVAL, "mipx_comm: nothing posted");
    if (c->rccl) {
        HIP_TRY(ctx, hipEventSynchronize(c->done));
        *gathered = c->h_recv;
    } else {
        *gathered = c->c_recv.data();
    }
    c->pending = false;
    return MIPX_OK;
}

// A block of bytes that lives in HBM at `dev` goes to / comes from `peer` (matched calls, in the same
// order on both sides).  RCCL moves it device to device over xGMI; custom stages it through the host.
int comm_send_dev(mipx_comm *c, int peer, const void *dev, size_t bytes) {
    mipx_ctx *ctx = c->ctx;
    if (c->rccl) {
        RCCL_TRY(ctx, g_rccl.Send(dev, bytes, ncclChar, peer, c->nccl, c->stream));
        HIP_TRY(ctx, hipStreamSynchronize(c->stream));
        return MIPX_OK;
    }
    c->c_send.resize(bytes);
    HIP_TRY(ctx, hipMemcpy(c->c_send.data(), dev, bytes, hipMemcpyDeviceToHost));
    if (c->ops.send(c->user, peer, c->c_send.data(), bytes) != 0) return fail(ctx, MIPX_EHIP, "mipx_comm: the custom send failed");
    return MIPX_OK;
}

int comm_recv_dev(mipx_comm *c, int peer, void *dev, size_t bytes) {
    mipx_ctx *ctx = c->ctx;
    if (c->rccl) {
        RCCL_TRY(ctx, g_rccl.Recv(dev, bytes, ncclChar, peer, c->nccl, c->stream));
        HIP_TRY(ctx, hipStreamSynchronize(c->stream));
        return MIPX_OK;
    }
    c->c_send.resize(bytes);
    if (c->ops.recv(c->user, peer, c->c_send.data(), bytes) != 0) return fail(ctx, MIPX_EHIP, "mipx_comm: the custom recv failed");
    HIP_TRY(ctx, hipMemcpy(dev, c->c_send.data(), bytes, hipMemcpyHostToDevice));
    return MIPX_OK;
}

// A block goes from this rank to itself through the point-to-point path (the send and the matching
// receive fused in one group, as RCCL requires of a rank that talks to itself): what a migration does
// between two ranks, on one.  Custom transport: a device-to-device copy.
int comm_sendrecv_self(mipx_comm *c, const void *src, void *dst, size_t bytes) {
    mipx_ctx *ctx = c->ctx;
    if (c->rccl) {
        RCCL_TRY(ctx, g_rccl.GroupStart());
        const ncclResult_t rs = g_rccl.Send(src, bytes, ncclChar, c->rank, c->nccl, c->stream);
        const ncclResult_t rr = rs == ncclSuccess ? g_rccl.Recv(dst, bytes, ncclChar, c->rank, c->nccl, c->stream) : rs;
        const ncclResult_t re = g_rccl.GroupEnd();
        if (rs != ncclSuccess) return rccl_fail(ctx, "ncclSend (to self)", rs);
        if (rr != ncclSuccess) return rccl_fail(ctx, "ncclRecv (from self)", rr);
        if (re != ncclSuccess) return rccl_fail(ctx, "ncclGroupEnd", re);
        HIP_TRY(ctx, hipStreamSynchronize(c->stream));
        return MIPX_OK;
    }
    HIP_TRY(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return MIPX_OK;
}

int comm_msg_buffer(mipx_comm *c, size_t bytes, char **dev) {
    if (bytes > c->cap_msg) {
        if (c->d_msg) (void)hipFree(c->d_msg);
        c->d_msg = nullptr; c->cap_msg = 0;
        HIP_TRY(c->ctx, hipMalloc((void **)&c->d_msg, bytes));
        c->cap_msg = bytes;
    }
    *dev = c->d_msg;
    return MIPX_OK;
}

}  // namespace

extern "C" {

int mipx_comm_unique_id(char id[128]) {
    if (!id) return MIPX_EINVAL;
    int rc = rccl_load(nullptr);
    if (rc) return rc;
    ncclUniqueId u;
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return MIPX_EHIP;
    std::memcpy(id, &u, 128);
    return MIPX_OK;
}

int mipx_comm_create_rccl(mipx_ctx *ctx, const char id[128], int rank, int world, mipx_comm **out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world)
        return fail(ctx, MIPX_EINVAL, "mipx_comm_create_rccl: bad argument");
    *out = nullptr;
    int rc = rccl_load(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mipx_comm *c = new (std::nothrow) mipx_comm();
    if (!c) return fail(ctx, MIPX_ENOMEM, "mipx_comm_create_rccl: host alloc");
    c->ctx = ctx; c->rank = rank; c->world = world; c->rccl = true;
    // the exchange must overtake the node-LP launches queued on the context's stream
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) hi = 0;
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        mipx_comm_destroy(c);
        return fail(ctx, MIPX_EHIP, "mipx_comm_create_rccl: stream");
    }
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) {
        c->nccl = nullptr;
        mipx_comm_destroy(c);
        return rccl_fail(ctx, "ncclCommInitRank", r);
    }
    *out = c;
    return MIPX_OK;
}

int mipx_comm_create_custom(mipx_ctx *ctx, int rank, int world, const mipx_comm_ops *ops, void *user,
                            mipx_comm **out) {
    // (ctx may be NULL: a host-only communicator -- all-gather and barrier -- for the CPU tests)
    if (!out || world < 1 || rank < 0 || rank >= world ||
        (world > 1 && (!ops || !ops->allgather || !ops->send || !ops->recv)))
        return fail(ctx, MIPX_EINVAL, "mipx_comm_create_custom: bad argument");
    mipx_comm *c = new (std::nothrow) mipx_comm();
    if (!c) return fail(ctx, MIPX_ENOMEM, "mipx_comm_create_custom: host alloc");
    c->ctx = ctx; c->rank = rank; c->world = world; c->rccl = false;
    if (ops) c->ops = *ops;
    c->user = user;
    *out = c;
    return MIPX_OK;
}

void mipx_comm_destroy(mipx_comm *c) {
    if (!c) return;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    void *dp[] = {c->d_send, c->d_recv, c->d_msg};
    for (void *q : dp)
        if (q) (void)hipFree(q);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    delete c;
}

int mipx_comm_rank(const mipx_comm *c) { return c ? c->rank : MIPX_EINVAL; }
int mipx_comm_size(const mipx_comm *c) { return c ? c->world : MIPX_EINVAL; }

int mipx_comm_allgather(mipx_comm *c, const void *send, void *recv, size_t bytes_per_rank) {
    if (!c || !send || !recv || bytes_per_rank == 0) return MIPX_EINVAL;
    int rc = comm_post(c, send, bytes_per_rank);
    if (rc) return rc;
    const char *g = nullptr;
    if ((rc = comm_collect(c, &g))) return rc;
    std::memcpy(recv, g, bytes_per_rank * (size_t)c->world);
    return MIPX_OK;
}

int mipx_comm_barrier(mipx_comm *c) {
    if (!c) return MIPX_EINVAL;
    if (c->ctx && c->ctx->stream) HIP_TRY(c->ctx, hipStreamSynchronize(c->ctx->stream));
    int64_t mine = c->rank;
    std::vector<int64_t> all((size_t)c->world);
    return mipx_comm_allgather(c, &mine, all.data(), sizeof mine);
}

}  // extern "C"
