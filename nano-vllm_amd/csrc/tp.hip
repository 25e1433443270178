// tp.hip -- tensor parallelism across the GPUs of one node, one process per GPU.
//
// The split is by attention head and by MLP column: every rank keeps the rows of
// wq/wk/wv that produce its heads and the rows of w_gate/w_up that produce its slice of
// the hidden activation, together with the matching COLUMNS of wo and w_down.  Each
// rank then holds a partial sum of the two projections that return to the residual
// width, and those two partial sums per layer are the only data the ranks exchange inside
// the layers: a sum all-reduce of [tokens, dim] fp32 after wo (reference src/model.c:385)
// and after w_down (src/model.c:392).  The classifier is split by vocabulary rows: every
// rank computes the logits of its slice and the ranks gather either one (maximum, index)
// pair per output row (arg-max hand-off) or the slices themselves (logits for the host or
// for sampling).  Everything else -- scheduler, block tables, allocator, embedding,
// sampler -- runs replicated and deterministically on every rank, so all ranks take the
// same decisions without talking.
//
// The all-reduce is enqueued on the library's own stream between the projection and
// the kernel that adds the result to the residual, so no host synchronisation is
// involved.  It is RCCL's ncclAllReduce (librccl is opened at nvllm_tp_init, never
// before), or a caller-supplied function for hosts that bring their own exchange.
#include <dlfcn.h>

#include <cstring>
#include <rccl/rccl.h>

#include "nvllm_internal.h"

namespace nvllm {

namespace {
struct RcclApi {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
ncclComm_t g_comm = nullptr;

bool open_rccl() {
    if (g_rccl.dl) return true;
    void* dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!dl) dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!dl) {
        fprintf(stderr, "nvllm: cannot open librccl: %s\n", dlerror());
        return false;
    }
    RcclApi a;
    a.dl = dl;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(dl, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(dl, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(dl, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(dl, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(dl, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(dl, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather || !a.GetErrorString) {
        fprintf(stderr, "nvllm: librccl lacks one of the six entry points used here\n");
        dlclose(dl);
        return false;
    }
    g_rccl = a;
    return true;
}

#define RCCL_CHECK(expr)                                                                          \
    do {                                                                                          \
        ncclResult_t r__ = (expr);                                                                \
        if (r__ != ncclSuccess)                                                                   \
            NVLLM_FATAL("RCCL error %d (%s) in %s", (int)r__, g_rccl.GetErrorString(r__), #expr); \
    } while (0)
}  // namespace

TpState& tp() {
    static TpState s;
    return s;
}

void tp_allreduce(float* buf, size_t count) {
    TpState& s = tp();
    if (!s.active || count == 0) return;
    if (s.fn) {
        s.fn(s.ctx, buf, count, (void*)rt().stream);
        return;
    }
    RCCL_CHECK(g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, g_comm, rt().stream));
}

void tp_allgather(const float* send, float* recv, size_t count) {
    TpState& s = tp();
    if (count == 0) return;
    if (!s.active || (s.fn && s.world == 1)) {   // (a one-rank RCCL communicator still goes through RCCL)
        HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, rt().stream));
        return;
    }
    if (s.fn) {
        // a caller-supplied exchange only sums: gather = sum of slices padded with zeros
        // (x + 0 is x for every float these buffers carry)
        HIP_CHECK(hipMemsetAsync(recv, 0, (size_t)s.world * count * sizeof(float), rt().stream));
        HIP_CHECK(hipMemcpyAsync(recv + (size_t)s.rank * count, send, count * sizeof(float), hipMemcpyDeviceToDevice, rt().stream));
        s.fn(s.ctx, recv, (size_t)s.world * count, (void*)rt().stream);
        return;
    }
    RCCL_CHECK(g_rccl.AllGather(send, recv, count, ncclFloat32, g_comm, rt().stream));
}

}  // namespace nvllm

using namespace nvllm;

extern "C" {

int nvllm_tp_shard(const Config* cfg, int rank, int world, NvllmTpShard* out) {
    if (!cfg || !out || world < 1 || rank < 0 || rank >= world) return -1;
    if (cfg->n_heads <= 0 || cfg->n_kv_heads <= 0 || cfg->dim % cfg->n_heads) return -1;
    const int hd = cfg->dim / cfg->n_heads;
    // whole kv heads per rank (and with them whole groups of query heads), whole
    // 32-column groups of the hidden activation per rank
    if (cfg->n_kv_heads % world || cfg->n_heads % world || cfg->hidden_dim % (32 * world)) return -1;
    out->rank = rank;
    out->world = world;
    out->n_heads = cfg->n_heads / world;
    out->head0 = rank * out->n_heads;
    out->n_kv_heads = cfg->n_kv_heads / world;
    out->kv_head0 = rank * out->n_kv_heads;
    out->hidden = cfg->hidden_dim / world;
    out->hidden0 = rank * out->hidden;
    // classifier rows: equal slices of whole 16-row blocks; the last ranks may hold fewer (or none)
    const int per = ((cfg->vocab_size + 15) / 16 + world - 1) / world * 16;
    out->vocab_slice = per;
    out->vocab0 = rank * per < cfg->vocab_size ? rank * per : cfg->vocab_size;
    out->vocab = cfg->vocab_size - out->vocab0 < per ? cfg->vocab_size - out->vocab0 : per;
    if (world > 1 && ((out->n_heads * hd) % 32 || (out->n_kv_heads * hd) % 16)) return -1;
    return 0;
}

int nvllm_tp_unique_id(void* id) {
    if (!id || !open_rccl()) return -1;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return -1;
    static_assert(sizeof(u) == NVLLM_TP_ID_BYTES, "RCCL unique id size");
    memcpy(id, &u, sizeof u);
    return 0;
}

static int tp_begin(int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) {
        fprintf(stderr, "nvllm: tensor-parallel rank %d of %d is not a valid placement\n", rank, world);
        return -1;
    }
    if (tp().active) {
        fprintf(stderr, "nvllm: tensor parallelism is already set up; call nvllm_tp_finalize first\n");
        return -1;
    }
    ensure_init();
    return 0;
}

int nvllm_tp_init(int rank, int world, const void* id) {
    if (!id || tp_begin(rank, world) != 0 || !open_rccl()) return -1;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RCCL_CHECK(g_rccl.CommInitRank(&g_comm, world, u, rank));
    TpState& s = tp();
    s.rank = rank;
    s.world = world;
    s.fn = nullptr;
    s.ctx = nullptr;
    s.active = true;
    return 0;
}

int nvllm_tp_init_custom(int rank, int world, nvllm_allreduce_fn fn, void* ctx) {
    if (!fn || tp_begin(rank, world) != 0) return -1;
    TpState& s = tp();
    s.rank = rank;
    s.world = world;
    s.fn = fn;
    s.ctx = ctx;
    s.active = true;
    return 0;
}

void nvllm_tp_finalize(void) {
    TpState& s = tp();
    if (!s.active) return;
    HIP_CHECK(hipStreamSynchronize(rt().stream));
    tp_peer_finalize();
    if (g_comm) {
        g_rccl.CommDestroy(g_comm);
        g_comm = nullptr;
    }
    s = TpState();
}

int nvllm_tp_rank(void) { return tp().rank; }
int nvllm_tp_world(void) { return tp().world; }

}  // extern "C"
