// noalloc.hip -- nvllm_matmul_noalloc: ops.h matmul() for callers that cannot allocate
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
//
// matmul() on a load_model handle needs device memory beyond its operands: the fp32
// activations are re-packed as fp16 hi/lo tiles for the matrix cores, a K split leaves
// partial tiles in a slab, and the slices of a tile meet at an arrival counter.  matmul()
// takes those from the library's grow-only buffers (hipMalloc on first use or growth);
// this entry point takes them from one file-scope array in device memory instead, so it
// calls no hipMalloc / hipFree and nothing that allocates.  The array has a fixed size,
// hence the size limits below; each GPU the library runs on holds its own copy until the
// process ends.
//
// Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two host
// threads, or two streams) race.  The library enqueues everything on its single stream,
// so consecutive calls are ordered.
#include "nvllm_internal.h"

namespace nvllm {

constexpr int NOALLOC_MAX_TOKENS = 64;
constexpr int NOALLOC_MAX_IN_DIM = 8192;
constexpr int NOALLOC_COUNTERS = 16384;
constexpr size_t NOALLOC_COUNTER_BYTES = (size_t)NOALLOC_COUNTERS * sizeof(unsigned);                   // 64 KiB
constexpr size_t NOALLOC_A16_BYTES = (size_t)(NOALLOC_MAX_TOKENS / 16) * (NOALLOC_MAX_IN_DIM / 32) * 2 *
                                     TILE_HALVES * sizeof(__half);                                      // 2 MiB
// Partial tiles of a K split.  The split matmul() picks by itself makes at most about 1.5 workgroups
// per CU with at most 16 KiB of partial tile each: 6 MiB on 256 CUs.  Should a split not fit all the
// same (a GPU with more CUs), the launcher drops it and every workgroup sweeps the whole K: same
// result to rounding, not to the bit, and slower.
constexpr size_t NOALLOC_SLAB_BYTES = (size_t)8 << 20;                                                  // 8 MiB
constexpr size_t NOALLOC_BYTES = NOALLOC_COUNTER_BYTES + NOALLOC_A16_BYTES + NOALLOC_SLAB_BYTES;

// In host code this name is only a stand-in for the symbol, not a device address: the
// address on a given GPU comes from hipGetSymbolAddress (noalloc_setup) and is what every
// launch and memset below uses.
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

constexpr int MAX_DEVICES = 64;
static unsigned char* g_scratch_ptr[MAX_DEVICES];   // per GPU: its own copy of the array, at its own address

void noalloc_setup(int device) {
    if (device < 0 || device >= MAX_DEVICES) NVLLM_FATAL("device ordinal %d out of range", device);
    void* p = nullptr;
    HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));   // `device` is current (ensure_init)
    if (!p) NVLLM_FATAL("no device address for the static scratch array on device %d", device);
    g_scratch_ptr[device] = (unsigned char*)p;
}

}  // namespace nvllm

using namespace nvllm;

extern "C" int nvllm_matmul_noalloc(float* out, const float* in, const float* weight, int num_tokens, int in_dim,
                                    int out_dim) {
    if (!rt().inited) {
        fprintf(stderr, "nvllm_matmul_noalloc: call init_ops() first (initialisation allocates, this call must not)\n");
        return -1;
    }
    ensure_init();   // initialised already: this only binds the calling thread to the library's GPU
    unsigned char* base = g_scratch_ptr[rt().device];
    if (!base) {
        fprintf(stderr, "nvllm_matmul_noalloc: no scratch address for device %d\n", rt().device);
        return -1;
    }
    const WeightEntry* w = find_weight(weight);
    if (!w || !w->tiles) {
        fprintf(stderr, "nvllm_matmul_noalloc: `weight` is not a projection handle from load_model\n");
        return -2;
    }
    if (w->K != in_dim || w->N != out_dim) {
        fprintf(stderr, "nvllm_matmul_noalloc: handle is [%d, %d] but the call says [%d, %d]\n", w->N, w->K, out_dim, in_dim);
        return -2;
    }
    if (num_tokens < 1 || num_tokens > NOALLOC_MAX_TOKENS || in_dim > NOALLOC_MAX_IN_DIM) {
        fprintf(stderr, "nvllm_matmul_noalloc: %d tokens x %d inputs exceed the static scratch (at most %d tokens, in_dim %d)\n",
                num_tokens, in_dim, NOALLOC_MAX_TOKENS, NOALLOC_MAX_IN_DIM);
        return -3;
    }
    if (a16_halves(num_tokens, in_dim) * sizeof(__half) > NOALLOC_A16_BYTES) return -3;   // implied by the limits above
    ProfSuppress quiet;   // timing would create events
    unsigned* counters = (unsigned*)base;
    __half* a16 = (__half*)(base + NOALLOC_COUNTER_BYTES);
    float* slab = (float*)(base + NOALLOC_COUNTER_BYTES + NOALLOC_A16_BYTES);
    // the array keeps the previous call's contents; the arrival counters this launch uses (one per output
    // tile) must start at zero (launches leave them zero, but a call that died half-way would not have)
    const GemmPlan pl = gemm_plan(out_dim, in_dim, num_tokens, GemmEpilogue::F32, 0, NOALLOC_COUNTERS, NOALLOC_SLAB_BYTES);
    size_t zero_bytes = ((size_t)pl.tiles_n * pl.tiles_t * sizeof(unsigned) + 255) & ~(size_t)255;
    if (zero_bytes > NOALLOC_COUNTER_BYTES) zero_bytes = NOALLOC_COUNTER_BYTES;
    HIP_CHECK(hipMemsetAsync(counters, 0, zero_bytes, rt().stream));
    launch_pack_a16(a16, in, num_tokens, in_dim);
    GemmEpilogue e;
    e.kind = GemmEpilogue::F32;
    e.out = out;
    e.ldo = out_dim;
    const GemmWorkspace wsp{slab, NOALLOC_SLAB_BYTES, counters, NOALLOC_COUNTERS};
    launch_gemm(*w, w->tiles, out_dim, a16, num_tokens, e, 0, PC_GEMM_OTHER, &wsp);
    return 0;
}
