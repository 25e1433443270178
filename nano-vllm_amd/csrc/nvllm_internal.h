// nvllm_internal.h -- declarations shared by the translation units of
// libnvllm_hip.so.  Nothing here is part of the public ABI (see include/*.h).
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

extern "C" {
#include "backend.h"
#include "model.h"
#include "nvllm.h"
#include "ops.h"
#include "sampler.h"
#include "structs.h"
}

#define HIP_CHECK(expr)                                                                       \
    do {                                                                                      \
        hipError_t err__ = (expr);                                                            \
        if (err__ != hipSuccess) {                                                            \
            fprintf(stderr, "nvllm: HIP error %d (%s) at %s:%d: %s\n", (int)err__,            \
                    hipGetErrorString(err__), __FILE__, __LINE__, #expr);                     \
            abort();                                                                          \
        }                                                                                     \
    } while (0)

#define NVLLM_FATAL(...)                                   \
    do {                                                   \
        fprintf(stderr, "nvllm: " __VA_ARGS__);            \
        fprintf(stderr, " (%s:%d)\n", __FILE__, __LINE__); \
        abort();                                           \
    } while (0)

namespace nvllm {

// ------------------------------------------------------------------ layouts
// "Tile order" (used for fp16 weight matrices and for fp16 activations): a
// [rows, K] matrix is cut into tiles of 16 rows x 32 k.  One tile is 512 halves
// (1 KiB) stored in the register order of one v_mfma_f32_16x16x32_f16 operand:
//   element (r, k) of the tile sits at half index  lane*8 + (k & 7),
//   lane = (r & 15) + 16 * ((k >> 3) & 3),
// so lane l of a wave reads its whole operand fragment with one 16-byte load and a
// wave reads the tile as 1 KiB of consecutive bytes.  Tiles of one 16-row block
// are consecutive in k:  tile(rb, kb) = rb * Kb + kb,  Kb = ceil(K / 32).
//
// Activations additionally carry two planes per tile, "hi" then "lo":
//   x = float(hi) + float(lo) + O(2^-22 |x|),  hi = fp16(x), lo = fp16(x - hi),
// so that two fp16 MFMAs reproduce an fp32 x fp16 product to fp32 accuracy.
//   a16 tile(tb, kb, plane) = ((tb * Kb + kb) * 2 + plane) * 512 halves.
constexpr int TILE_HALVES = 512;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t a16_halves(int tokens, int K) {
    return (size_t)cdiv(tokens, 16) * cdiv(K, 32) * 2 * TILE_HALVES;
}

// A weight matrix registered by load_model.  `handle` is what the public structs
// carry; matmul() looks it up.
struct WeightEntry {
    const __half* tiles;  // first tile of n-block 0
    int N, K;             // logical shape [N, K]
    int Kb;               // k-tiles per n-block
    long nb_stride;       // halves between consecutive n-blocks of THIS matrix
    const __half* rowmajor; // non-null for the embedding table ([N, K] fp16 row-major)
};
const WeightEntry* find_weight(const void* handle);
void register_weight(const void* handle, const WeightEntry& e);
void unregister_weight(const void* handle);

// ------------------------------------------------------------------ runtime
struct Runtime {
    bool inited = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // result copies of a pass when logits rows travel to the host (created on first use)
    int logits_mode = NVLLM_LOGITS_HOST;
    int kv_elem_bytes = 4;
    bool kv_elem_set = false;   // nvllm_set_kv_elem_bytes was called: the environment default does not apply
    int num_cus = 256;
};
Runtime& rt();
void ensure_init();
void sync_streams();   // the library stream and, if it exists, the copy stream
// noalloc.hip: resolves the device address of the library's static scratch array for the current
// device (called once by ensure_init, after hipSetDevice)
void noalloc_setup(int device);

// Tensor parallelism (tp.hip).  `active` selects the forward pass that leaves the two
// residual projections as partial sums and all-reduces them.
struct TpState {
    int rank = 0, world = 1;
    bool active = false;
    nvllm_allreduce_fn fn = nullptr;   // caller-supplied exchange; null = RCCL
    void* ctx = nullptr;
};
TpState& tp();
void tp_allreduce(float* buf, size_t count);   // in place, on the library stream
// recv[r * count .. ) = rank r's `count` floats at send, on every rank (send must not overlap recv)
void tp_allgather(const float* send, float* recv, size_t count);
// tp_peer.hip: the all-reduce of [T, D] partial sums with the residual tail (x += sum; a16 = x * gain tiles; ss) in its
// last phases; false = not available (no peer session of several ranks): all-reduce, then launch_resid_prescale
bool tp_peer_allreduce_resid(float* partial, float* x, __half* a16, float* ss, const float* gain, int T, int D);
void tp_peer_finalize();   // tp_peer.hip: unmaps the other ranks' exchange segments (no-op without a peer session)

// Grow-only device scratch, one buffer per slot.
enum WsSlot {
    WS_A16_X = 0,   // normalised activations feeding qkv / gate-up / classifier
    WS_A16_ATT,     // attention output feeding wo
    WS_A16_H,       // swiglu output feeding w_down
    WS_SLAB,        // split-K partial sums
    WS_ATT_PART,    // attention split partials (acc, m, l)
    WS_LOGITS,      // device logits [B, V]
    WS_ARGMAX,      // device argmax [B]
    WS_STEP,        // per-call integer arrays (tokens, pos, seq ids, tables ...)
    WS_MISC,
    WS_MISC2,
    WS_SAMPLE,      // a logits row handed to sample() that is not part of the last forward pass ...
    WS_SAMPLE_PROBS1,   // ... its probabilities and nucleus candidates
    WS_SAMPLE_CAND1,
    WS_SAMPLE_ROWS,     // logits mode HOST: all rows of a pass, uploaded again for sample() with a temperature
    WS_SAMPLE_PROBS,
    WS_SAMPLE_CAND,
    WS_TP_LOCAL,    // tensor parallelism: this rank's slice of the logits / its arg-max pairs
    WS_TP_GATHER,   // ... and what the ranks gathered
    WS_CHAIN_SYNC,  // arrival counters of the one-launch decode step
    WS_AMAX_PAIRS,  // the classifier tiles' (maximum, index) pairs
    WS_COUNT
};
void* ws_get(WsSlot slot, size_t bytes);

// Pinned host staging for the integer arrays of one call.
int* pinned_ints(size_t n);   // ring allocation, for callers that fill it themselves

// RoPE angle table for (theta, head_dim), covering positions [0, min_positions):
// float2 {cos, sin} per (position, pair), computed on the host with the same libm
// calls as the reference (src/kernels/layers.c:119-122) and cached in HBM.
const float2* rope_table(float theta, int head_dim, int min_positions);

// ---------------------------------------------------------------- profiling
enum ProfClass {
    PC_EMBED = 0,
    PC_RMSNORM,
    PC_GEMM_QKV,
    PC_GEMM_WO,
    PC_GEMM_GATEUP,
    PC_GEMM_DOWN,
    PC_GEMM_HEAD,
    PC_GEMM_OTHER,
    PC_ROPE_KV,
    PC_ATTN,
    PC_ATTN_COMBINE,
    PC_ELEMWISE,
    PC_ARGMAX,
    PC_PACK,
    PC_CHAIN,      // all layers of a decode step in one launch (chain.hip)
    PC_COUNT
};
hipEvent_t prof_start_event();   // events of the enclosing ProfScope, or null when timing is off
hipEvent_t prof_stop_event();
// Every kernel goes through this: on the library stream, with the timing events of the
// enclosing ProfScope attached to the dispatch itself.
// A launch the runtime rejects (grid, LDS size, missing code object) is reported here, not at some
// later unrelated call.
#define NVLLM_LAUNCH(kernel, grid, block, shmem, ...)                                                        \
    do {                                                                                                     \
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, ::nvllm::rt().stream, ::nvllm::prof_start_event(), \
                              ::nvllm::prof_stop_event(), 0, __VA_ARGS__);                                   \
        HIP_CHECK(hipGetLastError());                                                                        \
    } while (0)
// While one exists, ProfScope records nothing (and so creates no events): for entry points that
// must not allocate.
struct ProfSuppress {
    ProfSuppress();
    ~ProfSuppress();
};
struct ProfScope {  // names the class of, and attaches timing events to, the ONE launch made inside it
    ProfScope(int cls, double bytes, double flops);
    ~ProfScope();
    int slot;
};

// ----------------------------------------------------- kernel launchers (device ptrs)
// What a projection does with its complete sums (see gemm.hip).
struct GemmEpilogue {
    enum Kind { F32 = 0, SWIGLU_A16 = 1, RESID_A16 = 2, ROPE_KV = 3 } kind = F32;
    // Input normalisation.  When ss_in is set the activations fed to the projection are
    // x * gain (not yet divided by the RMS) and the result is scaled per token by
    // 1/sqrt(sum_j ss_in[t, j] / d_in + eps); ss_gather (optional) maps the projection's
    // token index to the row of ss_in.
    const float* ss_in = nullptr;
    const int* ss_gather = nullptr;
    int ss_tiles = 0;
    int d_in = 1;
    float eps = 0.f;
    // F32: out[t, ldo]
    float* out = nullptr;
    int ldo = 0;
    // F32, optional: also the first index of each row's maximum over all N outputs (the sampler's greedy
    // rule, reference sampler.c:86-97) -> amax_out[t].  Every tile leaves one (maximum, index) pair per
    // token in amax_pairs[tile][amax_ld] (room for cdiv(N, 32) tiles), a small launch merges them.
    int* amax_out = nullptr;
    int* amax_out_host = nullptr;   // optional: the indices also go straight to this page-locked host array (no copy follows)
    float2* amax_pairs = nullptr;
    int amax_ld = 0;
    // SWIGLU_A16 / RESID_A16: fp16 hi/lo tiles for the next projection
    __half* a16_out = nullptr;
    int Kb_out = 0;
    // RESID_A16: x[t, n] += result; a16_out = x * gain_next; ss_out[t, tile] = sum over the tile of x^2
    float* x = nullptr;
    const float* gain_next = nullptr;
    float* ss_out = nullptr;
    // ROPE_KV: rows [0,QD) = q, [QD,QD+KD) = k, rest = v
    float* q_out = nullptr;
    int QD = 0, KD = 0, hd = 0;
    const int* pos = nullptr;
    const int* kv_row = nullptr;   // per token: pool row (block * bs + slot) for its K/V, -1 = do not store
    const float2* rope = nullptr;
    void* pool_k = nullptr;
    void* pool_v = nullptr;
    int kv_elem_bytes = 4;
    int bs = 16;
    long layer_off = 0;
};
// Memory a projection launch needs beyond its operands, when the caller brings it instead of the
// library's grow-only buffers: a slab for split-K partial tiles and zeroed arrival counters (every
// launch leaves them zero again).  A slab too small for the chosen K split means no K split.
struct GemmWorkspace {
    float* slab;
    size_t slab_bytes;
    unsigned* counters;
    int n_counters;
};
// out = A16[T, K] x W[N, K]^T followed by the epilogue.  Returns the K-split used.  With `wsp` the
// launch allocates nothing (decode-sized batches only: the dense prefill kernel is not taken).
int launch_gemm(const WeightEntry& w, const __half* tiles, int N, const __half* a16, int T,
                const GemmEpilogue& epi, int force_ksplit, int prof_cls, const GemmWorkspace* wsp = nullptr);
int gemm_pick_ksplit(int N, int K, int T);
struct GemmPlan {
    int nbw, TB, nw;         // 16-row blocks per tile, 16-token blocks per group, waves per workgroup
    int ks, kb_per_split;    // K slices and k-tiles per slice
    int tiles_n, tiles_t;    // grid: feature tiles (padding included) x K slices x token groups
};
GemmPlan gemm_plan(int N, int K, int T, int epi_kind, int force_ksplit, long max_counters, size_t slab_cap);
int gemm_feature_tiles(int N);   // 32-feature groups of a projection output = columns of ss_out
void gemm_release();

unsigned* gemm_tile_counters();   // the split-K arrival counters (allocated on first use, zero between launches)

// ---- chain.hip: the layers of one decode step as one launch
struct ChainLayer {           // per layer, in HBM (built by load_model)
    const __half* qkv;
    const __half* wo;
    const __half* gateup;
    const __half* down;
    const float* gain_ffn;    // the gain that follows the output projection
    const float* gain_next;   // the gain that follows the down projection (next layer's, or the final norm's)
};
struct ChainStep {
    const ChainLayer* layers;
    int n_layers;
    int T, D, F, H, KVH, hd;
    float eps;
    float* x;
    __half* a16_x;
    __half* a16_att;
    __half* a16_h;
    float* ss_a;
    float* ss_b;
    float* q;
    const int* pos;
    const int* kv_row;
    const float2* rope;
    void* pool_k;
    void* pool_v;
    int kv_elem_bytes, block_size;
    long layer_stride;
    const int* tables;
    const int* runs;
    int n_runs;
    int runs_in_tb[4];        // attention work items' runs that touch each 16-token block
    bool one_token_per_seq;   // every sequence of the step contributes exactly one token (pure decode)
    unsigned* sync;           // chain_sync_words() words, zero when the launch starts
    unsigned* status;         // one word, zero when the launch starts (see chain.hip)
    double bytes, flops;      // algorithmic bytes / flops of the launch, for the kernel timing classes
};
int chain_enabled();
size_t chain_sync_words(int n_layers);
bool launch_decode_chain(const ChainStep& s);

void launch_pack_a16(__half* a16, const float* in, int T, int K);
void launch_matmul_f32w(float* out, const float* in, const float* w, int T, int K, int N);

void launch_rmsnorm_f32(float* o, const float* x, const float* gain, int D, int T, float eps);

struct KvPoolView {
    void* k;
    void* v;
    int elem_bytes;    // 4 or 2
    int block_size;
    int num_blocks;
    long layer_stride; // elements
};

// ---- fp16 pool: "page order".  The fp32 pool keeps the reference's row layout
// [layer][block][position][kv head][dim] (memory.c:37-43).  The fp16 pool keeps the
// same [layer][block] addressing (block ids, block tables and the allocator are
// untouched) but inside a block it stores one PAGE per kv head -- the block's 16
// positions x hd dims of that head -- with the elements in the register order of
// a v_mfma_f32_16x16x16_f16 operand, so that the attention kernel reads a page as
// hd/16 fully contiguous 512-byte wave loads per pool and feeds them to the matrix
// cores without conversion or shuffles:
//   K page (A operand, rows = keys):  (key, d) -> (d>>4)*256 + (((d>>2)&3)*16 + key)*4 + (d&3)
//   V page (B operand, k = keys):     (key, d) -> (d>>4)*256 + ((key>>2)*16 + (d&15))*4 + (key&3)
// Values are saturated to the finite fp16 range when written and fresh pools are
// zero-filled, so every half in the pool is finite (masked keys rely on 0 * finite = 0).
constexpr int KV16_PAGE_KEYS = 16;
__host__ __device__ __forceinline__ long kv16_page(long block, int kvh, int KVH, int hd) {
    return (block * KVH + kvh) * (long)(KV16_PAGE_KEYS * hd);
}
__host__ __device__ __forceinline__ int kv16_k_off(int key, int d) {
    return (d >> 4) * 256 + ((((d >> 2) & 3) * 16 + key) << 2) + (d & 3);
}
__host__ __device__ __forceinline__ int kv16_v_off(int key, int d) {
    return (d >> 4) * 256 + (((key >> 2) * 16 + (d & 15)) << 2) + (key & 3);
}
void launch_rope(float* q, float* k, const int* pos, int pos0, const float2* rope, int T, int H,
                 int KVH, int hd);
void launch_kv_write(const float* k, const float* v, const int* pos, int pos0, const int* seq,
                     const int* tbl_off, const int* tables, const KvPoolView& pool, int layer, int T,
                     int KVH, int hd);

struct AttnArgs {
    const float* q;        // [T, H*hd]
    float* out_f32;        // [T, H*hd] or null
    __half* out_a16;       // tiles of [T, H*hd] or null
    KvPoolView pool;
    int layer;
    const int* tables;     // device, concatenated block tables (null: contiguous cache)
    int T, H, KVH, hd;
    int max_pos;           // largest position in the batch (host knowledge, for split sizing)
    const int* runs;       // device int4 (first token, count, last position, table offset): runs of consecutive
    int n_runs;            //   positions of one sequence, at most attn_tokens_per_item() tokens each
    long kv_positions;     // sum over runs of (last position + 1): rows of K/V each kv head reads
};
void launch_paged_attention(const AttnArgs& a);
int attn_tokens_per_item(int H, int KVH);

// x[t,:] = table[tokens[t],:]; a16 = x * gain in hi/lo tiles; ss[t, j] = sum of x^2 over features 32j..32j+31
// tok_src / prev (optional): where tok_src[t] >= 0 the token id is prev[tok_src[t]] instead of tokens[t]
// zero / n_zero (optional): the kernel also clears word 0 of n_zero consecutive 128-byte lines at `zero`, and
// *zero_word (state of the launch that follows it)
void launch_embed_prescale(float* x, __half* a16, float* ss, const __half* table, const int* tokens,
                           const float* gain, int T, int D, const int* tok_src = nullptr, const int* prev = nullptr,
                           unsigned* zero = nullptr, int n_zero = 0, unsigned* zero_word = nullptr);
// x[t,:] += y[t,:]; a16 = x * gain in hi/lo tiles; ss as above (the tail of a residual projection
// whose sums were completed outside the projection kernel)
void launch_resid_prescale(float* x, const float* y, __half* a16, float* ss, const float* gain, int T, int D);
// rows idx[r] of an a16 activation (and of its ss table) compacted to rows r = 0..B-1
void launch_gather_a16(__half* dst, float* dst_ss, const __half* src, const float* src_ss, const int* idx,
                       int B, int K, int ss_tiles);
void launch_copy_ints(int* dst, const int* host_src, size_t n);   // host_src: page-locked, 16-byte aligned, padded to 4 ints
void launch_patch_tokens(int* tokens, const int* src, const int* prev, int T);
void launch_gather_rows_f32(float* dst, const float* src, const int* idx, int B, int D);
void launch_swiglu(float* out, const float* g, const float* u, long n);
void launch_accum(float* a, const float* b, long n);
void launch_softmax(float* x, int n);
void launch_argmax(int* out, const float* logits, int B, int V);
// tensor-parallel classifier: per-rank (max, global index) pairs, their merge, and the regrouping of
// gathered vocabulary slices [rank][row][Vl] into logits rows [row][V]
void launch_argmax_pairs(float2* pairs, const float* logits, int B, int n_valid, int ld, int index_base);
void launch_argmax_merge(int* out, const float2* pairs, int W, int B);
void launch_argmax_ld(int* out, const float* logits, int B, int V, int ld, int* out_host = nullptr);   // rows `ld` floats apart; out_host: also to that page-locked host array
void launch_argmax_tiles(int* out, const float2* pairs, int n_tiles, int B, int ld, int* out_host = nullptr);
void launch_logits_unshard(float* full, const float* parts, int W, int B, int Vl, int V);
// probs = softmax(logits / temperature) per row; cutoff >= 0 also compacts the entries >= cutoff, in
// index order, into cand[row][0..cap) and counts them (count may exceed cap: the row did not fit)
void launch_sample_prep(float* probs, const float* logits, int B, int V, float temperature, float cutoff,
                        ProbIndex* cand, int cap, int* count);
void launch_tile_weights(__half* tiles, const __half* rowmajor, int N, int K, long nb_stride);
void launch_f32_to_f16(__half* dst, const float* src, size_t n);
void launch_f16_to_f32(float* dst, const __half* src, size_t n);

}  // namespace nvllm
