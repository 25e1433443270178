// ops_api.hip -- the operator entry points of include/ops.h on top of the kernel
// launchers: argument checking, host->HBM staging of the integer arrays and block
// tables, and the host-side KV block allocator.
#include <cstring>
#include <unordered_map>
#include <vector>

#include "nvllm_internal.h"
#include "step_arrays.h"

namespace nvllm {

// Pools created by init_kv_cache_manager, keyed by pool_k: element size and the
// geometry the reference passes again with every call.
struct PoolInfo {
    int elem_bytes;
    int n_layers, n_kv_heads, head_dim;
};
static std::unordered_map<const void*, PoolInfo>& pools() {
    static std::unordered_map<const void*, PoolInfo> m;
    return m;
}

KvPoolView pool_view(const KVCacheManager* mgr, int n_kv_heads, int head_dim) {
    KvPoolView v;
    v.k = mgr->pool_k;
    v.v = mgr->pool_v;
    auto it = pools().find(mgr->pool_k);
    v.elem_bytes = it == pools().end() ? 4 : it->second.elem_bytes;
    v.block_size = mgr->block_size;
    v.num_blocks = mgr->num_blocks;
    v.layer_stride = (long)mgr->num_blocks * mgr->block_size * n_kv_heads * head_dim;
    return v;
}

// Flatten what the kernels need from the host-side batch description into one
// pinned region and upload it with a single copy.
StepArrays stage_step(const int* tokens, const int* pos_arr, const int* seq_ids, int T,
                      const int* output_indices, int B, BlockTable** tables, const BlockTable* single,
                      int pos0, int block_size, int attn_ct, const int* token_src) {
    StepArrays s{};
    s.T = T;
    s.B = B;
    int n_seqs = 0;
    int max_pos = 0;
    for (int t = 0; t < T; t++) {
        const int p = pos_arr ? pos_arr[t] : pos0 + t;
        if (p > max_pos) max_pos = p;
        if (seq_ids && seq_ids[t] + 1 > n_seqs) n_seqs = seq_ids[t] + 1;
    }
    if (!seq_ids) n_seqs = 1;
    s.max_pos = max_pos;
    s.n_seqs = n_seqs;
    // blocks each sequence needs for this call = (largest position / block_size) + 1
    std::vector<int> need(n_seqs, 0);
    s.write_limit = T;
    if (tables || single) {
        for (int t = 0; t < T; t++) {
            const int p = pos_arr ? pos_arr[t] : pos0 + t;
            const int sid = seq_ids ? seq_ids[t] : 0;
            const int lb = p / block_size;
            const BlockTable* bt = single ? single : tables[sid];
            // never read the caller's table past its num_blocks entries
            const int have = lb + 1 < bt->num_blocks ? lb + 1 : bt->num_blocks;
            if (have > need[sid]) need[sid] = have;
            if (lb >= bt->num_blocks) s.table_short = true;
            if (s.write_limit == T && lb >= bt->num_blocks) {
                // reference attention.c:159-162 (and :123-126): report and stop writing
                printf("Error: Block table too small! logical_idx=%d, num_blocks=%d\n", lb, bt->num_blocks);
                s.write_limit = t;
            }
        }
    }
    size_t total_tbl = 0;
    for (int i = 0; i < n_seqs; i++) total_tbl += (size_t)need[i];
    std::vector<int> tbl_off(n_seqs, 0);
    for (int i = 1; i < n_seqs; i++) tbl_off[i] = tbl_off[i - 1] + need[i - 1];
    // attention work items: (first token, token count, last position, offset of the
    // sequence's block table) -- everything a wave needs in one 16-byte load
    std::vector<int> runs;
    s.kv_positions = 0;
    if (attn_ct > 0) {
        auto pos_of = [&](int t) { return pos_arr ? pos_arr[t] : pos0 + t; };
        auto seq_of = [&](int t) { return seq_ids ? seq_ids[t] : 0; };
        for (int t = 0; t < T;) {
            int n = 1;
            while (t + n < T && n < attn_ct && seq_of(t + n) == seq_of(t) && pos_of(t + n) == pos_of(t + n - 1) + 1) n++;
            runs.push_back(t);
            runs.push_back(n);
            runs.push_back(pos_of(t + n - 1));
            runs.push_back(tbl_off[seq_of(t)]);
            s.kv_positions += pos_of(t + n - 1) + 1;
            for (int tb = t >> 4; tb <= (t + n - 1) >> 4; tb++)
                if (tb < 4) s.runs_in_tb[tb]++;
            t += n;
        }
    }
    {   // does any sequence occur twice?
        std::vector<char> seen(n_seqs, 0);
        s.one_token_per_seq = true;
        for (int t = 0; t < T && s.one_token_per_seq; t++) {
            const int sid = seq_ids ? seq_ids[t] : 0;
            if (seen[sid]) s.one_token_per_seq = false;
            seen[sid] = 1;
        }
    }
    s.n_runs = (int)runs.size() / 4;
    // where each token's K/V row goes: resolves position -> (block table entry, slot) on
    // the host so the kernels that store K/V need one load instead of a chain of three
    std::vector<int> kv_row;
    if (tables || single) {
        kv_row.resize(T);
        for (int t = 0; t < T; t++) {
            const int p = pos_arr ? pos_arr[t] : pos0 + t;
            const BlockTable* bt = single ? single : tables[seq_ids ? seq_ids[t] : 0];
            kv_row[t] = t < s.write_limit ? bt->block_indices[p / block_size] * block_size + p % block_size : -1;
        }
    }
    const size_t n_ints = (size_t)(tokens ? T : 0) + (token_src ? T : 0) + (pos_arr ? T : 0) + (seq_ids ? T : 0) + kv_row.size() +
                          (output_indices ? B : 0) + n_seqs + total_tbl + runs.size() + 64;
    int* host = pinned_ints(n_ints);
    int* dev = (int*)ws_get(WS_STEP, n_ints * sizeof(int));
    size_t o = 0;
    auto put = [&](const int* src, size_t n) -> int* {
        if (!src || n == 0) return nullptr;
        memcpy(host + o, src, n * sizeof(int));
        int* d = dev + o;
        o += n;
        return d;
    };
    s.d_tokens = put(tokens, T);
    s.d_tok_src = put(token_src, T);
    s.d_pos = put(pos_arr, T);
    s.d_seq = put(seq_ids, T);
    s.d_out_idx = put(output_indices, B);
    s.d_kv_row = kv_row.empty() ? nullptr : put(kv_row.data(), kv_row.size());
    s.d_tbl_off = nullptr;
    s.d_tables = nullptr;
    if (tables || single) {
        s.d_tbl_off = dev + o;
        int* off_host = host + o;
        o += n_seqs;
        s.d_tables = dev + o;
        size_t run = 0;
        for (int i = 0; i < n_seqs; i++) {
            off_host[i] = tbl_off[i];
            if (need[i] > 0) {
                const BlockTable* bt = single ? single : tables[i];
                memcpy(host + o + run, bt->block_indices, (size_t)need[i] * sizeof(int));
                run += (size_t)need[i];
            }
        }
        o += run;
    }
    o = (o + 3) & ~(size_t)3;  // int4 alignment
    s.d_runs = runs.empty() ? nullptr : put(runs.data(), runs.size());
    if (o > 0) {
        static const int by_kernel = getenv("NVLLM_STAGE_COPY") ? atoi(getenv("NVLLM_STAGE_COPY")) == 0 : 1;
        if (by_kernel && o <= (1u << 16)) launch_copy_ints(dev, host, o);   // small: a kernel reads the pinned region itself
        else HIP_CHECK(hipMemcpyAsync(dev, host, o * sizeof(int), hipMemcpyHostToDevice, rt().stream));
    }
    return s;
}

}  // namespace nvllm

using namespace nvllm;

extern "C" {

// ------------------------------------------------------------- dense ops
void rms_norm(float* o, float* x, float* weight, int size, int num_tokens, float eps) {
    ensure_init();
    launch_rmsnorm_f32(o, x, weight, size, num_tokens, eps);
}

void matmul(float* out, float* in, float* weight, int num_tokens, int in_dim, int out_dim) {
    ensure_init();
    const WeightEntry* w = find_weight(weight);
    if (!w) {  // a plain fp32 [out_dim, in_dim] device buffer
        launch_matmul_f32w(out, in, weight, num_tokens, in_dim, out_dim);
        return;
    }
    if (w->rowmajor && !w->tiles) NVLLM_FATAL("matmul: the embedding table handle is not a projection weight");
    if (w->K != in_dim || w->N != out_dim)
        NVLLM_FATAL("matmul: handle is [%d, %d] but the call says [%d, %d]", w->N, w->K, out_dim, in_dim);
    __half* a16 = (__half*)ws_get(WS_MISC, a16_halves(num_tokens, in_dim) * sizeof(__half));
    launch_pack_a16(a16, in, num_tokens, in_dim);
    GemmEpilogue e;
    e.kind = GemmEpilogue::F32;
    e.out = out;
    e.ldo = out_dim;
    launch_gemm(*w, w->tiles, out_dim, a16, num_tokens, e, 0, PC_GEMM_OTHER);
}

void apply_rope(float* q, float* k, int pos, float theta, int head_dim, int num_tokens, int n_heads,
                int n_kv_heads) {
    ensure_init();
    if (num_tokens <= 0) return;
    const float2* tab = rope_table(theta, head_dim, pos + num_tokens);
    launch_rope(q, k, nullptr, pos, tab, num_tokens, n_heads, n_kv_heads, head_dim);
}

void apply_rope_batch(float* q, float* k, int* pos_arr, float theta, int head_dim, int num_tokens,
                      int n_heads, int n_kv_heads) {
    ensure_init();
    if (num_tokens <= 0) return;
    StepArrays s = stage_step(nullptr, pos_arr, nullptr, num_tokens, nullptr, 0, nullptr, nullptr, 0, 1);
    const float2* tab = rope_table(theta, head_dim, s.max_pos + 1);
    launch_rope(q, k, s.d_pos, 0, tab, num_tokens, n_heads, n_kv_heads, head_dim);
}

void swiglu(float* hb, float* h_gate, float* h_up, int hidden_dim, int num_tokens) {
    ensure_init();
    launch_swiglu(hb, h_gate, h_up, (long)hidden_dim * num_tokens);
}

void softmax(float* x, int size) {
    ensure_init();
    launch_softmax(x, size);
}

void accum(float* a, float* b, int size) {
    ensure_init();
    launch_accum(a, b, size);
}

// ------------------------------------------------------- KV block pool
// Host bookkeeping exactly as reference memory.c:8-82; the two pools are HBM.
void init_kv_cache_manager(KVCacheManager* mgr, int block_size, int num_blocks, int n_layers,
                           int n_kv_heads, int head_dim) {
    ensure_init();
    // tensor parallelism: the caller passes the model's head count (it sizes the pool from Config);
    // this process stores and reads the K/V of its own heads only
    if (tp().active && tp().world > 1) {
        if (n_kv_heads % tp().world) NVLLM_FATAL("%d kv heads do not divide over %d tensor-parallel ranks", n_kv_heads, tp().world);
        n_kv_heads /= tp().world;
    }
    mgr->block_size = block_size;
    mgr->num_blocks = num_blocks;
    mgr->free_blocks_count = num_blocks;
    mgr->free_block_indices = (int*)malloc((size_t)num_blocks * sizeof(int));
    for (int slot = 0; slot < num_blocks; slot++) mgr->free_block_indices[slot] = num_blocks - 1 - slot;
    const size_t elems = (size_t)n_layers * num_blocks * block_size * n_kv_heads * head_dim;
    const int eb = rt().kv_elem_bytes;
    if (eb == 2 && (block_size != KV16_PAGE_KEYS || head_dim % 16 != 0))
        NVLLM_FATAL("an fp16 KV pool stores 16-position pages: block_size must be 16 (got %d) and head_dim a "
                    "multiple of 16 (got %d)", block_size, head_dim);
    void *pk = nullptr, *pv = nullptr;
    if (device_malloc(&pk, elems * eb) != 0 || device_malloc(&pv, elems * eb) != 0)
        NVLLM_FATAL("cannot allocate the KV pool (2 x %zu bytes)", elems * eb);
    // fresh pool reads as zeros, so attention over never-written rows is at least defined
    HIP_CHECK(hipMemsetAsync(pk, 0, elems * eb, rt().stream));
    HIP_CHECK(hipMemsetAsync(pv, 0, elems * eb, rt().stream));
    mgr->pool_k = (float*)pk;
    mgr->pool_v = (float*)pv;
    pools()[pk] = PoolInfo{eb, n_layers, n_kv_heads, head_dim};
}

void free_kv_cache_manager(KVCacheManager* mgr) {
    if (mgr->free_block_indices) free(mgr->free_block_indices);
    mgr->free_block_indices = nullptr;
    if (mgr->pool_k) {
        pools().erase(mgr->pool_k);
        device_free(mgr->pool_k);
    }
    if (mgr->pool_v) device_free(mgr->pool_v);
    mgr->pool_k = mgr->pool_v = nullptr;
}

int alloc_block(KVCacheManager* mgr) {
    if (mgr->free_blocks_count <= 0) return -1;
    mgr->free_blocks_count -= 1;
    return mgr->free_block_indices[mgr->free_blocks_count];
}

void free_block(KVCacheManager* mgr, int block_idx) {
    if (mgr->free_blocks_count >= mgr->num_blocks) return;
    mgr->free_block_indices[mgr->free_blocks_count] = block_idx;
    mgr->free_blocks_count += 1;
}

long get_physical_offset(KVCacheManager* mgr, int layer, int block_idx, int block_offset, int n_kv_heads,
                         int head_dim) {
    const long row = (long)n_kv_heads * head_dim;
    const long blk = (long)mgr->block_size * row;
    const long lay = (long)mgr->num_blocks * blk;
    return (long)layer * lay + (long)block_idx * blk + (long)block_offset * row;
}

// --------------------------------------------------------------- KV writes
static KvPoolView contiguous_view(float* kc, float* vc, int max_seq_len, int n_kv_heads, int head_dim) {
    KvPoolView v;
    v.k = kc;
    v.v = vc;
    v.elem_bytes = 4;
    v.block_size = 1;
    v.num_blocks = max_seq_len;
    v.layer_stride = (long)max_seq_len * n_kv_heads * head_dim;
    return v;
}

void update_kv_cache(float* key_cache, float* value_cache, float* k, float* v, int layer, int pos,
                     int max_seq_len, int n_kv_heads, int head_dim, int num_tokens) {
    ensure_init();
    const KvPoolView pv = contiguous_view(key_cache, value_cache, max_seq_len, n_kv_heads, head_dim);
    launch_kv_write(k, v, nullptr, pos, nullptr, nullptr, nullptr, pv, layer, num_tokens, n_kv_heads, head_dim);
}

void update_kv_cache_paged(KVCacheManager* mgr, BlockTable* block_table, float* k, float* v, int layer,
                           int pos, int n_kv_heads, int head_dim, int num_tokens) {
    ensure_init();
    if (num_tokens <= 0) return;
    StepArrays s = stage_step(nullptr, nullptr, nullptr, num_tokens, nullptr, 0, nullptr, block_table, pos,
                              mgr->block_size);
    launch_kv_write(k, v, nullptr, pos, nullptr, s.d_tbl_off, s.d_tables, pool_view(mgr, n_kv_heads, head_dim),
                    layer, s.write_limit, n_kv_heads, head_dim);
}

void update_kv_cache_paged_batch(KVCacheManager* mgr, BlockTable** block_tables, int* seq_ids, float* k,
                                 float* v, int layer, int* pos_arr, int n_kv_heads, int head_dim,
                                 int num_tokens) {
    ensure_init();
    if (num_tokens <= 0) return;
    StepArrays s = stage_step(nullptr, pos_arr, seq_ids, num_tokens, nullptr, 0, block_tables, nullptr, 0,
                              mgr->block_size);
    launch_kv_write(k, v, s.d_pos, 0, s.d_seq, s.d_tbl_off, s.d_tables, pool_view(mgr, n_kv_heads, head_dim),
                    layer, s.write_limit, n_kv_heads, head_dim);
}

// --------------------------------------------------------------- attention
void multi_head_attention(float* out, float* q, float* key_cache, float* value_cache, float* att, int layer,
                          int pos, int max_seq_len, int n_heads, int n_kv_heads, int head_dim,
                          int num_tokens) {
    (void)att;
    ensure_init();
    if (num_tokens <= 0) return;
    StepArrays s = stage_step(nullptr, nullptr, nullptr, num_tokens, nullptr, 0, nullptr, nullptr, pos, 16,
                              attn_tokens_per_item(n_heads, n_kv_heads));
    AttnArgs a{};
    a.q = q;
    a.out_f32 = out;
    a.pool = contiguous_view(key_cache, value_cache, max_seq_len, n_kv_heads, head_dim);
    a.layer = layer;
    a.T = num_tokens;
    a.H = n_heads;
    a.KVH = n_kv_heads;
    a.hd = head_dim;
    a.max_pos = s.max_pos;
    a.runs = s.d_runs;
    a.n_runs = s.n_runs;
    a.kv_positions = s.kv_positions;
    launch_paged_attention(a);
}

void paged_attention(float* out, float* q, KVCacheManager* mgr, BlockTable* block_table, float* att,
                     int layer, int pos, int max_seq_len, int n_heads, int n_kv_heads, int head_dim,
                     int num_tokens) {
    (void)att;
    (void)max_seq_len;
    ensure_init();
    if (num_tokens <= 0) return;
    StepArrays s = stage_step(nullptr, nullptr, nullptr, num_tokens, nullptr, 0, nullptr, block_table, pos,
                              mgr->block_size, attn_tokens_per_item(n_heads, n_kv_heads));
    // the reference reads block ids past the table here (attention.c:285 has no bounds check);
    // on the GPU that would be a wild read of the pool, so it ends the process instead
    if (s.table_short) NVLLM_FATAL("paged attention: a position lies beyond its sequence's block table");
    AttnArgs a{};
    a.q = q;
    a.out_f32 = out;
    a.pool = pool_view(mgr, n_kv_heads, head_dim);
    a.layer = layer;
    a.tables = s.d_tables;
    a.T = num_tokens;
    a.H = n_heads;
    a.KVH = n_kv_heads;
    a.hd = head_dim;
    a.max_pos = s.max_pos;
    a.runs = s.d_runs;
    a.n_runs = s.n_runs;
    a.kv_positions = s.kv_positions;
    launch_paged_attention(a);
}

void paged_attention_batch(float* out, float* q, KVCacheManager* mgr, BlockTable** block_tables, int* seq_ids,
                           float* att, int layer, int* pos_arr, int max_seq_len, int n_heads, int n_kv_heads,
                           int head_dim, int num_tokens) {
    (void)att;
    (void)max_seq_len;
    ensure_init();
    if (num_tokens <= 0) return;
    StepArrays s = stage_step(nullptr, pos_arr, seq_ids, num_tokens, nullptr, 0, block_tables, nullptr, 0,
                              mgr->block_size, attn_tokens_per_item(n_heads, n_kv_heads));
    // the reference reads block ids past the table here (attention.c:285 has no bounds check);
    // on the GPU that would be a wild read of the pool, so it ends the process instead
    if (s.table_short) NVLLM_FATAL("paged attention: a position lies beyond its sequence's block table");
    AttnArgs a{};
    a.q = q;
    a.out_f32 = out;
    a.pool = pool_view(mgr, n_kv_heads, head_dim);
    a.layer = layer;
    a.tables = s.d_tables;
    a.T = num_tokens;
    a.H = n_heads;
    a.KVH = n_kv_heads;
    a.hd = head_dim;
    a.max_pos = s.max_pos;
    a.runs = s.d_runs;
    a.n_runs = s.n_runs;
    a.kv_positions = s.kv_positions;
    launch_paged_attention(a);
}

}  // extern "C"
