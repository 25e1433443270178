/*
 * ops.h -- operator interface of the forward pass (C linkage).
 *
 * Same entry points, argument order and meaning as reference include/ops.h:10-75;
 * each declaration cites the CPU implementation it replaces.  Conventions:
 *   - `float*` tensor arguments are HBM pointers from device_malloc() (fp32,
 *     row-major), except `weight` of matmul, which may also be an fp16 handle
 *     produced by load_model;
 *   - `int* pos_arr`, `int* seq_ids`, `BlockTable*`, `BlockTable**`,
 *     `KVCacheManager*` are host memory, read during the call;
 *   - every operator is enqueued on the library stream and may return before the
 *     kernels have run; device_memcpy(DEVICE_TO_HOST)/device_sync() order after
 *     them.
 */
#ifndef OPS_H
#define OPS_H

#include "structs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Create / destroy the library's stream, staging buffers and RoPE tables.
 * Reference: src/kernels/layers.c:11-17 (no-ops on CPU). */
void init_ops(void);
void free_ops(void);

/* o[t,:] = x[t,:] * (1/sqrtf(mean(x[t,:]^2) + eps)) * weight[:]; o may alias x.
 *   o, x    [num_tokens, size] fp32 in HBM          weight  [size] fp32 in HBM
 * The sum of squares is taken in fp32 per row; the scaling is (x * r) * w in that
 * order, as the reference rounds it.
 * Reference: src/kernels/layers.c:23-38. */
void rms_norm(float* o, float* x, float* weight, int size, int num_tokens, float eps);

/* out[t,i] = sum_j in[t,j] * weight[i,j]; weight is [out_dim, in_dim].
 *   out     [num_tokens, out_dim] fp32 in HBM       in  [num_tokens, in_dim] fp32 in HBM
 *   weight  either a plain fp32 [out_dim, in_dim] HBM buffer, or one of the pointers
 *           load_model() stored in Weights / LayerWeights (fp16 tiles in HBM; the call
 *           checks that in_dim / out_dim match the tensor behind the handle)
 * fp32 accumulation on the matrix cores; the fp32 input enters as an fp16 pair
 * (value, remainder), so products are formed as the reference forms them from fp16
 * weights expanded to fp32, up to the order of the sum.
 * Reference: src/kernels/layers.c:44-67. */
void matmul(float* out, float* in, float* weight, int num_tokens, int in_dim, int out_dim);

/* Interleaved-pair rotary embedding on q (n_heads) and k (n_kv_heads), in place;
 * token t sits at position pos + t.
 *   q  [num_tokens, n_heads * head_dim]   k  [num_tokens, n_kv_heads * head_dim]   (fp32, HBM)
 * Angles come from a table computed on the host with the libm calls the reference
 * makes (powf, cosf, sinf), so the rotation is bit-identical to it.
 * Reference: src/kernels/layers.c:73-107. */
void apply_rope(float* q, float* k, int pos, float theta, int head_dim, int num_tokens,
                int n_heads, int n_kv_heads);

/* Same with one position per token (ragged batch).
 * Reference: src/kernels/layers.c:109-143. */
void apply_rope_batch(float* q, float* k, int* pos_arr, float theta, int head_dim,
                      int num_tokens, int n_heads, int n_kv_heads);

/* hb = silu(h_gate) * h_up, element-wise; hb may alias h_gate.
 *   hb, h_gate, h_up  [num_tokens, hidden_dim] fp32 in HBM;  silu(g) = g / (1 + expf(-g))
 * Reference: src/kernels/layers.c:149-161. */
void swiglu(float* hb, float* h_gate, float* h_up, int hidden_dim, int num_tokens);

/* In-place softmax of one vector.  Reference: src/kernels/layers.c:170-186. */
void softmax(float* x, int size);

/* a[i] += b[i].  Reference: src/kernels/layers.c:163-168. */
void accum(float* a, float* b, int size);

/* Contiguous (non-paged) KV cache write and attention.
 * Reference: src/kernels/attention.c:11-30 and :36-103 (no live caller). */
void update_kv_cache(float* key_cache, float* value_cache, float* k, float* v, int layer,
                     int pos, int max_seq_len, int n_kv_heads, int head_dim, int num_tokens);
void multi_head_attention(float* out, float* q, float* key_cache, float* value_cache,
                          float* att, int layer, int pos, int max_seq_len, int n_heads,
                          int n_kv_heads, int head_dim, int num_tokens);

/* KV block pool: host-side LIFO free stack + two HBM pools laid out
 * [n_layers, num_blocks, block_size, n_kv_heads, head_dim].
 * Reference: src/memory.c:8-44, :47-51, :54-61, :64-70, :74-82. */
void init_kv_cache_manager(KVCacheManager* mgr, int block_size, int num_blocks, int n_layers,
                           int n_kv_heads, int head_dim);
void free_kv_cache_manager(KVCacheManager* mgr);
int  alloc_block(KVCacheManager* mgr);              /* -1 when the pool is empty */
void free_block(KVCacheManager* mgr, int block_idx);
long get_physical_offset(KVCacheManager* mgr, int layer, int block_idx, int block_offset,
                         int n_kv_heads, int head_dim);

/* Scatter K/V rows of a contiguous run of positions of one sequence into the pool.
 * Reference: src/kernels/attention.c:109-139. */
void update_kv_cache_paged(KVCacheManager* mgr, BlockTable* block_table, float* k, float* v,
                           int layer, int pos, int n_kv_heads, int head_dim, int num_tokens);

/* Scatter K/V rows of a ragged batch; token t belongs to block_tables[seq_ids[t]]
 * at position pos_arr[t]: its row lands in physical block
 * block_indices[pos / block_size], slot pos % block_size, of layer `layer`.
 *   k, v  [num_tokens, n_kv_heads * head_dim] fp32 in HBM  A token whose logical block is >= table->num_blocks
 * stops the call with the reference's message (attention.c:159-162).
 * Reference: src/kernels/attention.c:142-175. */
void update_kv_cache_paged_batch(KVCacheManager* mgr, BlockTable** block_tables, int* seq_ids,
                                 float* k, float* v, int layer, int* pos_arr, int n_kv_heads,
                                 int head_dim, int num_tokens);

/* Causal attention of a contiguous run of one sequence over its block table.
 * Reference: src/kernels/attention.c:177-253. */
void paged_attention(float* out, float* q, KVCacheManager* mgr, BlockTable* block_table,
                     float* att, int layer, int pos, int max_seq_len, int n_heads,
                     int n_kv_heads, int head_dim, int num_tokens);

/* Causal attention of a ragged batch: token t attends to keys 0..pos_arr[t] of
 * sequence seq_ids[t].  `att` is accepted and ignored (scores never leave the CU).
 *   out, q        [num_tokens, n_heads * head_dim] fp32 in HBM
 *   block_tables  host array indexed by sequence id; entry s lists the physical blocks of
 *                 sequence s in logical order (only the first pos/block_size + 1 are read)
 *   seq_ids, pos_arr  host arrays of num_tokens ints
 * Query head h reads kv head h / (n_heads / n_kv_heads).  The pool's element type and
 * in-block layout are those chosen when `mgr` was initialised (nvllm_set_kv_elem_bytes).
 * Reference: src/kernels/attention.c:256-336. */
void paged_attention_batch(float* out, float* q, KVCacheManager* mgr, BlockTable** block_tables,
                           int* seq_ids, float* att, int layer, int* pos_arr, int max_seq_len,
                           int n_heads, int n_kv_heads, int head_dim, int num_tokens);

/* ASCII dashboards (reference src/visualizer.c:69-97, :100-172).  They read host
 * records only and are NOT provided by libnvllm_hip.so: link the reference's
 * visualizer.c next to the driver, as its Makefile does. */
void visualize_attention(float* att, int n_heads, int pos, int max_seq_len);
void visualize_kv_cache_usage(Sequence* seqs, int num_seqs, KVCacheManager* mgr,
                              int max_seq_len, int mode);

#ifdef __cplusplus
}
#endif

#endif /* OPS_H */
