/*
 * oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A CPU restatement of the reference's forward pass, exported under the same C
 * symbols as include/ops.h and include/model.h so that the very same driver code
 * (nano-vllm_amd/host/engine.c) can be linked against (a) this file, (b) the
 * reference's own sources built under oracle/_ref, or (c) the HIP library.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file; the HIP library never links or calls it.
 *
 * Every function keeps the reference's per-element operation order (strict
 * left-to-right fp32 sums, two roundings for a*b - c*d, the same libm calls), so
 * with the reference's compiler flags (gcc -O3, no -ffast-math, no FMA target) its
 * results are bit-identical to the reference's.  Pinning: the reference ships no
 * golden vectors or tests for this path; tests/test_oracle_vs_ref.py compares this
 * file with the reference compiled from /root/reference (oracle/_ref/libref.so)
 * on seeded inputs, and tests/golden/ holds vectors generated from that build by
 * tools/make_golden.py.
 *
 * Parallelism differs from the reference (OpenMP over independent outputs
 * everywhere); no output element is computed by more than one thread, so results
 * do not depend on the thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "model.h"
#include "ops.h"
#include "structs.h"

/* The driver defines these (reference src/main.c:51-55); engine.c does too. */
extern int g_paged_mode;
extern KVCacheManager g_kv_manager;

/* ------------------------------------------------------------------------- */
/* library life-cycle: nothing to do on the CPU (reference layers.c:11-17)    */
void init_ops(void) {}
void free_ops(void) {}

/* ------------------------------------------------------------------------- */
/* dense / element-wise operators                                             */

/* reference layers.c:23-38: sequential sum of squares, then (x*r)*w */
void rms_norm(float* o, float* x, float* weight, int size, int num_tokens, float eps) {
#pragma omp parallel for if (num_tokens > 8)
    for (int t = 0; t < num_tokens; t++) {
        const float* src = x + (size_t)t * size;
        float* dst = o + (size_t)t * size;
        float ss = 0.0f;
        for (int i = 0; i < size; i++) ss += src[i] * src[i];
        const float r = 1.0f / sqrtf(ss / size + eps);
        for (int i = 0; i < size; i++) dst[i] = src[i] * r * weight[i];
    }
}

/* strict left-to-right fp32 dot product (reference layers.c:56-63) */
static inline float dot_seq(const float* a, const float* b, int n) {
    float acc = 0.0f;
    for (int j = 0; j < n; j++) acc += a[j] * b[j];
    return acc;
}

/* reference layers.c:44-67 */
void matmul(float* out, float* in, float* weight, int num_tokens, int in_dim, int out_dim) {
    const long total = (long)num_tokens * out_dim;
#pragma omp parallel for schedule(static)
    for (long e = 0; e < total; e++) {
        const int t = (int)(e / out_dim);
        const int i = (int)(e % out_dim);
        out[e] = dot_seq(in + (size_t)t * in_dim, weight + (size_t)i * in_dim, in_dim);
    }
}

/* rotate every head of one token row by the angle table of `position`
 * (reference layers.c:118-141: freq via powf, angle = (float)pos * freq) */
static void rope_row(float* qrow, float* krow, int position, float theta, int head_dim,
                     int n_heads, int n_kv_heads) {
    for (int i = 0; i < head_dim; i += 2) {
        const float freq = 1.0f / powf(theta, i / (float)head_dim);
        const float ang = position * freq;
        const float c = cosf(ang);
        const float s = sinf(ang);
        for (int h = 0; h < n_heads; h++) {
            float* p = qrow + h * head_dim + i;
            const float a = p[0], b = p[1];
            p[0] = a * c - b * s;
            p[1] = a * s + b * c;
        }
        for (int h = 0; h < n_kv_heads; h++) {
            float* p = krow + h * head_dim + i;
            const float a = p[0], b = p[1];
            p[0] = a * c - b * s;
            p[1] = a * s + b * c;
        }
    }
}

/* reference layers.c:73-107 */
void apply_rope(float* q, float* k, int pos, float theta, int head_dim, int num_tokens,
                int n_heads, int n_kv_heads) {
    for (int t = 0; t < num_tokens; t++)
        rope_row(q + (size_t)t * n_heads * head_dim, k + (size_t)t * n_kv_heads * head_dim,
                 pos + t, theta, head_dim, n_heads, n_kv_heads);
}

/* reference layers.c:109-143 */
void apply_rope_batch(float* q, float* k, int* pos_arr, float theta, int head_dim,
                      int num_tokens, int n_heads, int n_kv_heads) {
#pragma omp parallel for if (num_tokens > 8)
    for (int t = 0; t < num_tokens; t++)
        rope_row(q + (size_t)t * n_heads * head_dim, k + (size_t)t * n_kv_heads * head_dim,
                 pos_arr[t], theta, head_dim, n_heads, n_kv_heads);
}

/* reference layers.c:149-161 */
void swiglu(float* hb, float* h_gate, float* h_up, int hidden_dim, int num_tokens) {
    const long n = (long)num_tokens * hidden_dim;
#pragma omp parallel for if (n > 65536)
    for (long i = 0; i < n; i++) {
        const float g = h_gate[i];
        const float act = g / (1.0f + expf(-g));
        hb[i] = act * h_up[i];
    }
}

/* reference layers.c:163-168 */
void accum(float* a, float* b, int size) {
    for (int i = 0; i < size; i++) a[i] += b[i];
}

/* reference layers.c:170-186 */
void softmax(float* x, int size) {
    float m = x[0];
    for (int i = 1; i < size; i++)
        if (x[i] > m) m = x[i];
    float denom = 0.0f;
    for (int i = 0; i < size; i++) {
        x[i] = expf(x[i] - m);
        denom += x[i];
    }
    for (int i = 0; i < size; i++) x[i] /= denom;
}

/* ------------------------------------------------------------------------- */
/* KV block pool (reference memory.c)                                          */

/* reference memory.c:8-44: stack filled so that the first pops give 0,1,2,... */
void init_kv_cache_manager(KVCacheManager* mgr, int block_size, int num_blocks, int n_layers,
                           int n_kv_heads, int head_dim) {
    mgr->block_size = block_size;
    mgr->num_blocks = num_blocks;
    mgr->free_blocks_count = num_blocks;
    mgr->free_block_indices = (int*)malloc((size_t)num_blocks * sizeof(int));
    for (int slot = 0; slot < num_blocks; slot++)
        mgr->free_block_indices[slot] = num_blocks - 1 - slot;
    const size_t elems = (size_t)n_layers * num_blocks * block_size * n_kv_heads * head_dim;
    mgr->pool_k = (float*)malloc(elems * sizeof(float));
    mgr->pool_v = (float*)malloc(elems * sizeof(float));
}

/* reference memory.c:47-51 */
void free_kv_cache_manager(KVCacheManager* mgr) {
    free(mgr->free_block_indices);
    free(mgr->pool_k);
    free(mgr->pool_v);
    mgr->free_block_indices = NULL;
    mgr->pool_k = mgr->pool_v = NULL;
}

/* reference memory.c:54-61 */
int alloc_block(KVCacheManager* mgr) {
    if (mgr->free_blocks_count <= 0) return -1;
    mgr->free_blocks_count -= 1;
    return mgr->free_block_indices[mgr->free_blocks_count];
}

/* reference memory.c:64-70 */
void free_block(KVCacheManager* mgr, int block_idx) {
    if (mgr->free_blocks_count >= mgr->num_blocks) return;
    mgr->free_block_indices[mgr->free_blocks_count] = block_idx;
    mgr->free_blocks_count += 1;
}

/* reference memory.c:74-82 */
long get_physical_offset(KVCacheManager* mgr, int layer, int block_idx, int block_offset,
                         int n_kv_heads, int head_dim) {
    const long row = (long)n_kv_heads * head_dim;
    const long blk = (long)mgr->block_size * row;
    const long lay = (long)mgr->num_blocks * blk;
    return (long)layer * lay + (long)block_idx * blk + (long)block_offset * row;
}

/* ------------------------------------------------------------------------- */
/* KV writes                                                                   */

/* reference attention.c:11-30 */
void update_kv_cache(float* key_cache, float* value_cache, float* k, float* v, int layer,
                     int pos, int max_seq_len, int n_kv_heads, int head_dim, int num_tokens) {
    const size_t row = (size_t)n_kv_heads * head_dim;
    const size_t base = (size_t)layer * max_seq_len * row;
    for (int t = 0; t < num_tokens; t++) {
        const size_t dst = base + (size_t)(pos + t) * row;
        memcpy(key_cache + dst, k + t * row, row * sizeof(float));
        memcpy(value_cache + dst, v + t * row, row * sizeof(float));
    }
}

/* one token row into the pool; returns 0 when the table is too short
 * (reference attention.c:153-173) */
static int kv_put(KVCacheManager* mgr, const BlockTable* bt, const float* krow, const float* vrow,
                  int layer, int position, int n_kv_heads, int head_dim) {
    const int lb = position / mgr->block_size;
    const int off = position % mgr->block_size;
    if (lb >= bt->num_blocks) {
        printf("Error: Block table too small! logical_idx=%d, num_blocks=%d\n", lb, bt->num_blocks);
        return 0;
    }
    const long dst = get_physical_offset(mgr, layer, bt->block_indices[lb], off, n_kv_heads, head_dim);
    const size_t bytes = (size_t)n_kv_heads * head_dim * sizeof(float);
    memcpy(mgr->pool_k + dst, krow, bytes);
    memcpy(mgr->pool_v + dst, vrow, bytes);
    return 1;
}

/* reference attention.c:109-139 */
void update_kv_cache_paged(KVCacheManager* mgr, BlockTable* block_table, float* k, float* v,
                           int layer, int pos, int n_kv_heads, int head_dim, int num_tokens) {
    const size_t row = (size_t)n_kv_heads * head_dim;
    for (int t = 0; t < num_tokens; t++)
        if (!kv_put(mgr, block_table, k + t * row, v + t * row, layer, pos + t, n_kv_heads, head_dim))
            return;
}

/* reference attention.c:142-175 */
void update_kv_cache_paged_batch(KVCacheManager* mgr, BlockTable** block_tables, int* seq_ids,
                                 float* k, float* v, int layer, int* pos_arr, int n_kv_heads,
                                 int head_dim, int num_tokens) {
    const size_t row = (size_t)n_kv_heads * head_dim;
    for (int t = 0; t < num_tokens; t++)
        if (!kv_put(mgr, block_tables[seq_ids[t]], k + t * row, v + t * row, layer, pos_arr[t],
                    n_kv_heads, head_dim))
            return;
}

/* ------------------------------------------------------------------------- */
/* attention                                                                   */

/* Address of key/value row `t` for one head: either the contiguous cache or the
 * block pool.  bt == NULL selects the contiguous layout. */
typedef struct {
    const float* k_base;
    const float* v_base;
    const BlockTable* bt;
    KVCacheManager* mgr;
    long layer_off;    /* contiguous layout only */
    int layer, n_kv_heads, head_dim;
} KvView;

static inline long kv_row_offset(const KvView* kv, int t) {
    if (!kv->bt) return kv->layer_off + (long)t * kv->n_kv_heads * kv->head_dim;
    const int bs = kv->mgr->block_size;
    return get_physical_offset(kv->mgr, kv->layer, kv->bt->block_indices[t / bs], t % bs,
                               kv->n_kv_heads, kv->head_dim);
}

/* One (token, query head): scores over keys 0..last, scale after the dot,
 * max / exp+sum / divide passes, then probability-weighted value sum in key
 * order (reference attention.c:282-333 and :60-100). */
static void attend_head(float* out_head, const float* q_head, const KvView* kv, int kv_head,
                        int last, float scale, float* scores) {
    const int hd = kv->head_dim;
    for (int t = 0; t <= last; t++) {
        const float* kr = kv->k_base + kv_row_offset(kv, t) + (long)kv_head * hd;
        float s = 0.0f;
        for (int i = 0; i < hd; i++) s += q_head[i] * kr[i];
        s *= scale;
        scores[t] = s;
    }
    float m = -INFINITY;
    for (int t = 0; t <= last; t++)
        if (scores[t] > m) m = scores[t];
    float denom = 0.0f;
    for (int t = 0; t <= last; t++) {
        const float e = expf(scores[t] - m);
        scores[t] = e;
        denom += e;
    }
    for (int t = 0; t <= last; t++) scores[t] /= denom;
    for (int i = 0; i < hd; i++) out_head[i] = 0.0f;
    for (int t = 0; t <= last; t++) {
        const float p = scores[t];
        const float* vr = kv->v_base + kv_row_offset(kv, t) + (long)kv_head * hd;
        for (int i = 0; i < hd; i++) out_head[i] += p * vr[i];
    }
}

/* Run attend_head for every (token, head).  The reference shares one
 * att[h*max_seq_len + t] scratch between tokens and therefore walks tokens
 * serially; a private scratch per thread gives the same numbers. */
static void attend_all(float* out, const float* q, const KvView* views, const int* last_pos,
                       int num_tokens, int n_heads, int n_kv_heads, int head_dim, int max_seq_len) {
    const float scale = 1.0f / sqrtf(head_dim);
    const int group = n_heads / n_kv_heads;
    const long work = (long)num_tokens * n_heads;
#pragma omp parallel
    {
        float* scores = (float*)malloc((size_t)max_seq_len * sizeof(float));
#pragma omp for schedule(dynamic, 4)
        for (long w = 0; w < work; w++) {
            const int t = (int)(w / n_heads);
            const int h = (int)(w % n_heads);
            const size_t at = ((size_t)t * n_heads + h) * head_dim;
            attend_head(out + at, q + at, &views[t], h / group, last_pos[t], scale, scores);
        }
        free(scores);
    }
}

/* reference attention.c:36-103 */
void multi_head_attention(float* out, float* q, float* key_cache, float* value_cache,
                          float* att, int layer, int pos, int max_seq_len, int n_heads,
                          int n_kv_heads, int head_dim, int num_tokens) {
    (void)att;
    KvView* views = (KvView*)malloc((size_t)num_tokens * sizeof(KvView));
    int* last = (int*)malloc((size_t)num_tokens * sizeof(int));
    for (int t = 0; t < num_tokens; t++) {
        views[t] = (KvView){key_cache, value_cache, NULL, NULL,
                            (long)layer * max_seq_len * n_kv_heads * head_dim, layer, n_kv_heads,
                            head_dim};
        last[t] = pos + t;
    }
    attend_all(out, q, views, last, num_tokens, n_heads, n_kv_heads, head_dim, max_seq_len);
    free(views);
    free(last);
}

/* reference attention.c:177-253 */
void paged_attention(float* out, float* q, KVCacheManager* mgr, BlockTable* block_table,
                     float* att, int layer, int pos, int max_seq_len, int n_heads,
                     int n_kv_heads, int head_dim, int num_tokens) {
    (void)att;
    KvView* views = (KvView*)malloc((size_t)num_tokens * sizeof(KvView));
    int* last = (int*)malloc((size_t)num_tokens * sizeof(int));
    for (int t = 0; t < num_tokens; t++) {
        views[t] = (KvView){mgr->pool_k, mgr->pool_v, block_table, mgr, 0, layer, n_kv_heads, head_dim};
        last[t] = pos + t;
    }
    attend_all(out, q, views, last, num_tokens, n_heads, n_kv_heads, head_dim, max_seq_len);
    free(views);
    free(last);
}

/* reference attention.c:256-336 */
void paged_attention_batch(float* out, float* q, KVCacheManager* mgr, BlockTable** block_tables,
                           int* seq_ids, float* att, int layer, int* pos_arr, int max_seq_len,
                           int n_heads, int n_kv_heads, int head_dim, int num_tokens) {
    (void)att;
    KvView* views = (KvView*)malloc((size_t)num_tokens * sizeof(KvView));
    for (int t = 0; t < num_tokens; t++)
        views[t] = (KvView){mgr->pool_k, mgr->pool_v, block_tables[seq_ids[t]], mgr, 0, layer,
                            n_kv_heads, head_dim};
    attend_all(out, q, views, pos_arr, num_tokens, n_heads, n_kv_heads, head_dim, max_seq_len);
    free(views);
}

/* ------------------------------------------------------------------------- */
/* model file -> fp32 tensors (reference model.c:15-285)                       */

/* IEEE binary16 -> binary32, exact, incl. subnormals / inf / nan
 * (reference model.c:15-48) */
static float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0x1Fu) {
        bits = sign | 0x7F800000u | (man << 13);
    } else if (exp != 0) {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    } else if (man == 0) {
        bits = sign;
    } else {
        int e = 113; /* 127 - 15 + 1 */
        while (!(man & 0x400u)) {
            man <<= 1;
            e--;
        }
        bits = sign | ((uint32_t)e << 23) | ((man & 0x3FFu) << 13);
    }
    float f;
    memcpy(&f, &bits, sizeof f);
    return f;
}

static float* read_tensor(FILE* f, size_t n, int file_is_fp32) {
    float* dst = (float*)malloc(n * sizeof(float));
    if (!dst) {
        fprintf(stderr, "oracle: out of memory for %zu floats\n", n);
        exit(1);
    }
    if (file_is_fp32) {
        if (fread(dst, sizeof(float), n, f) != n) {
            fprintf(stderr, "Failed to read FP32 tensor from file (expected %zu elements)\n", n);
            exit(1);
        }
        return dst;
    }
    uint16_t* raw = (uint16_t*)malloc(n * sizeof(uint16_t));
    if (!raw || fread(raw, sizeof(uint16_t), n, f) != n) {
        fprintf(stderr, "Failed to read FP16 tensor from file (expected %zu elements)\n", n);
        exit(1);
    }
    for (size_t i = 0; i < n; i++) dst[i] = half_bits_to_float(raw[i]);
    free(raw);
    return dst;
}

/* reference model.c:110-285 (format detection by file size, tensors grouped by
 * kind across layers, w_down stored before w_up) */
void load_model(Weights* w, Config* p, const char* checkpoint_path) {
    FILE* f = fopen(checkpoint_path, "rb");
    if (!f) {
        fprintf(stderr, "Failed to open model file: %s\n", checkpoint_path);
        exit(1);
    }
    if (fread(p, sizeof(int), 7, f) != 7) exit(1);
    const long header = ftell(f);
    fseek(f, 0, SEEK_END);
    const long long payload = ftell(f) - header;
    fseek(f, header, SEEK_SET);

    p->head_dim = p->dim / p->n_heads;
    p->rope_theta = 10000.0f;

    const long long D = p->dim, F = p->hidden_dim, L = p->n_layers, V = p->vocab_size;
    const long long QD = (long long)p->n_heads * p->head_dim;
    const long long KD = (long long)p->n_kv_heads * p->head_dim;
    const long long body = V * D + L * (2 * D + D * QD + 2 * D * KD + QD * D + 3 * D * F) + D;
    const long long head = V * D;
    const long long rope_tail = (long long)p->max_seq_len * p->head_dim * 4;

    int fp32, shared;
    if (payload == body * 4 || payload == body * 4 + rope_tail) { fp32 = 1; shared = 1; }
    else if (payload == (body + head) * 4)                      { fp32 = 1; shared = 0; }
    else if (payload == (body + head) * 2)                      { fp32 = 0; shared = 0; }
    else if (payload == body * 2)                               { fp32 = 0; shared = 1; }
    else                                                        { fp32 = 0; shared = 0; }

    w->layers = (LayerWeights*)malloc((size_t)L * sizeof(LayerWeights));
    w->_data_buffer = (void*)(intptr_t)L; /* layer count, for nvllm_free_model */
    w->token_embedding_table = read_tensor(f, (size_t)(V * D), fp32);
    for (int l = 0; l < L; l++) w->layers[l].rms_att_weight = read_tensor(f, (size_t)D, fp32);
    for (int l = 0; l < L; l++) w->layers[l].wq = read_tensor(f, (size_t)(D * QD), fp32);
    for (int l = 0; l < L; l++) w->layers[l].wk = read_tensor(f, (size_t)(D * KD), fp32);
    for (int l = 0; l < L; l++) w->layers[l].wv = read_tensor(f, (size_t)(D * KD), fp32);
    for (int l = 0; l < L; l++) w->layers[l].wo = read_tensor(f, (size_t)(QD * D), fp32);
    for (int l = 0; l < L; l++) w->layers[l].rms_ffn_weight = read_tensor(f, (size_t)D, fp32);
    for (int l = 0; l < L; l++) w->layers[l].w_gate = read_tensor(f, (size_t)(D * F), fp32);
    for (int l = 0; l < L; l++) w->layers[l].w_down = read_tensor(f, (size_t)(F * D), fp32);
    for (int l = 0; l < L; l++) w->layers[l].w_up = read_tensor(f, (size_t)(D * F), fp32);
    w->rms_final_weight = read_tensor(f, (size_t)D, fp32);
    w->lm_head = shared ? w->token_embedding_table : read_tensor(f, (size_t)(V * D), fp32);
    fclose(f);
}

/* Release what load_model allocated (the reference never frees weights). */
void nvllm_free_model(Weights* w) {
    if (!w->layers) return;
    /* the layer count is not stored in Weights; tensors of one kind were
     * allocated one per layer, so walk until the caller's count: kept in
     * _data_buffer by load_model below. */
    const long L = (long)(intptr_t)w->_data_buffer;
    for (long l = 0; l < L; l++) {
        LayerWeights* lw = &w->layers[l];
        free(lw->rms_att_weight); free(lw->wq); free(lw->wk); free(lw->wv); free(lw->wo);
        free(lw->rms_ffn_weight); free(lw->w_gate); free(lw->w_up); free(lw->w_down);
    }
    if (w->lm_head != w->token_embedding_table) free(w->lm_head);
    free(w->token_embedding_table);
    free(w->rms_final_weight);
    free(w->layers);
    memset(w, 0, sizeof *w);
}

/* reference model.c:287-321 */
void malloc_run_state(RunState* s, Config* p) {
    const size_t S = (size_t)p->max_seq_len;
    s->x = (float*)malloc(S * p->dim * sizeof(float));
    s->xb = (float*)malloc(S * p->dim * sizeof(float));
    s->xb2 = (float*)malloc(S * p->dim * sizeof(float));
    s->hb = (float*)malloc(S * p->hidden_dim * sizeof(float));
    s->hb2 = (float*)malloc(S * p->hidden_dim * sizeof(float));
    s->q = (float*)malloc(S * p->dim * sizeof(float));
    s->k = (float*)malloc(S * p->dim * sizeof(float));
    s->v = (float*)malloc(S * p->dim * sizeof(float));
    s->att = (float*)malloc((size_t)p->n_heads * S * sizeof(float));
    s->logits = (float*)malloc(S * p->vocab_size * sizeof(float));
    s->key_cache = s->value_cache = NULL;
    if (!g_paged_mode) {
        const size_t n = (size_t)p->n_layers * S * p->n_kv_heads * p->head_dim;
        s->key_cache = (float*)malloc(n * sizeof(float));
        s->value_cache = (float*)malloc(n * sizeof(float));
    }
}

/* reference model.c:323-336 */
void free_run_state(RunState* s) {
    free(s->x); free(s->xb); free(s->xb2); free(s->hb); free(s->hb2);
    free(s->q); free(s->k); free(s->v); free(s->att); free(s->logits);
    free(s->key_cache); free(s->value_cache);
}

/* reference model.c:339-344 */
void transformer(int* tokens, int num_tokens, int pos, Config* p, RunState* s, Weights* w,
                 BlockTable* bt) {
    (void)tokens; (void)num_tokens; (void)pos; (void)p; (void)s; (void)w; (void)bt;
    printf("Error: Old transformer called in Continuous Batching mode.\n");
    exit(1);
}

/* reference model.c:347-419 */
void transformer_batch(int* tokens, int num_tokens, int* pos_arr, int* seq_ids,
                       int* output_indices, int num_outputs, Config* p, RunState* s,
                       Weights* w, BlockTable** block_tables) {
    const int D = p->dim, F = p->hidden_dim, hd = p->head_dim;
    const int QD = p->n_heads * hd, KD = p->n_kv_heads * hd;
    if (!g_paged_mode) {
        printf("Error: Continuous batching requires paged mode.\n");
        exit(1);
    }
    for (int t = 0; t < num_tokens; t++) /* model.c:352-355, int index arithmetic */
        memcpy(s->x + (size_t)t * D, w->token_embedding_table + tokens[t] * D, (size_t)D * sizeof(float));

    for (int l = 0; l < p->n_layers; l++) {
        LayerWeights* lw = &w->layers[l];
        rms_norm(s->xb, s->x, lw->rms_att_weight, D, num_tokens, 1e-5f);
        matmul(s->q, s->xb, lw->wq, num_tokens, D, QD);
        matmul(s->k, s->xb, lw->wk, num_tokens, D, KD);
        matmul(s->v, s->xb, lw->wv, num_tokens, D, KD);
        apply_rope_batch(s->q, s->k, pos_arr, p->rope_theta, hd, num_tokens, p->n_heads, p->n_kv_heads);
        update_kv_cache_paged_batch(&g_kv_manager, block_tables, seq_ids, s->k, s->v, l, pos_arr,
                                    p->n_kv_heads, hd, num_tokens);
        paged_attention_batch(s->xb2, s->q, &g_kv_manager, block_tables, seq_ids, s->att, l, pos_arr,
                              p->max_seq_len, p->n_heads, p->n_kv_heads, hd, num_tokens);
        matmul(s->xb, s->xb2, lw->wo, num_tokens, QD, D);
        accum(s->x, s->xb, num_tokens * D);
        rms_norm(s->xb, s->x, lw->rms_ffn_weight, D, num_tokens, 1e-5f);
        matmul(s->hb, s->xb, lw->w_gate, num_tokens, D, F);
        matmul(s->hb2, s->xb, lw->w_up, num_tokens, D, F);
        swiglu(s->hb, s->hb, s->hb2, F, num_tokens);
        matmul(s->xb, s->hb, lw->w_down, num_tokens, F, D);
        accum(s->x, s->xb, num_tokens * D);
    }
    rms_norm(s->x, s->x, w->rms_final_weight, D, num_tokens, 1e-5f);
    for (int k = 0; k < num_outputs; k++) /* model.c:411-414 */
        memcpy(s->xb + (size_t)k * D, s->x + (size_t)output_indices[k] * D, (size_t)D * sizeof(float));
    matmul(s->logits, s->xb, w->lm_head, num_outputs, D, p->vocab_size);
}

/* ------------------------------------------------------------------------- */
/* sampler (reference sampler.c); layout of Sampler must match main.c:39-45   */

typedef struct {
    unsigned long long state;
    float temperature;
    float topp;
    ProbIndex* probindex;
    int vocab_size;
} Sampler;

/* reference sampler.c:20-27 */
void build_sampler(Sampler* s, int vocab_size, float temperature, float topp, unsigned long long seed) {
    s->state = seed;
    s->temperature = temperature;
    s->topp = topp;
    s->vocab_size = vocab_size;
    s->probindex = (ProbIndex*)malloc((size_t)vocab_size * sizeof(ProbIndex));
    srand((unsigned int)seed);
}

static int by_prob_desc(const void* a, const void* b) {
    const float pa = ((const ProbIndex*)a)->prob, pb = ((const ProbIndex*)b)->prob;
    return (pa > pb) ? -1 : (pa < pb) ? 1 : 0;
}

/* Nucleus sampling with a caller-supplied coin in [0,1]: candidates are the
 * entries >= (1-topp)/(n-1), sorted descending, truncated where the running sum
 * first exceeds topp (reference sampler.c:41-82). */
int oracle_sample_topp(float* probs, int n, float topp, ProbIndex* scratch, float coin) {
    const float cutoff = (1.0f - topp) / (n - 1);
    int cand = 0;
    for (int i = 0; i < n; i++)
        if (probs[i] >= cutoff) {
            scratch[cand].index = i;
            scratch[cand].prob = probs[i];
            cand++;
        }
    qsort(scratch, cand, sizeof(ProbIndex), by_prob_desc);
    float mass = 0.0f;
    int last = cand - 1;
    for (int i = 0; i < cand; i++) {
        mass += scratch[i].prob;
        if (mass > topp) { last = i; break; }
    }
    const float r = coin * mass;
    float cdf = 0.0f;
    for (int i = 0; i <= last; i++) {
        cdf += scratch[i].prob;
        if (r < cdf) return scratch[i].index;
    }
    return scratch[last].index;
}

/* reference sampler.c:86-97: first index of the maximum */
int oracle_argmax(const float* v, int n) {
    int best = 0;
    float top = v[0];
    for (int i = 1; i < n; i++)
        if (v[i] > top) { top = v[i]; best = i; }
    return best;
}

/* temperature + softmax in place (reference sampler.c:100-121) */
void oracle_tempered_softmax(float* logits, int n, float temperature) {
    for (int i = 0; i < n; i++) logits[i] /= temperature;
    float m = logits[0];
    for (int i = 1; i < n; i++)
        if (logits[i] > m) m = logits[i];
    float denom = 0.0f;
    for (int i = 0; i < n; i++) {
        const float e = expf(logits[i] - m);
        logits[i] = e;
        denom += e;
    }
    for (int i = 0; i < n; i++) logits[i] /= denom;
}

/* reference sampler.c:84-140 */
int sample(Sampler* s, float* logits) {
    if (s->temperature == 0.0f) return oracle_argmax(logits, s->vocab_size);
    oracle_tempered_softmax(logits, s->vocab_size, s->temperature);
    const float coin = (float)rand() / (float)RAND_MAX;
    if (s->topp <= 0 || s->topp >= 1) {
        float cdf = 0.0f;
        for (int i = 0; i < s->vocab_size; i++) {
            cdf += logits[i];
            if (coin < cdf) return i;
        }
        return s->vocab_size - 1;
    }
    return oracle_sample_topp(logits, s->vocab_size, s->topp, s->probindex, coin);
}
