/*
 * structs.h -- binary layout of the records that cross the drop-in boundary.
 *
 * The driver (reference src/main.c) allocates these records itself, pokes at a
 * handful of their fields directly (g_kv_manager.block_size at main.c:322,
 * batch_state.logits at main.c:377, seqs[i].table at main.c:200-201,332-333)
 * and hands pointers to them to the library.  Field order, field types and
 * therefore sizeof/offsetof of every record below are part of the ABI and are
 * kept identical to reference include/structs.h:6-111.  What the *pointers*
 * inside them refer to is this library's business and is documented per field:
 * "HBM" means memory obtained from device_malloc(), "host" means ordinary
 * process memory.
 *
 * tests/test_abi.py checks sizeof/offsetof of every record against the numbers
 * the reference header produces.
 */
#ifndef STRUCTS_H
#define STRUCTS_H

#include <stdint.h>

/* ---- model hyper-parameters ------------------------------------------------
 * The first seven ints are read verbatim from the model file header
 * (reference src/model.c:118); head_dim and rope_theta are derived by
 * load_model (reference src/model.c:144-146).                                */
typedef struct {
    int dim;          /* residual width D                                      */
    int hidden_dim;   /* MLP width F                                           */
    int n_layers;     /* L                                                     */
    int n_heads;      /* query heads H                                         */
    int n_kv_heads;   /* key/value heads KVH (H % KVH == 0)                    */
    int vocab_size;   /* V                                                     */
    int max_seq_len;  /* S: positions per sequence AND tokens per forward call */
    int head_dim;     /* D / H, filled in by load_model                        */
    float rope_theta; /* 10000.0f, filled in by load_model                     */
} Config;

/* ---- weights -----------------------------------------------------------------
 * Reference layout: one fp32 buffer per tensor, matrices [out, in] row-major
 * (reference include/structs.h:19-38).  In this library every pointer is an
 * opaque HBM handle returned by load_model: matrices stay fp16 in HBM in an
 * MFMA-fragment tile order, norm vectors are fp32.  The handles are accepted
 * wherever ops.h takes a `weight` argument; they must not be dereferenced on
 * the host.                                                                   */
typedef struct {
    float* rms_att_weight; /* [D]            fp32 HBM                          */
    float* wq;             /* [H*hd, D]      fp16 tiles, HBM handle            */
    float* wk;             /* [KVH*hd, D]                                      */
    float* wv;             /* [KVH*hd, D]                                      */
    float* wo;             /* [D, H*hd]                                        */
    float* rms_ffn_weight; /* [D]            fp32 HBM                          */
    float* w_gate;         /* [F, D]                                           */
    float* w_up;           /* [F, D]                                           */
    float* w_down;         /* [D, F]                                           */
} LayerWeights;

typedef struct {
    float* token_embedding_table; /* [V, D] fp16 row-major, HBM handle         */
    LayerWeights* layers;         /* host array of n_layers records            */
    float* rms_final_weight;      /* [D] fp32 HBM                              */
    float* lm_head;               /* [V, D] fp16 tiles, HBM handle             */
    void* _data_buffer;           /* unused by the reference (structs.h:37);
                                     here: the library's private model record  */
} Weights;

/* ---- per-call activation workspace ------------------------------------------
 * Sized by malloc_run_state for max_seq_len tokens (reference
 * src/model.c:297-310).  All buffers are fp32 in HBM except `logits`, which is
 * page-locked host memory because the driver reads it (main.c:377) and the
 * sampler rewrites it in place (reference src/sampler.c:100-121).            */
typedef struct {
    float* x;      /* [S, D]   residual stream                                 */
    float* xb;     /* [S, D]                                                   */
    float* xb2;    /* [S, D]                                                   */
    float* hb;     /* [S, F]                                                   */
    float* hb2;    /* [S, F]                                                   */
    float* q;      /* [S, D]                                                   */
    float* k;      /* [S, D]                                                   */
    float* v;      /* [S, D]                                                   */
    float* att;    /* [H, S]   score scratch of the reference; unused here     */
    float* logits; /* [S, V]   HOST-visible                                    */
    float* key_cache;   /* [L, S, KVH, hd] only when !g_paged_mode, else NULL  */
    float* value_cache; /* same                                                */
} RunState;

/* ---- paged KV pool ----------------------------------------------------------
 * Reference include/structs.h:63-81 and src/memory.c:8-82.  The free stack is
 * host memory and is manipulated only by alloc_block/free_block; pool_k and
 * pool_v live in HBM, laid out [n_layers, num_blocks, block_size, KVH, hd].  */
typedef struct {
    int block_size;
    int num_blocks;
    int free_blocks_count;
    int* free_block_indices; /* host; top of stack = [free_blocks_count-1]     */
    float* pool_k;           /* HBM                                            */
    float* pool_v;           /* HBM                                            */
} KVCacheManager;

typedef struct {
    int* block_indices; /* host; logical block -> physical block               */
    int num_blocks;     /* entries in use                                      */
} BlockTable;

/* ---- scheduler records (owned and interpreted by the driver only) ---------- */
typedef enum {
    SEQ_WAITING,
    SEQ_PREFILLING,
    SEQ_DECODING,
    SEQ_FINISHED
} SeqStatus;

typedef struct {
    int id;
    int active;
    int pos;               /* tokens already forwarded                         */
    int seq_len;           /* prompt length + tokens to generate               */
    int current_token;
    int* prompt_tokens;
    int num_prompt_tokens;
    int* output_history;
    RunState* state;       /* unused (main.c:180)                              */
    BlockTable table;
    int arrival_step;
    SeqStatus status;
    int first_token_latency;
} Sequence;

/* top-p scratch element (reference src/sampler.c:41-82) */
typedef struct {
    float prob;
    int index;
} ProbIndex;

#endif /* STRUCTS_H */
