/*
 * nvllm.h -- entry points of libnvllm_hip.so that have no one-to-one counterpart
 * in the reference's headers: device selection, result hand-off modes, kernel
 * timing, and an N-sequence driver that restates the scheduler loop which the
 * reference hard-codes for two sequences inside main() (src/main.c:151-166,
 * 259-419).  Plain C ABI: pointers, ints, no C++ or torch types.
 */
#ifndef NVLLM_H
#define NVLLM_H

#include <stddef.h>

#include "structs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ runtime */

/* Select the GPU every later call of this process uses (one process per GPU).
 * Must be called before init_ops()/device_malloc(); default is device 0. */
int nvllm_set_device(int ordinal);

/* hipStream_t on which the library enqueues every kernel and copy. */
void* nvllm_stream(void);

/* What transformer_batch copies back to the host for each output row.
 *   NVLLM_LOGITS_HOST   (default, the reference's contract, main.c:377): the
 *                       full fp32 row lands in RunState.logits.
 *   NVLLM_LOGITS_ARGMAX only the index of the row maximum is copied back
 *                       (4 bytes instead of vocab_size*4); RunState.logits is
 *                       left untouched and sample() with temperature 0 answers
 *                       from those indices. */
enum { NVLLM_LOGITS_HOST = 0, NVLLM_LOGITS_ARGMAX = 1 };
void nvllm_set_logits_mode(int mode);
int  nvllm_get_logits_mode(void);

/* Arg-max indices of the most recent transformer_batch call (host array with
 * num_outputs entries, valid until the next call). */
const int* nvllm_last_argmax(int* num_outputs);

/* The device arg-max of the forward pass on caller-supplied rows: out[r] (host) = first
 * index of the maximum of logits[r*vocab_size ..] (device memory), the reference's rule
 * (strict '>' scan from index 0, src/sampler.c:86-97: ties go to the lowest index, a row
 * of -inf answers 0).  slices > 1 takes the path of a tensor-parallel classifier instead:
 * the vocabulary is cut into `slices` runs of whole 16-row blocks as nvllm_tp_shard cuts
 * it over ranks, every slice reports one (maximum, index) pair per row and the pairs are
 * merged.  Synchronous.  Returns 0, or -1 on bad arguments. */
int nvllm_argmax_rows(int* out, const float* logits, int num_rows, int vocab_size, int slices);

/* Element type of the KV pool that init_kv_cache_manager allocates next:
 * 4 = fp32 (the reference's row layout, src/memory.c:37-43, bit-exact copies);
 * 2 = fp16 (half the bytes; values rounded to nearest-even and saturated to the
 *     finite fp16 range on write).  An fp16 pool keeps the [layer][block]
 *     addressing -- block ids, block tables and the allocator are unchanged -- but
 *     stores, inside a block, one 16-position page per kv head in matrix-core
 *     operand order; it needs block_size 16 and head_dim % 16 == 0 and is opaque to
 *     the caller (the pool pointers are never dereferenced by the reference's
 *     driver either). */
void nvllm_set_kv_elem_bytes(int bytes);
int  nvllm_get_kv_elem_bytes(void);

/* -------------------------------------------------- one launch per decode step */
/* No counterpart in the reference, whose layer loop (src/model.c:358-394) calls one operator
 * after the other.  With the switch on (off by default; environment NVLLM_DECODE_CHAIN=1 turns it
 * on at start-up) transformer_batch runs ALL layers of a decode-sized step -- at most 64 tokens,
 * fp16 KV pool, the whole model on this GPU, a model shape the library has a tiling for -- as one
 * kernel launch whose workgroups hand their tiles to each other through arrival counters, so that
 * the weights of the next projection are already travelling while the current one is reduced and
 * stored; every other step takes one launch per operator as before.  Results are identical either
 * way (same tiles, same summation order).  The one-launch form needs every CU of the GPU for
 * itself: two processes that drive the same GPU must leave it off.  Measured on MI355X it is
 * no faster than the separate kernels (a hand-off through memory costs what a kernel boundary
 * costs), which is why it is opt-in.
 * nvllm_decode_chain_launches() counts the steps that took the one-launch form. */
void nvllm_set_decode_chain(int on);
int  nvllm_get_decode_chain(void);
long nvllm_decode_chain_launches(void);

/* ------------------------------------------------ matmul without allocation */
/* No counterpart in the reference.  For callers that cannot allocate device memory:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same result as matmul() (ops.h; reference src/kernels/layers.c:44-67) for a `weight`
 * handle that load_model produced and at most 64 tokens with in_dim <= 8192: the re-packed
 * activations, the split-K partial tiles and the arrival counters live in a static
 * device array of the library (10.1 MiB per GPU, held until the process ends) instead of
 * buffers the library allocates on demand.  Calls no hipMalloc / hipFree.  (The same K split
 * as matmul(), hence the same bits, as long as the split's partial tiles fit 8 MiB of that
 * array -- every split matmul() picks on a 256-CU GPU does; a split that does not fit is
 * dropped, the result then agrees to rounding only.)  The library
 * must already be initialised (init_ops()).  Enqueues on nvllm_stream() and returns
 * without synchronising, like matmul().  All calls on one GPU share the one array:
 * calls that overlap on a GPU race.
 * Returns 0; -1 library not initialised; -2 `weight` is not a projection handle of this
 * shape; -3 sizes beyond the limits (message on stderr, nothing enqueued). */
int nvllm_matmul_noalloc(float* out, const float* in, const float* weight, int num_tokens,
                         int in_dim, int out_dim);

/* ---------------------------------------------------------- kernel timing */
/* With timing on, every kernel is launched with a start and a stop event attached to
 * its dispatch (hipExtLaunchKernelGGL), so the elapsed time is the kernel's own
 * begin-to-end time on the library stream.  Classes are small integers;
 * nvllm_prof_name() names them. */
void nvllm_prof_enable(int on);
void nvllm_prof_reset(void);
int  nvllm_prof_num_classes(void);
const char* nvllm_prof_name(int cls);
/* Sums since the last reset: launches, total milliseconds, algorithmic bytes
 * (what the kernel must read+write at least once) and flops.  Resolves pending
 * events (synchronises the stream). */
void nvllm_prof_get(int cls, long long* launches, double* total_ms, double* bytes, double* flops);

/* ------------------------------------------------------- tensor parallelism */
/* No counterpart in the reference, which is one process on one device.  The forward
 * pass of ONE model is split over `world` processes (one per GPU): each keeps
 * n_heads/world attention heads (whole kv heads with their query heads) and
 * hidden_dim/world columns of the MLP, and the partial results of the two projections
 * that return to the residual width are summed over the ranks -- after the wo
 * projection (src/model.c:385) and after w_down (src/model.c:392); the classifier
 * (src/model.c:411-418) is split by vocabulary rows and the ranks gather arg-max pairs
 * or logits slices.  All ranks must
 * issue identical calls with identical arguments (same model file, requests, block
 * tables); Config, RunState and the block tables keep their full-model meaning, the
 * weight handles in `Weights` and the KV pool cover this rank's slice only.
 * Set up before load_model() and init_kv_cache_manager(). */
typedef struct {
    int rank, world;
    int n_heads, head0;        /* query heads of this rank: [head0, head0 + n_heads)        */
    int n_kv_heads, kv_head0;  /* kv heads of this rank                                     */
    int hidden, hidden0;       /* columns of the hidden activation: [hidden0, hidden0+hidden) */
    int vocab, vocab0;         /* classifier rows [vocab0, vocab0 + vocab); may be empty      */
    int vocab_slice;           /* rows per rank before clipping (a multiple of 16)            */
} NvllmTpShard;
/* The slice rank `rank` of `world` owns for this model shape.  0, or -1 when the shape
 * does not divide (kv heads and heads by world, hidden_dim by 32*world). */
int nvllm_tp_shard(const Config* cfg, int rank, int world, NvllmTpShard* out);

/* RCCL over xGMI.  One rank obtains an id and hands it to the others by whatever means
 * the host has (a file, a socket, torch.distributed); then every rank calls
 * nvllm_tp_init with the same id.  The all-reduces run on nvllm_stream(). */
#define NVLLM_TP_ID_BYTES 128
int nvllm_tp_unique_id(void* id /* NVLLM_TP_ID_BYTES */);
int nvllm_tp_init(int rank, int world, const void* id);

/* The same with the exchange supplied by the caller: fn must leave the element-wise sum
 * over all ranks of the `count` floats at device_buf in device_buf on every rank,
 * ordered after the work already enqueued on hip_stream and before work enqueued
 * later (enqueue on that stream, or synchronise it).  The library also builds its gathers
 * from it (a sum over slices padded with zeros). */
typedef void (*nvllm_allreduce_fn)(void* ctx, float* device_buf, size_t count, void* hip_stream);
int nvllm_tp_init_custom(int rank, int world, nvllm_allreduce_fn fn, void* ctx);

/* The same with the library's own exchange over peer-mapped memory (no RCCL): every rank owns one
 * exchange segment in its HBM, exported by hipIpcGetMemHandle; a rank's all-reduce kernel stores its
 * slices straight into the other ranks' segments over xGMI and reads their results back (one launch
 * per all-reduce, two flag hops, count/world floats each way per link; every element is summed by
 * one rank in rank order, so all ranks hold the same bits).  No counterpart in the reference, whose
 * forward pass runs in one process (the sums it replaces: src/model.c:385,392).
 *   every rank:  nvllm_tp_peer_handle(mine);            then all-gather the handles in rank order
 *   every rank:  nvllm_tp_init_peer(rank, world, all);  (world <= 8)
 * Every rank must make the same sequence of forward calls.  A rank that waits 30 s for a peer stops
 * the process with a message at its next call.  Both return 0, or -1 with a message. */
#define NVLLM_TP_PEER_HANDLE_BYTES 64
int nvllm_tp_peer_handle(void* handle /* NVLLM_TP_PEER_HANDLE_BYTES */);
int nvllm_tp_init_peer(int rank, int world, const void* handles /* world x NVLLM_TP_PEER_HANDLE_BYTES */);
/* all-reduces made through the peer exchange since nvllm_tp_init_peer, and the floats they carried */
void nvllm_tp_peer_stats(unsigned long long* calls, unsigned long long* floats);

/* The session's all-reduce on a caller's device buffer (in place, sum over the ranks, enqueued on
 * nvllm_stream()): for hosts that need the same exchange outside the forward pass.  -1 without a session. */
int nvllm_tp_allreduce(float* device_buf, size_t count);

void nvllm_tp_finalize(void);
int  nvllm_tp_rank(void);
int  nvllm_tp_world(void);

/* --------------------------------------------- the forward pass in two halves */
/* No counterpart in the reference, whose transformer_batch returns with the results.  A driver
 * whose scheduling does not depend on the sampled token VALUES (greedy decoding, sequences that
 * end by length) can keep the GPU busy across steps: enqueue pass n+1 before collecting pass n,
 * naming for each decode row the output row of pass n whose arg-max it continues with; the GPU
 * takes those token ids from HBM.  In NVLLM_LOGITS_HOST mode the logits rows of a pass are copied to
 * the `logits` member of the RunState given to it, complete once the pass is collected: passes in flight
 * at the same time need a RunState each.  Same arguments as transformer_batch (src/model.c:347-419) plus
 *   token_src[t]  -1: the token id is tokens[t];  k >= 0: it is the arg-max of output row k of
 *                 the pass enqueued immediately before this one (null = all -1).
 * Returns a ticket (>= 0), or -1 on a null argument or num_tokens == 0.  The host
 * arrays may be reused as soon as the call returns.  At most 4 passes may be in flight. */
long nvllm_forward_enqueue(int* tokens, int num_tokens, int* pos_arr, int* seq_ids, int* output_indices,
                           int num_outputs, const int* token_src, Config* p, RunState* s, Weights* w,
                           BlockTable** block_tables);
/* Waits for that pass and copies its arg-max indices (min(cap, num_outputs) of them); returns
 * num_outputs. */
int nvllm_forward_collect(long ticket, int* argmax_out, int cap);

/* ------------------------------------------------------ N-sequence driver */
/* The reference's scheduler with the sequence count as a parameter: arrival
 * gating, is_prefill = pos < n_prompt-1, chunked prefill, batch-capacity check,
 * one block allocated whenever a position is a multiple of block_size, one
 * sample() per scheduled sequence per step -- including mid-prefill chunks,
 * whose sample overwrites current_token exactly as main.c:353-355,402 do. */
typedef struct NvllmEngine NvllmEngine;

typedef struct {
    int chunk_size;      /* --chunk-size; <= 0 means unlimited (main.c:80)          */
    int block_size;      /* 0 -> 16 (main.c:188)                                    */
    int num_blocks;      /* 0 -> n_seqs * ceil(max_seq_len/block_size) (main.c:189) */
    int max_seq_len;     /* 0 -> header value; raised to >= 1024 if raise_to_1024   */
    int raise_to_1024;   /* main.c:140-143                                          */
    int recycle_blocks;  /* 1: free_block() every block of a sequence when it
                               finishes (the reference never does, memory.c:64-70)  */
    int max_steps;       /* 0 -> 500 (main.c:418); the loop stops when step > this  */
    float temperature;   /* 0 => greedy                                             */
    float topp;
    unsigned long long seed;
} NvllmEngineOptions;

typedef struct {
    long long steps;            /* scheduler iterations that ran a forward pass */
    long long tokens_forwarded; /* sum of batch sizes                           */
    long long prefill_tokens;   /* tokens forwarded by prefill chunks           */
    long long decode_tokens;    /* tokens forwarded by decode rows              */
    long long outputs;          /* sample() calls                               */
    double forward_seconds;     /* wall time inside transformer_batch; two-halves loop: enqueueing a pass
                                   plus waiting for its results (sampling excluded)                     */
    double sample_seconds;      /* wall time inside sample()                    */
    double prefill_step_seconds;/* forward time of passes containing any prefill (a pass's wait counts
                                   for that pass, whichever step collects it)                           */
    double decode_step_seconds; /* forward time of pure-decode passes           */
    long long decode_step_tokens;/* tokens forwarded by pure-decode steps       */
    long long prefill_step_tokens;
} NvllmEngineStats;

NvllmEngine* nvllm_engine_create(const char* model_path, int n_seqs, const NvllmEngineOptions* opt);
void nvllm_engine_destroy(NvllmEngine* e);

/* Define sequence `seq`: prompt token ids (copied), number of tokens to generate
 * and the scheduler step at which it arrives.  Returns 0, or -1 on bad input. */
int nvllm_engine_set_request(NvllmEngine* e, int seq, const int* prompt, int n_prompt,
                             int n_new, int arrival_step);

/* Measurement aid (no counterpart in the reference): treat the first `upto` prompt tokens of a sequence
 * that has not started yet as already processed -- its blocks are allocated in the loop's order
 * (main.c:320-335), the K/V rows of those positions keep whatever the pool holds, and the sequence
 * continues at position `upto` (at most n_prompt - 1) with prompt[upto].  A CPU timing of decode steps at
 * a long context uses it instead of a prefill that would take the CPU half an hour; results of such a
 * run are not a model's output.  Returns the position set, or -1 on bad input. */
int nvllm_engine_skip_prefill(NvllmEngine* e, int seq, int upto);

/* One iteration of the loop at main.c:259-419.  Returns 1 while work remains,
 * 0 once every sequence has finished (or the step cap was hit). */
int nvllm_engine_step(NvllmEngine* e);
/* Up to `n` iterations; returns how many ran before the loop ended. */
int nvllm_engine_run(NvllmEngine* e, int n);

int  nvllm_engine_global_step(const NvllmEngine* e);
void nvllm_engine_get_stats(const NvllmEngine* e, NvllmEngineStats* out);
void nvllm_engine_reset_stats(NvllmEngine* e);
const Config* nvllm_engine_config(const NvllmEngine* e);

/* Per-sequence state: returns pos; copies min(cap, filled) history entries. */
int nvllm_engine_seq_state(const NvllmEngine* e, int seq, int* status, int* current_token,
                           int* n_prompt, int* seq_len);
int nvllm_engine_history(const NvllmEngine* e, int seq, int* out, int cap);
int nvllm_engine_block_table(const NvllmEngine* e, int seq, int* out, int cap);

/* The ragged batch of the most recent forward pass. */
int nvllm_engine_last_batch(const NvllmEngine* e, int* tokens, int* pos, int* seq_ids, int cap,
                            int* output_indices, int* sampled, int out_cap, int* num_outputs);
/* Host logits of the most recent forward pass ([num_outputs, vocab]); valid in
 * NVLLM_LOGITS_HOST mode only. */
const float* nvllm_engine_logits(const NvllmEngine* e);

/* Allocation trace: every alloc_block() the scheduler made, in order, as
 * (seq, logical_block, physical_block) triples.  Returns the number of triples. */
int nvllm_engine_alloc_trace(const NvllmEngine* e, int* out, int cap_triples);

#ifdef __cplusplus
}
#endif

#endif /* NVLLM_H */
