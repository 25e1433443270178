/*
 * engine.c -- N-sequence continuous-batching driver (host C).
 *
 * The reference's scheduler lives inside main() and is fixed at two sequences
 * with built-in prompts (src/main.c:151-166, 259-419).  This file restates that
 * loop with the sequence count, prompts, arrival steps and generation lengths as
 * inputs, so that every batch it builds, every block it allocates and every
 * token it feeds back is what main() would produce for the same requests.  It
 * talks to the forward pass only through include/ops.h, include/model.h and
 * include/sampler.h, so one object file serves the HIP library, the CPU oracle
 * and the reference build alike.
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "model.h"
#include "nvllm.h"
#include "ops.h"
#include "sampler.h"
#include "structs.h"

/* The reference's driver defines these three (src/main.c:51-55) and model.c reads
 * two of them through `extern`.  Weak, so that an executable that brings its own
 * main.c definitions wins. */
__attribute__((weak)) int g_visualize_paged = 0;
__attribute__((weak)) int g_paged_mode = 1;
__attribute__((weak)) KVCacheManager g_kv_manager;

/* Optional: a backend that can release what load_model allocated exports this
 * (the reference has no free_model and simply leaks at exit). */
extern void nvllm_free_model(Weights* w) __attribute__((weak));
/* Optional: a backend that can run the forward pass in two halves (include/nvllm.h).  With greedy
 * decoding nothing the scheduler decides depends on the VALUES of the sampled tokens (sequences end
 * by length), so the driver enqueues pass n+1 -- decode rows take their token from pass n's arg-max
 * on the GPU -- before it collects pass n: the GPU never waits for the host between steps.  With the
 * arg-max hand-off the collected indices are the tokens; with logits rows on the host the tokens are
 * still the sampler's, taken from the collected pass's rows while the next pass runs (two RunStates
 * alternate), and checked against the index the GPU continued with.  Every query of sampled tokens
 * or logits first collects what is in flight, so the driver's observable state is that of the
 * step-by-step loop. */
extern long nvllm_forward_enqueue(int* tokens, int num_tokens, int* pos_arr, int* seq_ids, int* output_indices,
                                  int num_outputs, const int* token_src, Config* p, RunState* s, Weights* w,
                                  BlockTable** block_tables) __attribute__((weak));
extern int nvllm_forward_collect(long ticket, int* argmax_out, int cap) __attribute__((weak));
extern int nvllm_get_logits_mode(void) __attribute__((weak));

struct NvllmEngine {
    NvllmEngineOptions opt;
    Config config;
    Weights weights;
    RunState state;          /* the step-by-step loop's; pass slot 0 of the two-halves loop */
    RunState state2;         /* pass slot 1 (allocated on first use: host logits rows of two passes in flight) */
    int state2_ready, flip, pend_state, seen_state;
    Sampler sampler;
    int n_seqs;
    Sequence* seqs;
    int* history_cap;
    int capacity;          /* MAX_BATCH_CAPACITY = max_seq_len tokens (main.c:148) */
    int max_blocks_per_seq;
    int global_step;
    int finished;
    /* batch buffers (main.c:249-256) */
    int* batch_tokens;
    int* batch_pos;
    int* batch_seq_ids;
    int* batch_output_indices;
    int* batch_sampled;
    BlockTable** batch_block_tables;
    int last_batch_count;
    int last_output_count;
    /* the pass in flight (two-halves mode): which sequence each output row belongs to and where
     * its token goes in that sequence's history (-1: nowhere) */
    int pend_active;
    long pend_ticket;
    int pend_outputs;
    int pend_prefill;      /* the pass in flight carries prompt tokens (its time counts as prefill time) */
    int* pend_seq;
    int* pend_hist_pos;
    int* row_of_seq;        /* per sequence: its output row in the pass in flight, or -1 */
    int* batch_token_src;   /* per token of the current batch: -1, or the row of the pass in flight it continues */
    int* collected;         /* scratch for collected token ids */
    /* allocation trace */
    int* trace;
    int trace_len, trace_cap;
    NvllmEngineStats stats;
};

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void trace_push(NvllmEngine* e, int seq, int logical, int physical) {
    if (e->trace_len == e->trace_cap) {
        e->trace_cap = e->trace_cap ? e->trace_cap * 2 : 1024;
        e->trace = (int*)realloc(e->trace, (size_t)e->trace_cap * 3 * sizeof(int));
    }
    int* t = e->trace + 3 * (size_t)e->trace_len++;
    t[0] = seq; t[1] = logical; t[2] = physical;
}

NvllmEngine* nvllm_engine_create(const char* model_path, int n_seqs, const NvllmEngineOptions* opt) {
    if (n_seqs <= 0 || !model_path) return NULL;
    NvllmEngine* e = (NvllmEngine*)calloc(1, sizeof(NvllmEngine));
    if (opt) e->opt = *opt;
    /* clamps of main.c:121-124 */
    if (e->opt.temperature < 0.0f) e->opt.temperature = 0.0f;
    if (e->opt.topp < 0.0f || 1.0f < e->opt.topp) e->opt.topp = 0.9f;
    if (e->opt.chunk_size <= 0) e->opt.chunk_size = INT_MAX;
    if (e->opt.block_size <= 0) e->opt.block_size = 16;
    if (e->opt.max_steps <= 0) e->opt.max_steps = 500;

    g_paged_mode = 1;       /* main.c:114-118: paged mode is always on */
    g_visualize_paged = 1;
    init_ops();
    load_model(&e->weights, &e->config, model_path);
    if (e->opt.max_seq_len > 0) e->config.max_seq_len = e->opt.max_seq_len;
    if (e->opt.raise_to_1024 && e->config.max_seq_len < 1024) e->config.max_seq_len = 1024; /* main.c:140-143 */
    malloc_run_state(&e->state, &e->config);
    e->capacity = e->config.max_seq_len;

    e->n_seqs = n_seqs;
    e->seqs = (Sequence*)calloc((size_t)n_seqs, sizeof(Sequence));
    e->history_cap = (int*)calloc((size_t)n_seqs, sizeof(int));
    const int bs = e->opt.block_size;
    e->max_blocks_per_seq = (e->config.max_seq_len + bs - 1) / bs;
    int pool_blocks = e->opt.num_blocks;
    if (pool_blocks <= 0) pool_blocks = n_seqs * e->max_blocks_per_seq; /* main.c:189-193 */
    init_kv_cache_manager(&g_kv_manager, bs, pool_blocks, e->config.n_layers, e->config.n_kv_heads,
                          e->config.head_dim);
    for (int i = 0; i < n_seqs; i++) {
        Sequence* s = &e->seqs[i];
        s->id = i;
        s->status = SEQ_FINISHED; /* until a request is set */
        s->table.block_indices = (int*)malloc((size_t)e->max_blocks_per_seq * sizeof(int));
        s->table.num_blocks = 0;
    }
    build_sampler(&e->sampler, e->config.vocab_size, e->opt.temperature, e->opt.topp, e->opt.seed);

    e->batch_tokens = (int*)malloc((size_t)e->capacity * sizeof(int));
    e->batch_pos = (int*)malloc((size_t)e->capacity * sizeof(int));
    e->batch_seq_ids = (int*)malloc((size_t)e->capacity * sizeof(int));
    e->batch_output_indices = (int*)malloc((size_t)n_seqs * sizeof(int));
    e->batch_sampled = (int*)malloc((size_t)n_seqs * sizeof(int));
    e->batch_block_tables = (BlockTable**)malloc((size_t)n_seqs * sizeof(BlockTable*));
    for (int i = 0; i < n_seqs; i++) e->batch_block_tables[i] = &e->seqs[i].table;
    e->pend_seq = (int*)malloc((size_t)n_seqs * sizeof(int));
    e->pend_hist_pos = (int*)malloc((size_t)n_seqs * sizeof(int));
    e->row_of_seq = (int*)malloc((size_t)n_seqs * sizeof(int));
    e->collected = (int*)malloc((size_t)n_seqs * sizeof(int));
    e->batch_token_src = (int*)malloc((size_t)e->capacity * sizeof(int));
    for (int i = 0; i < n_seqs; i++) e->row_of_seq[i] = -1;
    return e;
}

/* Collect the pass in flight, if any: its tokens go to the histories, become the sequences'
 * current tokens and complete the record of the last batch. */
static void engine_flush(NvllmEngine* e) {
    if (!e->pend_active) return;
    /* the wait for the pass belongs to the forward time of ITS class (prefill or decode), whichever step
     * of the driver happens to collect it */
    const double c0 = now_seconds();
    const int n = nvllm_forward_collect(e->pend_ticket, e->collected, e->n_seqs);
    const double waited = now_seconds() - c0;
    e->stats.forward_seconds += waited;
    if (e->pend_prefill) e->stats.prefill_step_seconds += waited;
    else e->stats.decode_step_seconds += waited;
    const int host_rows = nvllm_get_logits_mode() != NVLLM_LOGITS_ARGMAX;
    const RunState* st = e->pend_state ? &e->state2 : &e->state;
    e->seen_state = e->pend_state;
    const double t0 = now_seconds();
    for (int k = 0; k < n && k < e->pend_outputs; k++) {
        Sequence* s = &e->seqs[e->pend_seq[k]];
        int next = e->collected[k];
        if (host_rows) {
            /* the token is the sampler's, from the row on the host (main.c:377); the pass after this
             * one was fed the GPU's arg-max of the same row, which follows the same rule */
            next = sample(&e->sampler, st->logits + (size_t)k * e->config.vocab_size);
            if (next != e->collected[k]) {
                fprintf(stderr, "nvllm engine: host and device arg-max disagree on output %d (%d vs %d)\n", k, next, e->collected[k]);
                exit(1);
            }
        }
        if (e->pend_hist_pos[k] >= 0) s->output_history[e->pend_hist_pos[k]] = next;
        s->current_token = next;
        e->batch_sampled[k] = next;
        e->row_of_seq[e->pend_seq[k]] = -1;
    }
    if (host_rows) e->stats.sample_seconds += now_seconds() - t0;
    e->pend_active = 0;
}

void nvllm_engine_destroy(NvllmEngine* e) {
    if (!e) return;
    engine_flush(e);
    free(e->pend_seq);
    free(e->pend_hist_pos);
    free(e->row_of_seq);
    free(e->collected);
    free(e->batch_token_src);
    for (int i = 0; i < e->n_seqs; i++) {
        free(e->seqs[i].prompt_tokens);
        free(e->seqs[i].output_history);
        free(e->seqs[i].table.block_indices);
    }
    free(e->seqs);
    free(e->history_cap);
    free(e->batch_tokens);
    free(e->batch_pos);
    free(e->batch_seq_ids);
    free(e->batch_output_indices);
    free(e->batch_sampled);
    free(e->batch_block_tables);
    free(e->trace);
    free(e->sampler.probindex);
    free_run_state(&e->state);
    if (e->state2_ready) free_run_state(&e->state2);
    free_ops();
    free_kv_cache_manager(&g_kv_manager);
    if (nvllm_free_model) nvllm_free_model(&e->weights);
    else free(e->weights.layers);
    free(e);
}

int nvllm_engine_set_request(NvllmEngine* e, int seq, const int* prompt, int n_prompt, int n_new,
                             int arrival_step) {
    if (!e || seq < 0 || seq >= e->n_seqs || n_prompt <= 0 || n_new < 0 || !prompt) return -1;
    if (n_prompt + n_new > e->config.max_seq_len) return -1; /* block table would overflow */
    Sequence* s = &e->seqs[seq];
    free(s->prompt_tokens);
    free(s->output_history);
    s->prompt_tokens = (int*)malloc((size_t)n_prompt * sizeof(int));
    memcpy(s->prompt_tokens, prompt, (size_t)n_prompt * sizeof(int));
    s->num_prompt_tokens = n_prompt;
    e->history_cap[seq] = n_prompt + n_new + 1;
    s->output_history = (int*)calloc((size_t)e->history_cap[seq], sizeof(int));
    memcpy(s->output_history, prompt, (size_t)n_prompt * sizeof(int)); /* main.c:228-230 */
    s->active = 0;                        /* main.c:168-174 */
    s->status = SEQ_WAITING;
    s->arrival_step = arrival_step;
    s->pos = 0;
    s->current_token = prompt[0];         /* main.c:226 */
    s->seq_len = n_prompt + n_new;        /* main.c:231 */
    s->table.num_blocks = 0;
    e->finished = 0;
    return 0;
}

int nvllm_engine_skip_prefill(NvllmEngine* e, int seq, int upto) {
    if (!e || seq < 0 || seq >= e->n_seqs) return -1;
    Sequence* s = &e->seqs[seq];
    if (s->status != SEQ_WAITING || s->pos != 0 || !s->prompt_tokens || upto < 0) return -1;
    if (upto > s->num_prompt_tokens - 1) upto = s->num_prompt_tokens - 1;
    const int bs = g_kv_manager.block_size;
    for (int p = 0; p < upto; p++) {
        if (p % bs == 0) {                                  /* main.c:322-335 */
            const int blk = alloc_block(&g_kv_manager);
            if (blk == -1) return -1;
            s->table.block_indices[p / bs] = blk;
            s->table.num_blocks++;
            trace_push(e, seq, p / bs, blk);
        }
    }
    s->pos = upto;
    s->current_token = s->prompt_tokens[upto];
    return upto;
}

static void release_blocks(NvllmEngine* e, Sequence* s) {
    /* extension: hand a finished sequence's blocks back, last allocated first */
    for (int b = s->table.num_blocks - 1; b >= 0; b--) free_block(&g_kv_manager, s->table.block_indices[b]);
    s->table.num_blocks = 0;
}

int nvllm_engine_step(NvllmEngine* e) {
    if (!e || e->finished) return 0;
    /* two-halves mode: greedy, and a backend that offers it */
    const int two_halves = nvllm_forward_enqueue && nvllm_forward_collect && nvllm_get_logits_mode &&
                           e->sampler.temperature == 0.0f && !getenv("NVLLM_ENGINE_STEPWISE");
    if (!two_halves) engine_flush(e);
    int any_src = 0;
    int all_finished = 1;
    int batch_count = 0, output_count = 0;
    int any_prefill = 0, prefill_tokens = 0, decode_tokens = 0;

    /* scheduler phase (main.c:269-356) */
    for (int i = 0; i < e->n_seqs; i++) {
        Sequence* s = &e->seqs[i];
        if (s->status == SEQ_FINISHED) continue;       /* main.c:284-287 (also: slot unused) */
        if (e->global_step < s->arrival_step) {        /* main.c:271-275 */
            all_finished = 0;
            continue;
        }
        if (s->status == SEQ_WAITING) {                /* main.c:278-282 */
            s->active = 1;
            s->status = SEQ_PREFILLING;
        }
        all_finished = 0;

        const int is_prefill = (s->pos < s->num_prompt_tokens - 1); /* main.c:292 */
        const int start_pos = s->pos;
        int n_tokens;
        const int* token_ptr;
        if (is_prefill) {
            s->status = SEQ_PREFILLING;
            const int remaining = s->num_prompt_tokens - start_pos;
            n_tokens = remaining > e->opt.chunk_size ? e->opt.chunk_size : remaining;
            token_ptr = s->prompt_tokens + start_pos;
        } else {
            s->status = SEQ_DECODING;
            n_tokens = 1;
            token_ptr = &s->current_token;
        }
        if (batch_count + n_tokens > e->capacity) continue; /* main.c:315-318 */

        const int bs = g_kv_manager.block_size;             /* main.c:322-335 */
        for (int t = 0; t < n_tokens; t++) {
            const int p = start_pos + t;
            if (p % bs == 0) {
                const int blk = alloc_block(&g_kv_manager);
                if (blk == -1) {
                    printf("Error: Out of KV Cache blocks!\n");
                    exit(1);
                }
                s->table.block_indices[p / bs] = blk;
                s->table.num_blocks++;
                trace_push(e, i, p / bs, blk);
            }
        }
        for (int t = 0; t < n_tokens; t++) {                /* main.c:338-342 */
            e->batch_tokens[batch_count + t] = token_ptr[t];
            e->batch_pos[batch_count + t] = start_pos + t;
            e->batch_seq_ids[batch_count + t] = i;
            e->batch_token_src[batch_count + t] = -1;
        }
        if (!is_prefill && e->pend_active && e->row_of_seq[i] >= 0) {
            /* current_token is the sample of the pass in flight: the GPU supplies it */
            e->batch_token_src[batch_count] = e->row_of_seq[i];
            any_src = 1;
        }
        e->batch_output_indices[output_count++] = batch_count + n_tokens - 1; /* main.c:344 */
        batch_count += n_tokens;
        s->pos += n_tokens;
        if (is_prefill) {
            s->current_token = token_ptr[n_tokens - 1];     /* main.c:353-355 */
            any_prefill = 1;
            prefill_tokens += n_tokens;
        } else {
            decode_tokens += 1;
        }
    }

    if (batch_count == 0 && !all_finished) {                /* main.c:358-362 */
        e->global_step++;
        return 1;
    }
    if (all_finished) {                                     /* main.c:364 */
        engine_flush(e);
        e->finished = 1;
        return 0;
    }

    if (two_halves) {
        const double t0 = now_seconds();
        /* the two passes in flight alternate between two RunStates (their host logits rows) */
        if (e->flip && !e->state2_ready) {
            malloc_run_state(&e->state2, &e->config);
            e->state2_ready = 1;
        }
        RunState* st = e->flip ? &e->state2 : &e->state;
        const long ticket = nvllm_forward_enqueue(e->batch_tokens, batch_count, e->batch_pos, e->batch_seq_ids,
                                                  e->batch_output_indices, output_count, any_src ? e->batch_token_src : NULL,
                                                  &e->config, st, &e->weights, e->batch_block_tables);
        if (ticket < 0) {
            fprintf(stderr, "nvllm engine: the backend refused to enqueue the pass\n");
            exit(1);
        }
        const double t1 = now_seconds();   /* enqueue only: staging the batch and launching the kernels */
        /* the pass before this one (its wait and its sampling are accounted for in engine_flush): its tokens
         * also complete this batch's record */
        if (e->pend_active) {
            engine_flush(e);
            for (int t = 0; t < batch_count; t++)
                if (e->batch_token_src[t] >= 0) e->batch_tokens[t] = e->collected[e->batch_token_src[t]];
        }
        /* bookkeeping that does not need the token values (main.c:395-409) */
        for (int k = 0; k < output_count; k++) {
            const int sid = e->batch_seq_ids[e->batch_output_indices[k]];
            Sequence* s = &e->seqs[sid];
            e->pend_seq[k] = sid;
            e->pend_hist_pos[k] = (s->pos >= s->num_prompt_tokens && s->pos < s->seq_len) ? s->pos : -1;
            e->row_of_seq[sid] = k;
            e->batch_sampled[k] = -1;   /* filled when the pass is collected */
            if (s->pos >= s->seq_len) {
                s->active = 0;
                s->status = SEQ_FINISHED;
                if (e->opt.recycle_blocks) release_blocks(e, s);
            }
        }
        e->pend_active = 1;
        e->pend_prefill = any_prefill;
        e->pend_state = e->flip;
        e->flip ^= 1;
        e->pend_ticket = ticket;
        e->pend_outputs = output_count;
        e->last_batch_count = batch_count;
        e->last_output_count = output_count;
        e->stats.steps += 1;
        e->stats.tokens_forwarded += batch_count;
        e->stats.prefill_tokens += prefill_tokens;
        e->stats.decode_tokens += decode_tokens;
        e->stats.outputs += output_count;
        e->stats.forward_seconds += t1 - t0;
        if (any_prefill) {
            e->stats.prefill_step_seconds += t1 - t0;
            e->stats.prefill_step_tokens += batch_count;
        } else {
            e->stats.decode_step_seconds += t1 - t0;
            e->stats.decode_step_tokens += batch_count;
        }
        e->global_step++;
        if (e->global_step > e->opt.max_steps) {            /* main.c:416-418 */
            engine_flush(e);
            e->finished = 1;
            return 0;
        }
        return 1;
    }

    /* forward pass (main.c:367-369) */
    e->seen_state = 0;
    const double t0 = now_seconds();
    transformer_batch(e->batch_tokens, batch_count, e->batch_pos, e->batch_seq_ids,
                      e->batch_output_indices, output_count, &e->config, &e->state, &e->weights,
                      e->batch_block_tables);
    const double t1 = now_seconds();

    /* sampling and bookkeeping (main.c:372-410) */
    for (int k = 0; k < output_count; k++) {
        const int sid = e->batch_seq_ids[e->batch_output_indices[k]];
        Sequence* s = &e->seqs[sid];
        const int next = sample(&e->sampler, e->state.logits + (size_t)k * e->config.vocab_size);
        e->batch_sampled[k] = next;
        if (s->pos >= s->num_prompt_tokens && s->pos < s->seq_len) s->output_history[s->pos] = next;
        s->current_token = next;
        if (s->pos >= s->seq_len) {
            s->active = 0;
            s->status = SEQ_FINISHED;
            if (e->opt.recycle_blocks) release_blocks(e, s);
        }
    }
    const double t2 = now_seconds();

    e->last_batch_count = batch_count;
    e->last_output_count = output_count;
    e->stats.steps += 1;
    e->stats.tokens_forwarded += batch_count;
    e->stats.prefill_tokens += prefill_tokens;
    e->stats.decode_tokens += decode_tokens;
    e->stats.outputs += output_count;
    e->stats.forward_seconds += t1 - t0;
    e->stats.sample_seconds += t2 - t1;
    if (any_prefill) {
        e->stats.prefill_step_seconds += t1 - t0;
        e->stats.prefill_step_tokens += batch_count;
    } else {
        e->stats.decode_step_seconds += t1 - t0;
        e->stats.decode_step_tokens += batch_count;
    }

    e->global_step++;
    if (e->global_step > e->opt.max_steps) {                /* main.c:416-418 */
        e->finished = 1;
        return 0;
    }
    return 1;
}

int nvllm_engine_run(NvllmEngine* e, int n) {
    int ran = 0;
    while (ran < n) {
        const int more = nvllm_engine_step(e);
        ran++;
        if (!more) break;
    }
    return ran;
}

int nvllm_engine_global_step(const NvllmEngine* e) { return e->global_step; }
void nvllm_engine_get_stats(const NvllmEngine* e, NvllmEngineStats* out) { *out = e->stats; }
void nvllm_engine_reset_stats(NvllmEngine* e) { memset(&e->stats, 0, sizeof e->stats); }
const Config* nvllm_engine_config(const NvllmEngine* e) { return &e->config; }

int nvllm_engine_seq_state(const NvllmEngine* e, int seq, int* status, int* current_token,
                           int* n_prompt, int* seq_len) {
    if (seq < 0 || seq >= e->n_seqs) return -1;
    if (current_token) engine_flush((NvllmEngine*)e);
    const Sequence* s = &e->seqs[seq];
    if (status) *status = (int)s->status;
    if (current_token) *current_token = s->current_token;
    if (n_prompt) *n_prompt = s->num_prompt_tokens;
    if (seq_len) *seq_len = s->seq_len;
    return s->pos;
}

int nvllm_engine_history(const NvllmEngine* e, int seq, int* out, int cap) {
    if (seq < 0 || seq >= e->n_seqs) return -1;
    engine_flush((NvllmEngine*)e);
    const Sequence* s = &e->seqs[seq];
    int filled = s->pos < s->seq_len ? s->pos : s->seq_len; /* main.c:430-431 */
    if (!s->output_history) return 0;
    const int n = filled < cap ? filled : cap;
    memcpy(out, s->output_history, (size_t)n * sizeof(int));
    return filled;
}

int nvllm_engine_block_table(const NvllmEngine* e, int seq, int* out, int cap) {
    if (seq < 0 || seq >= e->n_seqs) return -1;
    const BlockTable* bt = &e->seqs[seq].table;
    const int n = bt->num_blocks < cap ? bt->num_blocks : cap;
    memcpy(out, bt->block_indices, (size_t)n * sizeof(int));
    return bt->num_blocks;
}

int nvllm_engine_last_batch(const NvllmEngine* e, int* tokens, int* pos, int* seq_ids, int cap,
                            int* output_indices, int* sampled, int out_cap, int* num_outputs) {
    engine_flush((NvllmEngine*)e);
    const int n = e->last_batch_count < cap ? e->last_batch_count : cap;
    if (tokens) memcpy(tokens, e->batch_tokens, (size_t)n * sizeof(int));
    if (pos) memcpy(pos, e->batch_pos, (size_t)n * sizeof(int));
    if (seq_ids) memcpy(seq_ids, e->batch_seq_ids, (size_t)n * sizeof(int));
    const int m = e->last_output_count < out_cap ? e->last_output_count : out_cap;
    if (output_indices) memcpy(output_indices, e->batch_output_indices, (size_t)m * sizeof(int));
    if (sampled) memcpy(sampled, e->batch_sampled, (size_t)m * sizeof(int));
    if (num_outputs) *num_outputs = e->last_output_count;
    return e->last_batch_count;
}

const float* nvllm_engine_logits(const NvllmEngine* e) {
    engine_flush((NvllmEngine*)e);
    return e->seen_state ? e->state2.logits : e->state.logits;
}

int nvllm_engine_alloc_trace(const NvllmEngine* e, int* out, int cap_triples) {
    const int n = e->trace_len < cap_triples ? e->trace_len : cap_triples;
    if (out) memcpy(out, e->trace, (size_t)n * 3 * sizeof(int));
    return e->trace_len;
}
