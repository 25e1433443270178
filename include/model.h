/*
 * model.h -- model loading, workspace management and the batched forward pass.
 * Same entry points as reference include/model.h:7-17.
 */
#ifndef MODEL_H
#define MODEL_H

#include "structs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Read a model file in the reference's format (7-int header, tensors grouped by
 * parameter type across layers, fp16 or fp32, shared or separate classifier,
 * detected from the file size) and place it in HBM.  Matrices are kept as fp16
 * in MFMA tile order, norm vectors as fp32.  An fp32 file takes that form only if
 * every matrix value is exactly representable in fp16 (nothing is rounded);
 * otherwise its tensors stay fp32 in HBM, one row-major buffer per tensor as in the
 * reference, and the forward pass runs on the fp32 operators -- exact weights,
 * slower.  Prints the reference's report of the header and the detected format.
 * Fills p (incl. head_dim, rope_theta) and w.  Fatal errors print to stderr and
 * exit(1) like the reference.  Reference: src/model.c:110-285. */
void load_model(Weights* w, Config* p, const char* checkpoint_path);

/* Allocate the activation workspace for up to p->max_seq_len tokens.
 * Reference: src/model.c:287-321. */
void malloc_run_state(RunState* s, Config* p);
void free_run_state(RunState* s);             /* reference src/model.c:323-336 */

/* Single-sequence entry point; like the reference it only reports that it is
 * retired and exits.  Reference: src/model.c:339-344. */
void transformer(int* tokens, int num_tokens, int pos, Config* p, RunState* s, Weights* w,
                 BlockTable* bt);

/* Forward pass of one ragged batch.  tokens/pos_arr/seq_ids have num_tokens
 * entries, output_indices has num_outputs entries (all host).  On return
 * s->logits[k*vocab_size ..] holds the fp32 logits of token output_indices[k]
 * and the KV pool of the global g_kv_manager holds K/V of every (seq,pos) in
 * the batch for all layers.  The call is synchronous.  The other RunState
 * buffers are scratch.  Reference: src/model.c:347-419. */
void transformer_batch(int* tokens, int num_tokens, int* pos_arr, int* seq_ids,
                       int* output_indices, int num_outputs, Config* p, RunState* s,
                       Weights* w, BlockTable** block_tables);

#ifdef __cplusplus
}
#endif

#endif /* MODEL_H */
