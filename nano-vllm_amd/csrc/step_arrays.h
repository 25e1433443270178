// step_arrays.h -- device copies of the host-side description of one ragged batch.
#pragma once
#include "nvllm_internal.h"

namespace nvllm {

struct StepArrays {
    int T, B;
    int n_seqs;
    int max_pos;      // largest position in the batch
    int write_limit;  // tokens [0, write_limit) get their K/V stored (see attention.c:159-162)
    bool table_short; // some token's logical block lies beyond its sequence's table
    int* d_tokens;    // [T]        (null when not supplied)
    int* d_tok_src;   // [T]        (null unless supplied) -1, or the output row of the previous pass whose arg-max is the token
    int* d_pos;       // [T]
    int* d_seq;       // [T]
    int* d_out_idx;   // [B]
    int* d_tbl_off;   // [n_seqs]   start of each sequence's table inside d_tables
    int* d_tables;    // concatenated logical->physical block maps
    int* d_kv_row;    // [T] pool row (block * block_size + slot) where the token's K/V go, -1: not stored
    int* d_runs;      // [n_runs][4] (first token, count, last position, table offset): attention work items
    int n_runs;
    long kv_positions;
    int runs_in_tb[4];       // runs that touch each of the first four 16-token blocks
    bool one_token_per_seq;  // no sequence contributes more than one token (a pure decode step)
};

// Copies the arrays into the pinned ring, lays the block tables of the sequences
// that occur in the batch end to end, and enqueues ONE host-to-HBM copy.
// `tables` (indexed by seq id) or `single` (one table, positions pos0 + t) may be
// null for operators that do not touch the pool.  attn_ct > 0 also cuts the batch into
// attention work items: runs of at most attn_ct tokens of one sequence at
// consecutive positions.
StepArrays stage_step(const int* tokens, const int* pos_arr, const int* seq_ids, int T,
                      const int* output_indices, int B, BlockTable** tables, const BlockTable* single,
                      int pos0, int block_size, int attn_ct = 0, const int* token_src = nullptr);

KvPoolView pool_view(const KVCacheManager* mgr, int n_kv_heads, int head_dim);

}  // namespace nvllm
