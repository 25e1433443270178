"""Host logic of the N-sequence driver (nano-vllm_amd/host/engine.c) on the CPU
oracle: batch construction, block allocation order and the scheduler quirks of
reference src/main.c:259-419."""
import numpy as np
import pytest

from nvllm_amd import capi, synth
from tests import util


def test_block_allocation_order_and_ragged_batches(oracle, tmp_models):
    path, shape = tmp_models("tiny")
    prompts = [np.arange(1, 1 + n, dtype=np.int32) for n in (40, 5, 17)]
    e = capi.Engine(oracle, path, 3, chunk_size=16, temperature=0.0)
    for i, p in enumerate(prompts):
        e.set_request(i, p, 4, arrival_step=[0, 1, 1][i])
    # step 0: only seq 0 has arrived; one 16-token chunk at positions 0..15, one new block (id 0)
    assert e.step() == 1
    b = e.last_batch()
    assert b["tokens"].tolist() == list(range(1, 17)) and b["pos"].tolist() == list(range(16))
    assert b["output_indices"].tolist() == [15]
    assert e.alloc_trace().tolist() == [[0, 0, 0]]
    # step 1: seq 0 chunk 2 (block 1), seq 1 whole prompt (block 2), seq 2 first chunk (block 3)
    e.step()
    b = e.last_batch()
    assert b["seq_ids"].tolist() == [0] * 16 + [1] * 5 + [2] * 16
    assert b["output_indices"].tolist() == [15, 20, 36]
    assert e.alloc_trace().tolist() == [[0, 0, 0], [0, 1, 1], [1, 0, 2], [2, 0, 3]]
    # seq 2 has 17 prompt tokens: after 16 it sits at pos == n_prompt-1, so it DECODES the
    # token sampled in the previous step instead of its last prompt token (main.c:292,402)
    sampled_prev = dict(zip([0, 1, 2], b["sampled"].tolist()))
    e.step()
    b = e.last_batch()
    idx2 = b["seq_ids"].tolist().index(2)
    assert b["pos"][idx2] == 16 and b["tokens"][idx2] == sampled_prev[2]
    e.run_all()
    for i, p in enumerate(prompts):
        h = e.history(i)
        assert len(h) == len(p) + 4 and h[:len(p)].tolist() == p.tolist()
    # blocks are never recycled by default (reference memory.c:64-70 has no caller)
    tr = e.alloc_trace()
    assert tr[:, 2].tolist() == list(range(len(tr)))
    e.close()


def test_capacity_check_defers_a_sequence(oracle, tmp_models):
    path, shape = tmp_models("tiny")          # max_seq_len 128 = batch capacity
    prompts = [np.full(100, 7, np.int32), np.full(100, 9, np.int32)]
    e = capi.Engine(oracle, path, 2, temperature=0.0)   # unlimited chunk
    for i, p in enumerate(prompts):
        e.set_request(i, p, 2, 0)
    e.step()
    b = e.last_batch()
    assert len(b["tokens"]) == 100 and set(b["seq_ids"].tolist()) == {0}   # seq 1 did not fit (main.c:315-318)
    e.step()
    b = e.last_batch()
    assert b["seq_ids"].tolist() == [0] + [1] * 100
    e.close()


def test_recycling_extension_reuses_blocks(oracle, tmp_models):
    path, shape = tmp_models("tiny")
    e = capi.Engine(oracle, path, 2, temperature=0.0, recycle_blocks=1, num_blocks=3)
    e.set_request(0, np.arange(1, 20, dtype=np.int32), 3, 0)    # 2 blocks, finishes early
    e.set_request(1, np.arange(1, 30, dtype=np.int32), 3, 6)    # arrives after, needs 2 blocks of a 3-block pool
    e.run_all()
    tr = e.alloc_trace()
    assert tr[tr[:, 0] == 0][:, 2].tolist() == [0, 1]
    assert tr[tr[:, 0] == 1][:, 2].tolist() == [0, 1]           # freed last-first, so popped 0 then 1 again
    assert len(e.history(1)) == 32
    e.close()


def test_sampler_rows_and_ties(oracle):
    import ctypes as C
    s = capi.Sampler()
    oracle.build_sampler(C.byref(s), 8, 0.0, 0.9, 1)
    row = np.array([0.5, 2.0, 2.0, -1.0, 2.0, 0.0, 1.9, 2.0], np.float32)
    assert oracle.sample(C.byref(s), row.ctypes.data) == 1      # strict '>' keeps the first maximum


def test_skip_prefill_starts_a_sequence_at_a_position(oracle, tmp_models):
    """The measurement aid behind the CPU baseline: blocks for the skipped positions are allocated in
    order, the sequence continues with a decode row at that position, and nothing else changes."""
    path, shape = tmp_models("tiny")
    e = capi.Engine(oracle, path, 2, chunk_size=16, temperature=0.0, max_seq_len=128)
    prompts = synth.make_requests(2, 40, 40, shape.vocab_size, seed=5)
    for i, p in enumerate(prompts):
        e.set_request(i, p, 3, 0)
    assert e.skip_prefill(0, 39) == 39
    assert e.skip_prefill(1, 1000) == 39                      # clamped to the last prompt token
    assert list(e.block_table(0)) == [0, 1, 2] and list(e.block_table(1)) == [3, 4, 5]
    e.step()
    b = e.last_batch()
    assert list(b["pos"]) == [39, 39] and list(b["tokens"]) == [prompts[0][39], prompts[1][39]]
    while e.step():
        pass
    assert all(len(e.history(i)) == 43 for i in range(2))
    e.step()
    with pytest.raises(ValueError):
        e.skip_prefill(0, 5)                                  # only before a sequence has started
    e.close()
