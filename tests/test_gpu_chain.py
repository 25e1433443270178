"""All layers of a decode step as ONE launch (csrc/chain.hip) against the same step as one launch
per operator, and against the CPU oracle.  The one-launch form uses the tiles and the summation
order of the separate kernels, so its results are required to be the very same bits; what differs
is how tiles travel between workgroups (arrival counters, write-through stores, loads past the
vector cache) -- exactly what a stale or early read would corrupt.  Reference loop: src/model.c:358-394.
"""
import numpy as np
import pytest

from nvllm_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu

LOGITS_TOL = 2e-2   # max-abs bar against the oracle (fp16 weights, fp32 accumulation)


@pytest.fixture(scope="module")
def wide(tmp_path_factory):
    """TinyLlama widths (dim 2048, hidden 5632, 32 / 4 heads of 64) with three layers."""
    shape = synth.ModelShape.named("tinyllama-1.1b", max_seq_len=2048, n_layers=3)
    path = str(tmp_path_factory.mktemp("chain") / "wide3.bin")
    synth.write_model(path, shape, seed=5)
    return path, shape


@pytest.fixture(scope="module")
def narrow(tmp_path_factory):
    """TinyStories-42M widths (dim 512, hidden 1376, 8 / 8 heads of 64, 8 layers)."""
    shape = synth.ModelShape.named("stories42M", max_seq_len=1024)
    path = str(tmp_path_factory.mktemp("chain") / "narrow.bin")
    synth.write_model(path, shape, seed=6)
    return path, shape


class _Modes:
    """fp16 page pool, logits rows on the host, and the chain switch set per run."""

    def __init__(self, hip):
        self.hip = hip

    def __enter__(self):
        self.kv = self.hip.nvllm_get_kv_elem_bytes()
        self.chain = self.hip.nvllm_get_decode_chain()
        self.hip.nvllm_set_kv_elem_bytes(2)
        return self

    def run(self, chain, *args, **kw):
        self.hip.nvllm_set_decode_chain(1 if chain else 0)
        before = self.hip.nvllm_decode_chain_launches()
        out = util.run_engine(self.hip, *args, **kw)
        out["chained"] = self.hip.nvllm_decode_chain_launches() - before
        return out

    def __exit__(self, *exc):
        self.hip.nvllm_set_decode_chain(self.chain)
        self.hip.nvllm_set_kv_elem_bytes(self.kv)


def _same_bits(a, b):
    np.testing.assert_array_equal(a["trace"], b["trace"])
    assert len(a["steps"]) == len(b["steps"])
    for sa, sb in zip(a["steps"], b["steps"]):
        for key in ("tokens", "pos", "seq_ids", "output_indices", "sampled"):
            np.testing.assert_array_equal(sa[key], sb[key])
        if "logits" in sa:
            assert np.array_equal(sa["logits"], sb["logits"]), float(np.abs(sa["logits"] - sb["logits"]).max())
    for ha, hb in zip(a["histories"], b["histories"]):
        np.testing.assert_array_equal(ha, hb)
    if "kv" in a:
        for (ak, av), (bk, bv) in zip(a["kv"], b["kv"]):
            assert np.array_equal(ak, bk) and np.array_equal(av, bv)


def _close_to_oracle(g, o, tol):
    np.testing.assert_array_equal(g["trace"], o["trace"])
    worst = 0.0
    for sg, so in zip(g["steps"], o["steps"]):
        np.testing.assert_array_equal(sg["sampled"], so["sampled"])
        worst = max(worst, float(np.abs(sg["logits"] - so["logits"]).max()))
    for a, b in zip(g["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)
    assert worst < tol, worst


def test_64_decode_rows_one_launch_equals_separate_kernels_and_the_oracle(hip, oracle, wide):
    """64 sequences decode together (the benchmark's batch): every pure-decode step takes the
    one-launch form, and logits, tokens and pool rows are the bits of the separate kernels."""
    path, shape = wide
    prompts = synth.make_requests(64, 2, 3, shape.vocab_size, seed=31)
    kw = dict(n_new=8, collect_logits=True, kv_rows=True, chunk_size=512, temperature=0.0, max_steps=40)
    with _Modes(hip) as m:
        sep = m.run(False, path, prompts, **kw)
        one = m.run(True, path, prompts, **kw)
    assert sep["chained"] == 0
    decode_steps = sum(1 for s in one["steps"] if 49 <= len(s["tokens"]) <= 64)
    assert decode_steps >= 7 and one["chained"] == decode_steps, (one["chained"], decode_steps)
    _same_bits(one, sep)
    o = util.run_engine(oracle, path, prompts, **{k: v for k, v in kw.items() if k != "kv_rows"})
    _close_to_oracle(one, o, LOGITS_TOL / 10)


def test_one_launch_at_contexts_beyond_1000(hip, wide):
    """The attention phase over ~66 pages per item, with the first pages requested before the
    q/k/v projection has finished: 64 sequences continue at positions 1040-1056 with their prompts
    skipped (the earlier pool rows are whatever the pool holds: zeros), three layers deep.  Same
    bits as the separate kernels."""
    path, shape = wide
    prompts = synth.make_requests(64, 1060, 1060, shape.vocab_size, seed=32)

    def run(m, chain):
        hip.nvllm_set_decode_chain(1 if chain else 0)
        before = hip.nvllm_decode_chain_launches()
        e = capi.Engine(hip, path, 64, chunk_size=512, temperature=0.0, max_seq_len=2048, max_steps=100)
        for i, p in enumerate(prompts):
            e.set_request(i, p, 6, 0)
            e.skip_prefill(i, 1040 + (i % 17))
        rows = []
        while e.step():
            lb = e.last_batch()
            rows.append((lb["tokens"].copy(), lb["sampled"].copy(), e.logits(len(lb["output_indices"])).copy()))
        hist = [e.history(i) for i in range(64)]
        e.close()
        return rows, hist, hip.nvllm_decode_chain_launches() - before

    with _Modes(hip) as m:
        sep_rows, sep_hist, n0 = run(m, False)
        one_rows, one_hist, n1 = run(m, True)
    assert n0 == 0 and n1 >= 5
    assert len(sep_rows) == len(one_rows)
    for (ta, sa, la), (tb, sb, lb) in zip(one_rows, sep_rows):
        np.testing.assert_array_equal(ta, tb)
        np.testing.assert_array_equal(sa, sb)
        assert np.array_equal(la, lb), float(np.abs(la - lb).max())
    for a, b in zip(one_hist, sep_hist):
        np.testing.assert_array_equal(a, b)


def test_narrow_model_32_rows_one_launch(hip, oracle, narrow):
    """The 42M shape at batch 32 (its own tiling of the chain: 16-token groups, a K split in gate/up)."""
    path, shape = narrow
    prompts = synth.make_requests(32, 3, 9, shape.vocab_size, seed=33)
    kw = dict(n_new=12, collect_logits=True, kv_rows=True, chunk_size=64, temperature=0.0, max_steps=60)
    with _Modes(hip) as m:
        sep = m.run(False, path, prompts, **kw)
        one = m.run(True, path, prompts, **kw)
    assert one["chained"] >= 10, one["chained"]
    _same_bits(one, sep)
    o = util.run_engine(oracle, path, prompts, **{k: v for k, v in kw.items() if k != "kv_rows"})
    _close_to_oracle(one, o, LOGITS_TOL / 10)


def test_prompt_chunks_next_to_decode_rows_in_one_launch(hip, narrow):
    """Steps that mix prompt chunks (several tokens of one sequence, whose attention reads keys the
    same launch wrote for the chunk's earlier tokens) with decode rows, staggered arrivals.  Steps of
    17..32 tokens have a tiling in the chain at this shape; the others run as separate kernels."""
    path, shape = narrow
    n = 24
    prompts = synth.make_requests(n, 5, 30, shape.vocab_size, seed=34)
    kw = dict(n_new=10, arrivals=[i // 2 for i in range(n)], collect_logits=True, kv_rows=True, chunk_size=8,
              temperature=0.0, max_steps=200)
    with _Modes(hip) as m:
        sep = m.run(False, path, prompts, **kw)
        one = m.run(True, path, prompts, **kw)
    fits = [s for s in one["steps"] if 17 <= len(s["tokens"]) <= 32]
    mixed = [s for s in fits if len(s["tokens"]) > len(s["output_indices"])]
    assert len(mixed) >= 3 and one["chained"] == len(fits), (len(mixed), len(fits), one["chained"])
    _same_bits(one, sep)


def test_two_halves_driver_over_the_one_launch_step(hip, wide):
    """Greedy arg-max hand-off with the next pass enqueued before the previous one is collected (the
    benchmark's loop): 40 steps of 64 rows, tokens identical with and without the one-launch form."""
    path, shape = wide
    prompts = synth.make_requests(64, 2, 2, shape.vocab_size, seed=35)
    kw = dict(n_new=40, chunk_size=512, temperature=0.0, max_steps=100)
    with _Modes(hip) as m:
        hip.nvllm_set_logits_mode(1)
        try:
            sep = m.run(False, path, prompts, **kw)
            one = m.run(True, path, prompts, **kw)
        finally:
            hip.nvllm_set_logits_mode(0)
    assert one["chained"] >= 38
    for a, b in zip(one["histories"], sep["histories"]):
        np.testing.assert_array_equal(a, b)
