"""The benchmark's own shape (TinyLlama-1.1B: dim 2048, hidden 5632, 22 layers, 32/4 heads,
vocab 32000) against the CPU oracle: the work decompositions the projections and attention
pick for these sizes at 64 decode rows are exactly the ones the benchmark times.  The runs
are short because the oracle needs about 2 GFLOP per token on the host cores."""
import numpy as np
import pytest

from nvllm_amd import synth
from tests import util

pytestmark = pytest.mark.gpu

LOGITS_TOL = 2e-2   # the bar stated for this path (max-abs, fp16 weights, fp32 accumulation)


@pytest.fixture(scope="module")
def tinyllama(tmp_path_factory):
    shape = synth.ModelShape.named("tinyllama-1.1b", max_seq_len=2048)
    path = str(tmp_path_factory.mktemp("full") / "tinyllama.bin")
    synth.write_model(path, shape, seed=0)
    return path, shape


def _compare(g, o, tol):
    np.testing.assert_array_equal(g["trace"], o["trace"])
    worst = 0.0
    for sg, so in zip(g["steps"], o["steps"]):
        for key in ("tokens", "pos", "seq_ids", "output_indices", "sampled"):
            np.testing.assert_array_equal(sg[key], so[key])
        worst = max(worst, float(np.abs(sg["logits"] - so["logits"]).max()))
    for a, b in zip(g["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)
    assert worst < tol, worst
    return worst


def test_decode_batch_of_64_at_the_benchmark_shape(hip, oracle, tinyllama):
    """64 sequences: one 128-token prefill step, then pure-decode steps of 64 rows, fp16 page pool
    (the benchmark's configuration)."""
    path, shape = tinyllama
    prompts = synth.make_requests(64, 2, 2, shape.vocab_size, seed=11)
    kw = dict(n_new=10, collect_logits=True, chunk_size=512, temperature=0.0, max_steps=50)
    o = util.run_engine(oracle, path, prompts, **kw)
    assert sum(len(s["tokens"]) == 64 for s in o["steps"]) >= 9     # the 64-row decode steps are in there
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(2)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_kv_elem_bytes(before)
    _compare(g, o, LOGITS_TOL / 10)


def test_chunked_prefill_and_decode_at_the_benchmark_shape(hip, oracle, tinyllama):
    """Ragged batches (prefill chunks of 16 next to decode rows, staggered arrivals), fp32 row pool."""
    path, shape = tinyllama
    prompts = synth.make_requests(4, 5, 40, shape.vocab_size, seed=12)
    kw = dict(n_new=6, arrivals=[0, 0, 1, 3], collect_logits=True, chunk_size=16, temperature=0.0, max_steps=100)
    o = util.run_engine(oracle, path, prompts, **kw)
    g = util.run_engine(hip, path, prompts, **kw)
    _compare(g, o, LOGITS_TOL / 10)


def _kv_close(g, o, fp16_pool):
    """K/V rows left in the pool.  fp32 rows are copies of the projection's fp32 results (summation
    order away from the oracle's).  fp16 pages: the first layer's rows depend on the embedding and the
    weights alone, so they are the fp16 roundings of the oracle's values (the neighbouring fp16 number
    where the two fp32 results straddle a rounding boundary); deeper layers see attention over
    rounded K/V, which the fp32 oracle does not, and are held to the tolerance of the logits."""
    for (gk, gv), (ok, ov) in zip(g["kv"], o["kv"]):
        for a, b in ((gk, ok), (gv, ov)):
            assert a.shape == b.shape and a.shape[1] > 0
            scale = max(1.0, float(np.abs(b).max()))
            if fp16_pool:
                b16 = np.clip(b, -65504, 65504).astype(np.float16).astype(np.float32)
                # one fp16 step of the value, plus the fp32 summation-order error of the projection itself
                # (which is many fp16 steps of a value that happens to lie near zero)
                bound = np.maximum(np.abs(b16[0]), 2.0 ** -14) * 2.0 ** -10 + 4e-6 * scale
                worst0 = float((np.abs(a[0] - b16[0]) / bound).max())
                assert worst0 <= 1.0, worst0
                assert float(np.mean(a[0] != b16[0])) < 1e-2     # almost every half is the very same number
                worst = float(np.abs(a - b16).max())
                assert worst < LOGITS_TOL / 4 * scale, worst
            else:
                worst = float(np.abs(a - b).max())
                assert worst < 2e-5 * scale, worst


def test_256_greedy_steps_at_the_benchmark_shape(hip, oracle, tinyllama):
    """The acceptance run at full width: greedy tokens identical to the CPU path for 256 steps (four
    sequences with staggered prompts so that prefill chunks and decode rows share passes; fp16 page
    pool, arg-max hand-off as in the benchmark).  Reference loop: main.c:259-419."""
    path, shape = tinyllama
    prompts = synth.make_requests(4, 5, 40, shape.vocab_size, seed=23)
    kw = dict(n_new=256, arrivals=[0, 0, 2, 5], chunk_size=16, temperature=0.0, max_steps=400)
    o = util.run_engine(oracle, path, prompts, **kw)
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(2)
    hip.nvllm_set_logits_mode(1)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_logits_mode(0)
        hip.nvllm_set_kv_elem_bytes(before)
    assert all(len(h) == len(p) + 256 for h, p in zip(o["histories"], prompts))
    for a, b in zip(g["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(g["tables"], o["tables"]):
        np.testing.assert_array_equal(a, b)


def test_512_token_chunks_then_decode_beyond_1024_at_the_benchmark_shape(hip, oracle, tinyllama):
    """What the benchmark times, as a forward pass against the oracle: 1040-token prompts forwarded in
    512-token chunks (two sequences per pass: 1024 tokens through the dense-contraction projection
    kernel, both of its wave arrangements, and chunked-prefill attention over fp16 pages), a ragged
    tail pass, then decode rows at contexts beyond 1024.  Tokens, logits and pool contents."""
    path, shape = tinyllama
    prompts = synth.make_requests(2, 1040, 1040, shape.vocab_size, seed=21)
    kw = dict(n_new=5, collect_logits=True, kv_rows=True, chunk_size=512, temperature=0.0, max_steps=50)
    o = util.run_engine(oracle, path, prompts, **kw)
    assert [len(s["tokens"]) for s in o["steps"][:3]] == [1024, 1024, 32]
    assert max(int(s["pos"].max()) for s in o["steps"]) >= 1040
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(2)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_kv_elem_bytes(before)
    worst = _compare(g, o, LOGITS_TOL / 10)
    _kv_close(g, o, fp16_pool=True)
    print(f"max |dlogit| = {worst:.3e}")


def test_long_context_4k_forward_pass(hip, oracle, tmp_path_factory):
    """BASELINE.json configs[3]: TinyLlama widths with max_seq_len 4096 -- a 4096-row RoPE table, 256-entry
    block tables, 3,900-token prompts in 512-token chunks, then decode rows that attend over ~3.9 k keys.
    Two layers keep the oracle affordable; every kernel sees the full-size shapes."""
    shape = synth.ModelShape.named("tinyllama-1.1b", max_seq_len=4096, n_layers=2)
    path = str(tmp_path_factory.mktemp("ctx4k") / "tinyllama_2l_4k.bin")
    synth.write_model(path, shape, seed=3)
    prompts = synth.make_requests(2, 3890, 3900, shape.vocab_size, seed=22)
    kw = dict(n_new=4, collect_logits=True, kv_rows=True, chunk_size=512, temperature=0.0, max_steps=60)
    o = util.run_engine(oracle, path, prompts, **kw)
    assert max(int(s["pos"].max()) for s in o["steps"]) >= 3890
    assert all(len(t) >= 244 for t in o["tables"])               # block tables near their 256-entry capacity
    for kv_bytes in (2, 4):
        before = hip.nvllm_get_kv_elem_bytes()
        hip.nvllm_set_kv_elem_bytes(kv_bytes)
        try:
            g = util.run_engine(hip, path, prompts, **kw)
        finally:
            hip.nvllm_set_kv_elem_bytes(before)
        _compare(g, o, LOGITS_TOL / 10 if kv_bytes == 2 else 2e-4)
        _kv_close(g, o, fp16_pool=kv_bytes == 2)


def test_full_size_run_is_reproducible_and_mode_independent(hip, tinyllama):
    """At the benchmark's full width and length, where the oracle is out of reach: 64 sequences, 128-token
    prompts in 512-token passes, 256 greedy steps.  The kernels sum in a fixed order, so two runs must give
    the same tokens, and the driver's two-halves mode (tokens chained on the GPU) the same as its
    step-by-step loop; allocation is the plain 0, 1, 2, ... of the reference's stack."""
    import os
    from nvllm_amd import capi
    path, shape = tinyllama
    prompts = synth.make_requests(64, 128, 128, shape.vocab_size, seed=31)

    def run(stepwise):
        if stepwise:
            os.environ["NVLLM_ENGINE_STEPWISE"] = "1"
        try:
            e = capi.Engine(hip, path, 64, chunk_size=512, temperature=0.0, max_seq_len=2048, max_steps=1 << 20)
            for i, p in enumerate(prompts):
                e.set_request(i, p, 256, 0)
            steps = e.run_all()
            out = ([e.history(i) for i in range(64)], e.alloc_trace(), steps)
            e.close()
            return out
        finally:
            os.environ.pop("NVLLM_ENGINE_STEPWISE", None)

    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(2)
    hip.nvllm_set_logits_mode(1)
    try:
        a, b, c = run(False), run(False), run(True)
    finally:
        hip.nvllm_set_logits_mode(0)
        hip.nvllm_set_kv_elem_bytes(before)
    assert a[2] == b[2] == c[2] and a[2] >= 256
    np.testing.assert_array_equal(a[1][:, 2], np.arange(a[1].shape[0]))
    for other in (b, c):
        np.testing.assert_array_equal(a[1], other[1])
        for x, y in zip(a[0], other[0]):
            np.testing.assert_array_equal(x, y)
    assert all(len(h) == 128 + 256 for h in a[0])
    assert len({tuple(h[128:].tolist()) for h in a[0]}) > 32        # the sequences really generate different text
