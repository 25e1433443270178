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
    shape = synth.ModelShape.named("tinyllama-1.1b", max_seq_len=512)
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
