"""GPU parity of the forward pass (transformer_batch behind the N-sequence driver)
against the CPU oracle on the same model file and requests."""
import numpy as np
import pytest

from nvllm_amd import capi, synth
from tests import checkers, util

pytestmark = pytest.mark.gpu

LOGITS_TOL = 2e-2   # the bar stated for this path: max-abs, fp16 weights, fp32 accumulation
TIGHT_TOL = 2e-4    # what this implementation is expected to achieve


def _compare_runs(g, o, check_logits=True):
    np.testing.assert_array_equal(g["trace"], o["trace"])            # allocator: bit-exact
    for a, b in zip(g["tables"], o["tables"]):
        np.testing.assert_array_equal(a, b)
    assert len(g["steps"]) == len(o["steps"])
    worst = 0.0
    for sg, so in zip(g["steps"], o["steps"]):
        for key in ("tokens", "pos", "seq_ids", "output_indices"):
            np.testing.assert_array_equal(sg[key], so[key])          # same ragged batches
        if check_logits:
            worst = max(worst, float(np.abs(sg["logits"] - so["logits"]).max()))
        np.testing.assert_array_equal(sg["sampled"], so["sampled"])  # greedy tokens identical
    for a, b in zip(g["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)
    return worst


@pytest.mark.parametrize("name,chunk", [("tiny", 16), ("tiny", 7), ("tiny-mha48", 1000)])
def test_mixed_prefill_decode_tiny(hip, oracle, tmp_models, name, chunk):
    path, shape = tmp_models(name)
    prompts = synth.make_requests(4, 3, 50, shape.vocab_size, seed=5)
    kw = dict(n_new=24, arrivals=[0, 2, 2, 9], collect_logits=True, chunk_size=chunk, temperature=0.0)
    o = util.run_engine(oracle, path, prompts, **kw)
    g = util.run_engine(hip, path, prompts, **kw)
    worst = _compare_runs(g, o)
    assert worst < TIGHT_TOL < LOGITS_TOL, worst


def test_stories15m_shape_greedy_256(hip, oracle, tmp_models):
    """head_dim 48, MHA, vocab 32000: 3 sequences, 256 greedy steps each."""
    path, shape = tmp_models("stories15M", max_seq_len=512)
    prompts = synth.make_requests(3, 20, 90, shape.vocab_size, seed=9)
    kw = dict(n_new=256, arrivals=[0, 0, 3], collect_logits=True, chunk_size=64, temperature=0.0, max_steps=2000)
    o = util.run_engine(oracle, path, prompts, **kw)
    g = util.run_engine(hip, path, prompts, **kw)
    worst = _compare_runs(g, o)
    assert worst < TIGHT_TOL, worst


def test_fp16_kv_pool_keeps_greedy_parity(hip, oracle, tmp_models):
    """The benchmark's configuration: KV rows stored as fp16 (the reference keeps fp32).
    Greedy tokens must stay identical over 256 steps and logits within the stated bar."""
    path, shape = tmp_models("stories15M", max_seq_len=512)
    prompts = synth.make_requests(3, 20, 90, shape.vocab_size, seed=11)
    kw = dict(n_new=256, arrivals=[0, 0, 3], collect_logits=True, chunk_size=64, temperature=0.0, max_steps=2000)
    o = util.run_engine(oracle, path, prompts, **kw)
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(2)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_kv_elem_bytes(before)
    worst = _compare_runs(g, o)
    assert worst < LOGITS_TOL / 10, worst


@pytest.mark.parametrize("temperature,topp", [(0.8, 0.9), (1.3, 0.5), (1.0, 1.0)])
def test_temperature_and_nucleus_sampling_follow_the_reference(hip, oracle, tmp_models, temperature, topp):
    """sample() with temperature > 0 (reference sampler.c:100-139): the GPU computes the
    probabilities and filters the nucleus candidates, the host sorts them and draws with
    libc rand() seeded like the reference, so the SAME tokens must come out as from the
    oracle run with the same seed.  topp = 1.0 is the plain-sampling branch."""
    path, shape = tmp_models("tiny")
    prompts = synth.make_requests(3, 4, 20, shape.vocab_size, seed=21)
    kw = dict(n_new=40, arrivals=[0, 1, 1], chunk_size=8, temperature=temperature, topp=topp, seed=1234)
    o = util.run_engine(oracle, path, prompts, **kw)      # srand(seed) happens in build_sampler: run one after the other
    g = util.run_engine(hip, path, prompts, **kw)
    for a, b in zip(g["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)
    hip.nvllm_set_logits_mode(1)                           # logits stay in HBM: same answer
    try:
        g2 = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_logits_mode(0)
    for a, b in zip(g2["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)


def test_sample_on_rows_outside_the_forward_pass(hip, oracle):
    """sample() handed an arbitrary host row (not one of the last forward pass)."""
    import ctypes as C
    V = 1000
    r = np.random.Generator(np.random.PCG64(8))
    rows = (r.standard_normal((12, V)) * 3).astype(np.float32)
    res = {}
    for lib in (oracle, hip):
        smp = capi.Sampler()
        lib.build_sampler(C.byref(smp), V, 0.7, 0.85, 99)
        out = []
        for k in range(rows.shape[0]):
            row = rows[k].copy()
            out.append(lib.sample(C.byref(smp), row.ctypes.data_as(C.POINTER(C.c_float))))
        smp.temperature = 0.0
        row = rows[3].copy()
        out.append(lib.sample(C.byref(smp), row.ctypes.data_as(C.POINTER(C.c_float))))
        res[lib.kind] = out
    assert res["hip"] == res["oracle"]
    assert res["hip"][-1] == int(np.argmax(rows[3]))


@pytest.mark.parametrize("kv_bytes", [4, 2])
def test_batch_of_64_sequences_matches_the_oracle(hip, oracle, tmp_models, kv_bytes):
    """The benchmark's batch width (64 decode rows per step: the 64-token code path of
    every projection and of attention) on the 15M shape, both pool formats."""
    path, shape = tmp_models("stories15M", max_seq_len=256)
    prompts = synth.make_requests(64, 3, 30, shape.vocab_size, seed=13)
    kw = dict(n_new=12, arrivals=[(5 * i) % 4 for i in range(64)], collect_logits=True, chunk_size=32,
              temperature=0.0, max_steps=400)
    o = util.run_engine(oracle, path, prompts, **kw)
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(kv_bytes)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_kv_elem_bytes(before)
    worst = _compare_runs(g, o)
    assert worst < (TIGHT_TOL if kv_bytes == 4 else LOGITS_TOL / 10), worst


def _random_shapes(n, seed):
    """Random small Llama shapes inside what the reference and the library both accept:
    widths that are multiples of 32, head_dim in the supported set, GQA ratios 1..8,
    vocabularies that are not multiples of the kernel's tile width."""
    r = np.random.Generator(np.random.PCG64(seed))
    out = []
    while len(out) < n:
        hd = int(r.choice([16, 32, 48, 64, 96, 128]))
        kvh = int(r.choice([1, 2, 3, 4]))
        group = int(r.choice([1, 2, 4, 8]))
        heads = kvh * group
        dim = heads * hd
        if dim % 32 or dim > 768 or (kvh * hd) % 16:
            continue
        out.append(dict(dim=dim, hidden_dim=int(r.integers(2, 12)) * 32, n_layers=int(r.integers(1, 4)), n_heads=heads,
                        n_kv_heads=kvh, vocab_size=int(r.integers(200, 900)), max_seq_len=160))
    return out


@pytest.mark.parametrize("idx,shape_kw", list(enumerate(_random_shapes(10, 2024))))
def test_random_shapes_mixed_batches(hip, oracle, tmp_models, idx, shape_kw):
    """Whole runs (chunked prefill + decode, staggered arrivals) on randomly drawn model
    shapes, fp32 rows for even draws and fp16 pages for odd ones."""
    path, shape = tmp_models("tiny", seed=30 + idx, **shape_kw)
    n_seqs = 2 + idx % 5
    prompts = synth.make_requests(n_seqs, 2, 60, shape.vocab_size, seed=40 + idx)
    kw = dict(n_new=14, arrivals=[(3 * i) % 5 for i in range(n_seqs)], collect_logits=True,
              chunk_size=[5, 16, 33, 1000][idx % 4], temperature=0.0, max_steps=500)
    o = util.run_engine(oracle, path, prompts, **kw)
    kv_bytes = 2 if idx % 2 else 4
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(kv_bytes)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_kv_elem_bytes(before)
    worst = _compare_runs(g, o)
    assert worst < (TIGHT_TOL if kv_bytes == 4 else LOGITS_TOL / 10), (shape_kw, worst)


def test_argmax_only_mode_matches_host_logits_mode(hip, tmp_models):
    path, shape = tmp_models("tiny")
    prompts = synth.make_requests(3, 3, 30, shape.vocab_size, seed=6)
    kw = dict(n_new=16, chunk_size=8, temperature=0.0)
    a = util.run_engine(hip, path, prompts, **kw)
    hip.nvllm_set_logits_mode(1)
    try:
        b = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_logits_mode(0)
    for x, y in zip(a["histories"], b["histories"]):
        np.testing.assert_array_equal(x, y)


def test_matmul_accepts_load_model_handles(hip, oracle, tmp_models):
    """ops.h matmul with the `weight` pointers that load_model hands out (fp16 tiles
    in HBM, incl. the interleaved gate/up storage) against the oracle's fp32 tensors."""
    import ctypes as C
    path, shape = tmp_models("tiny")
    r = np.random.Generator(np.random.PCG64(3))
    res = {}
    for lib in (oracle, hip):
        w, cfg = capi.Weights(), capi.Config()
        lib.load_model(C.byref(w), C.byref(cfg), path.encode())
        D, F, V = cfg.dim, cfg.hidden_dim, cfg.vocab_size
        QD, KD = cfg.n_heads * cfg.head_dim, cfg.n_kv_heads * cfg.head_dim
        lw = w.layers[1]
        cases = [("wq", lw.wq, D, QD), ("wk", lw.wk, D, KD), ("wv", lw.wv, D, KD), ("wo", lw.wo, QD, D),
                 ("w_gate", lw.w_gate, D, F), ("w_up", lw.w_up, D, F), ("w_down", lw.w_down, F, D),
                 ("lm_head", w.lm_head, D, V)]
        out = {}
        for T in (1, 5, 40, 70, 512):
            rr = np.random.Generator(np.random.PCG64(100 + T))
            for name, handle, K, N in cases:
                x = rr.standard_normal((T, K)).astype(np.float32)
                dx, do = lib.to_device(x), lib.empty_device(T * N * 4)
                lib.matmul(do, dx, handle, T, K, N)
                out[(name, T)] = lib.to_host(do, (T, N))
                lib.free(dx)
                lib.free(do)
        res[lib.kind] = out
        if lib.kind == "hip":
            lib.nvllm_free_model(C.byref(w))
    for key, want in res["oracle"].items():
        got = res["hip"][key]
        # fp16 weights are exact in both; the activation enters as hi+lo fp16 (~2^-22 relative)
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-6 * np.abs(want).max() + 1e-7, err_msg=str(key))


def test_stories42m_shape_with_32_concurrent_requests(hip, oracle, tmp_models):
    """BASELINE.json configs[1]'s shape and concurrency (TinyStories-42M, 32 requests, continuous
    batching with staggered arrivals and block recycling), shortened so that the oracle stays cheap."""
    path, shape = tmp_models("stories42M", max_seq_len=256)
    prompts = synth.make_requests(32, 8, 48, shape.vocab_size, seed=19)
    kw = dict(n_new=[6 + (i % 5) * 4 for i in range(32)], arrivals=[(7 * i) % 6 for i in range(32)], collect_logits=True,
              chunk_size=64, temperature=0.0, max_steps=400, recycle_blocks=1)
    o = util.run_engine(oracle, path, prompts, **kw)
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(2)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_kv_elem_bytes(before)
    worst = _compare_runs(g, o)
    assert worst < LOGITS_TOL / 10, worst


@pytest.mark.parametrize("kv_bytes,shape_kw", [
    (2, dict(dim=256, hidden_dim=512, n_layers=3, n_heads=8, n_kv_heads=4, vocab_size=600, max_seq_len=1024)),
    (4, dict(dim=288, hidden_dim=768, n_layers=2, n_heads=6, n_kv_heads=6, vocab_size=1000, max_seq_len=768)),   # head_dim 48; widths that are not multiples of 128
])
def test_prefill_batches_of_hundreds_of_tokens(hip, oracle, tmp_models, kv_bytes, shape_kw):
    """Forward passes that carry several hundred tokens (whole prompts / large chunks of several
    sequences at once) take the projections' dense-contraction kernel: all four epilogues, ragged
    token and feature edges, decode rows mixed in."""
    path, shape = tmp_models("tiny", seed=61, **shape_kw)
    prompts = synth.make_requests(6, 90, 300, shape.vocab_size, seed=62)
    kw = dict(n_new=5, arrivals=[0, 0, 0, 1, 1, 3], collect_logits=True, chunk_size=256, temperature=0.0, max_steps=200)
    o = util.run_engine(oracle, path, prompts, **kw)
    assert max(len(s["tokens"]) for s in o["steps"]) >= 512
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(kv_bytes)
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        hip.nvllm_set_kv_elem_bytes(before)
    worst = _compare_runs(g, o)
    assert worst < (TIGHT_TOL if kv_bytes == 4 else LOGITS_TOL / 10), worst


def test_forward_pass_with_blocks_of_32_positions(hip, oracle, tmp_models):
    """The driver with block_size 32 (fp32 pool rows): allocation trace, tokens and logits as the oracle's."""
    path, shape = tmp_models("tiny")
    prompts = synth.make_requests(4, 5, 70, shape.vocab_size, seed=41)
    kw = dict(n_new=20, arrivals=[0, 0, 1, 3], collect_logits=True, chunk_size=16, block_size=32, temperature=0.0, max_steps=300)
    o = util.run_engine(oracle, path, prompts, **kw)
    g = util.run_engine(hip, path, prompts, **kw)
    worst = _compare_runs(g, o)
    assert worst < TIGHT_TOL, worst


def test_unmodified_reference_driver_runs_on_the_gpu(tmp_path):
    """The reference's own main() -- its scheduler loop, tokenizer, sampler, dashboards, compiled
    unmodified (oracle/_ref/nano_vllm_hip, built where /root/reference exists) and linked with
    libnvllm_hip.so -- against the reference's own CPU binary on the same model and tokenizer
    files: the whole transcript (every scheduler step, the dashboards, the generated text of both
    sequences) must be identical but for the wall-clock line (2-sequence demo, paged, chunked
    prefill, greedy)."""
    import os
    import subprocess
    ref_dir = checkers.REF_DIR
    exe_hip, exe_ref = os.path.join(ref_dir, "nano_vllm_hip"), os.path.join(ref_dir, "nano_vllm_ref")
    if not (os.path.exists(exe_hip) and os.path.exists(exe_ref)):
        pytest.skip("oracle/_ref demo binaries not built (they need /root/reference at build time)")
    shape = synth.ModelShape.named("stories15M", vocab_size=512, max_seq_len=256)
    synth.write_model(str(tmp_path / "model.bin"), shape, seed=33)
    synth.write_tokenizer(str(tmp_path / "tokenizer.bin"), shape.vocab_size)
    outs = {}
    for name, exe in (("ref", exe_ref), ("hip", exe_hip)):
        r = subprocess.run([exe, str(tmp_path / "model.bin"), "-t", "0", "-n", "40", "--paged", "--chunk-size", "16"],
                           cwd=str(tmp_path), capture_output=True, timeout=300)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        out = r.stdout
        assert b"=== Final Generated Sequences ===" in out, (name, out[-2000:])
        # the whole transcript -- loader report, scheduler log of every step, block dashboards, generated
        # text -- except the wall-clock line (src/main.c:443)
        outs[name] = [ln for ln in out.decode("utf-8", "replace").splitlines() if not ln.startswith("Time: ")]
    assert any(ln.startswith("[Seq 0]: ") for ln in outs["ref"]) and any(ln.startswith("[Seq 1]: ") for ln in outs["ref"])
    if outs["hip"] != outs["ref"]:
        import difflib
        diff = list(difflib.unified_diff(outs["ref"], outs["hip"], "reference", "hip", lineterm="", n=1))
        raise AssertionError("transcripts differ:\n" + "\n".join(diff[:60]))
