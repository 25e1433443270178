"""The CPU oracle against the reference's own sources compiled in place
(oracle/_ref/libref.so): same seeded inputs, bit-identical outputs.  Skipped where
the reference build is absent (it cannot travel; tests/golden/ then stands in)."""
import ctypes as C

import numpy as np
import pytest

from nvllm_amd import capi, synth
from tests import checkers, util


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


@pytest.mark.parametrize("T,K,N", [(1, 64, 48), (9, 288, 100), (64, 512, 64)])
def test_matmul_rms_swiglu(oracle, ref, T, K, N):
    r = rng(T + K)
    x = r.standard_normal((T, K)).astype(np.float32)
    w = (0.05 * r.standard_normal((N, K))).astype(np.float32)
    np.testing.assert_array_equal(util.op_matmul(oracle, x, w), util.op_matmul(ref, x, w))
    g = (1 + 0.1 * r.standard_normal(K)).astype(np.float32)
    np.testing.assert_array_equal(util.op_rms_norm(oracle, x, g), util.op_rms_norm(ref, x, g))
    np.testing.assert_array_equal(util.op_rms_norm(oracle, x, g, inplace=True), util.op_rms_norm(ref, x, g, inplace=True))
    u = r.standard_normal((T, K)).astype(np.float32)
    np.testing.assert_array_equal(util.op_swiglu(oracle, x * 8, u), util.op_swiglu(ref, x * 8, u))
    np.testing.assert_array_equal(util.op_softmax(oracle, x[0] * 5), util.op_softmax(ref, x[0] * 5))
    np.testing.assert_array_equal(util.op_accum(oracle, x, u), util.op_accum(ref, x, u))


@pytest.mark.parametrize("hd,H,KVH", [(64, 8, 1), (48, 6, 6), (128, 2, 2)])
def test_rope(oracle, ref, hd, H, KVH):
    r = rng(hd)
    pos = np.array([0, 1, 15, 16, 17, 511, 2047, 4095], np.int32)
    q = r.standard_normal((8, H * hd)).astype(np.float32)
    k = r.standard_normal((8, KVH * hd)).astype(np.float32)
    for a, b in zip(util.op_rope_batch(oracle, q, k, pos, 10000.0, hd, H, KVH),
                    util.op_rope_batch(ref, q, k, pos, 10000.0, hd, H, KVH)):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(util.op_rope(oracle, q, k, 100, 10000.0, hd, H, KVH),
                    util.op_rope(ref, q, k, 100, 10000.0, hd, H, KVH)):
        np.testing.assert_array_equal(a, b)


def test_allocator_offsets_kv_attention(oracle, ref):
    L, bs, KVH, hd, H, S = 2, 16, 2, 32, 8, 256
    pools = [util.Pool(lib, bs, 24, L, KVH, hd) for lib in (oracle, ref)]
    r = rng(77)
    ctx = [100, 17, 16, 1]
    perm = r.permutation(24)
    kfill = r.standard_normal(pools[0].elems()).astype(np.float32)
    vfill = r.standard_normal(pools[0].elems()).astype(np.float32)
    pos, sid = list(range(80, 100)), [0] * 20
    for s in range(1, 4):
        pos.append(ctx[s] - 1)
        sid.append(s)
    T = len(pos)
    k = r.standard_normal((T, KVH * hd)).astype(np.float32)
    v = r.standard_normal((T, KVH * hd)).astype(np.float32)
    q = r.standard_normal((T, H * hd)).astype(np.float32)
    outs = []
    for p, lib in zip(pools, (oracle, ref)):
        cur = 0
        for s, c in enumerate(ctx):
            p.new_table(S // bs)
            need = (c + bs - 1) // bs
            p.set_table(s, perm[cur:cur + need].tolist())
            cur += need
        p.fill(kfill, vfill)
        util.op_kv_write_batch(lib, p, k, v, 1, pos, sid, KVH, hd)
        att = util.op_paged_attention_batch(lib, p, q, 1, pos, sid, S, H, KVH, hd)
        offs = [lib.get_physical_offset(C.byref(p.mgr), 1, 5, 3, KVH, hd), lib.get_physical_offset(C.byref(p.mgr), 0, 23, 15, KVH, hd)]
        outs.append((p.dump("k"), p.dump("v"), att, offs))
        p.close()
    for a, b in zip(outs[0][:3], outs[1][:3]):
        np.testing.assert_array_equal(a, b)
    assert outs[0][3] == outs[1][3]


@pytest.mark.parametrize("name,chunk,dtype,shared,extra", [
    ("tiny", 16, "f16", False, {}), ("tiny", 5, "f32", True, {}), ("tiny-mha48", 1000, "f16", True, {}),
    ("tiny", 3, "f32", False, {}),
    ("tiny", 16, "f32", True, dict(rope_tail=True, exact16=False)),    # like a trained llama2.c file: values off the fp16 grid
    ("tiny", 7, "f32", False, dict(exact16=False))])
def test_forward_pass_and_scheduler(oracle, ref, tmp_models, name, chunk, dtype, shared, extra):
    path, shape = tmp_models(name, seed=4, dtype=dtype, shared=shared, **extra)
    prompts = synth.make_requests(4, 2, 45, shape.vocab_size, seed=8)
    kw = dict(n_new=18, arrivals=[0, 1, 1, 7], collect_logits=True, chunk_size=chunk, temperature=0.0)
    a = util.run_engine(oracle, path, prompts, **kw)
    b = util.run_engine(ref, path, prompts, **kw)
    np.testing.assert_array_equal(a["trace"], b["trace"])
    assert len(a["steps"]) == len(b["steps"])
    for sa, sb in zip(a["steps"], b["steps"]):
        for key in ("tokens", "pos", "seq_ids", "output_indices", "sampled", "logits"):
            np.testing.assert_array_equal(sa[key], sb[key])
    for x, y in zip(a["histories"], b["histories"]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("temperature,topp", [(0.8, 0.9), (1.3, 0.5), (1.0, 1.0), (0.5, 0.0), (2.0, 0.99)])
def test_sampler_with_temperature_matches_the_reference(oracle, ref, temperature, topp):
    """sample() beyond greedy (reference src/sampler.c:41-139): in-place temperature and softmax,
    libc rand() coin, plain sampling or the nucleus (cutoff, qsort, cumulative walk).  Same seed,
    same rows, one library after the other (both draw from the process-wide rand()): the tokens
    and the rows they leave behind must be identical."""
    V = 1500
    r = rng(int(temperature * 100 + topp * 10))
    rows = (r.standard_normal((40, V)) * 3).astype(np.float32)
    rows[5, :] = 0.0                     # a flat row: every probability equals the cutoff region
    rows[6, 17] = 40.0                   # a one-token nucleus
    rows[7, [3, 4]] = 12.0               # equal probabilities inside the nucleus
    res = {}
    for lib in (oracle, ref):
        smp = capi.Sampler()
        lib.build_sampler(C.byref(smp), V, temperature, topp, 4242)
        toks, left = [], []
        for k in range(rows.shape[0]):
            row = rows[k].copy()
            toks.append(lib.sample(C.byref(smp), row.ctypes.data_as(C.POINTER(C.c_float))))
            left.append(row)
        res[lib.kind] = (toks, np.stack(left))
    assert res["oracle"][0] == res["ref"][0]
    np.testing.assert_array_equal(res["oracle"][1], res["ref"][1])
    assert len(set(res["oracle"][0])) > 3


def test_allocation_trace_of_a_64_sequence_run(oracle, ref, tmp_models):
    """Integer bookkeeping at the benchmark's width: 64 sequences, staggered arrivals,
    chunked prefill, recycling off (the reference never frees).  The allocation trace
    (seq, logical block, physical block), the block tables and the ragged batches of
    every step must be bit-identical, and physical ids start at 0, 1, 2, ..."""
    path, shape = tmp_models("tiny", seed=6)
    prompts = synth.make_requests(64, 2, 40, shape.vocab_size, seed=17)
    arrivals = [(7 * i) % 11 for i in range(64)]
    kw = dict(n_new=20, arrivals=arrivals, chunk_size=16, temperature=0.0, max_steps=1000)
    a = util.run_engine(oracle, path, prompts, **kw)
    b = util.run_engine(ref, path, prompts, **kw)
    np.testing.assert_array_equal(a["trace"], b["trace"])
    assert a["trace"].shape[0] >= 64
    np.testing.assert_array_equal(a["trace"][:, 2], np.arange(a["trace"].shape[0]))   # LIFO stack pops 0,1,2,...
    for ta, tb in zip(a["tables"], b["tables"]):
        np.testing.assert_array_equal(ta, tb)
    assert len(a["steps"]) == len(b["steps"])
    for sa, sb in zip(a["steps"], b["steps"]):
        for key in ("tokens", "pos", "seq_ids", "output_indices", "sampled"):
            np.testing.assert_array_equal(sa[key], sb[key])


def test_two_sequence_demo_matches_reference_binary(oracle, tmp_path):
    """The driver restated in host/engine.c against the reference's UNMODIFIED
    main() (oracle/_ref/nano_vllm_ref): same model, a tokenizer whose pieces are
    invertible, greedy; the decoded transcript gives back the reference's token ids."""
    import os
    import re
    import subprocess
    exe = os.path.join(checkers.REF_DIR, "nano_vllm_ref")
    if not os.path.exists(exe):
        pytest.skip("reference demo binary not built")
    shape = synth.ModelShape.named("tiny", vocab_size=512, max_seq_len=64)
    synth.write_model(str(tmp_path / "model.bin"), shape, seed=21)
    synth.write_tokenizer(str(tmp_path / "tokenizer.bin"), shape.vocab_size)
    n_new, chunk = 9, 64
    out = subprocess.run([exe, str(tmp_path / "model.bin"), "-t", "0", "-n", str(n_new), "--paged", "--chunk-size", str(chunk)],
                         cwd=str(tmp_path), capture_output=True, timeout=300).stdout   # bytes: byte tokens print raw
    m0 = re.search(rb"Seq 0 \(Long\): (\d+) tokens", out)
    m1 = re.search(rb"Seq 1 \(Short\): (\d+) tokens", out)
    assert m0 and m1, out[-2000:]
    n0, n1 = int(m0.group(1)), int(m1.group(1))
    # what the reference generated: "[Seq i]: <decoded history>" up to the closing rule
    tail = out[out.index(b"=== Final Generated Sequences ==="):]
    f0 = tail.index(b"[Seq 0]: ") + 9
    f1 = tail.index(b"\n[Seq 1]: ")
    end = tail.index(b"\n=================================")
    gen_text = {0: tail[f0:f1], 1: tail[f1 + 10:end]}

    # rebuild the prompts' token ids with the same tokenizer rules: every printable
    # character is its own piece, ' ' merges are impossible (all scores are 0 and no
    # multi-character piece occurs in the prompts), BOS first, dummy-prefix space
    # (reference tokenizer.c:136-160)
    vocab = ["<unk>", "<s>", "</s>"] + [f"<0x{b:02X}>" for b in range(256)] + [" "] + [chr(c) for c in range(33, 127)]
    lookup = {p: i for i, p in enumerate(vocab)}
    long_prompt = re.search(r'strcpy\(long_prompt, "((?:[^"\\]|\\.)*)"\)', open("/root/reference/src/main.c").read()).group(1)
    long_prompt = long_prompt.replace('\\"', '"')
    prompts_txt = [long_prompt, "One day, a cat"]
    prompts = [np.array([1, lookup[" "]] + [lookup[ch] for ch in txt], np.int32) for txt in prompts_txt]
    assert [len(p) for p in prompts] == [n0, n1]

    e = capi.Engine(oracle, str(tmp_path / "model.bin"), 2, chunk_size=chunk, temperature=0.0, raise_to_1024=1)
    e.set_request(0, prompts[0], n_new, 0)
    e.set_request(1, prompts[1], n_new, 5)   # arrival steps {0, 5}: main.c:162
    e.run_all()
    inv = synth_pieces(shape.vocab_size)
    for i in range(2):
        ids = e.history(i)
        text = b"".join(inv[t] for t in ids)
        assert text == gen_text[i], (i, text[-80:], gen_text[i][-80:])
    e.close()


def synth_pieces(vocab_size):
    """decode_token() of the reference on the synthetic tokenizer (tokenizer.c:92-103)."""
    pieces = [b"<unk>", b"<s>", b"</s>"] + [bytes([b]) for b in range(256)] + [b" "] + [bytes([c]) for c in range(33, 127)]
    pieces[3] = b""   # '<0x00>' decodes to a NUL byte, which printf("%s") drops
    i = 0
    while len(pieces) < vocab_size:
        pieces.append(f"▁t{i}".encode("utf-8"))
        i += 1
    return pieces[:vocab_size]
