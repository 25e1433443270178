"""The consumer of the logits (reference src/sampler.c:84-140) on the GPU side: the device
arg-max with the reference's tie rule, whole rows and rank-style vocabulary slices, and
sample() reading host rows as the caller left them."""
import ctypes as C

import numpy as np
import pytest

from nvllm_amd import capi, synth

pytestmark = pytest.mark.gpu


def _first_max(rows):
    """sampler.c:86-97: strict '>' scan from index 0 (a NaN elsewhere never wins, a NaN at 0 is never displaced)."""
    return np.array([0 if np.isnan(r[0]) else int(np.flatnonzero(r == np.nanmax(r))[0]) for r in rows], np.int32)


def _device_argmax(hip, rows, slices):
    B, V = rows.shape
    d = hip.to_device(rows)
    out = (C.c_int * B)()
    assert hip.nvllm_argmax_rows(out, d, B, V, slices) == 0
    hip.free(d)
    return np.array(out[:], np.int32)


def _tie_rows(V, r):
    rows = (r.standard_normal((12, V)) * 2).astype(np.float32)
    top = np.float32(50.0)
    rows[0, [5, V - 1]] = top                       # tie between the ends of the row
    rows[1, [V - 1]] = top                          # maximum in the last element (tail of a vector load)
    rows[2, :] = np.float32(1.25)                   # every element ties: index 0
    rows[3, :] = -np.inf                            # the scan never moves: index 0
    rows[4, [V // 2, V // 2 + 1, V // 2 + 2]] = top # adjacent ties inside one 16-byte load
    rows[5, [1023 % V, (1024 + 7) % V, (4096 + 3) % V]] = top   # ties in different threads / waves of the scan
    rows[6, 0] = top
    rows[7, [3, 2]] = top
    rows[8, :] = -np.inf
    rows[8, V - 2] = np.float32(-1e30)              # one finite value among -inf
    rows[9, [V - 3, 7]] = np.float32(np.finfo(np.float32).max)
    rows[10, :] = np.float32(-0.0)
    rows[10, 11] = np.float32(0.0)                  # +0 == -0: still the first index
    rows[11, 0] = np.nan                            # the scan starts from a NaN and never leaves it: index 0
    rows[11, 7] = top
    return rows


@pytest.mark.parametrize("V", [32000, 32003, 777, 13])
def test_device_argmax_follows_the_reference_on_ties(hip, oracle, V):
    r = np.random.Generator(np.random.PCG64(V))
    rows = _tie_rows(V, r)
    want = _first_max(rows)
    smp = capi.Sampler()
    oracle.build_sampler(C.byref(smp), V, 0.0, 0.9, 1)
    from_oracle = [oracle.sample(C.byref(smp), rows[k].copy().ctypes.data_as(C.POINTER(C.c_float))) for k in range(len(rows))]
    assert from_oracle == want.tolist()             # the rule as the CPU restatement states it
    np.testing.assert_array_equal(_device_argmax(hip, rows, 1), want)


@pytest.mark.parametrize("V,slices", [(1000, 2), (1000, 4), (32000, 8), (100, 8), (777, 3)])
def test_sliced_argmax_merges_ties_across_slices(hip, V, slices):
    """The tensor-parallel classifier's hand-off: one (maximum, index) pair per vocabulary slice,
    merged; ties that straddle slice boundaries must still go to the lowest index."""
    r = np.random.Generator(np.random.PCG64(V + slices))
    rows = _tie_rows(V, r)
    per = ((V + 15) // 16 + slices - 1) // slices * 16
    extra = (r.standard_normal((6, V)) * 2).astype(np.float32)
    top = np.float32(60.0)
    b = min(per, V - 1)
    extra[0, [b - 1, b]] = top                      # last element of slice 0 and first of slice 1
    extra[1, [b, b - 1]] = top
    extra[2, [min(per * (slices - 1), V - 1), 2]] = top   # first and last slice
    extra[3, [V - 1, b]] = top
    extra[4, :] = -np.inf                           # every slice reports -inf: index 0
    extra[5, min(2 * per, V - 1):] = top            # a run of ties from a later slice to the end
    rows = np.concatenate([rows, extra])
    np.testing.assert_array_equal(_device_argmax(hip, rows, slices), _first_max(rows))


def test_sample_reads_the_host_row_as_the_caller_left_it(hip, oracle, tmp_models):
    """In the default hand-off mode RunState.logits is host memory the driver may edit between
    transformer_batch and sample() (logit bias, banned tokens): sample() must see the edit, and
    with a temperature it leaves the probabilities in the row like sampler.c:100-121."""
    path, shape = tmp_models("tiny")
    V = shape.vocab_size
    assert hip.nvllm_get_logits_mode() == 0
    e = capi.Engine(hip, path, 2, temperature=0.0, chunk_size=8)
    e.set_request(0, synth.make_requests(1, 9, 9, V, seed=3)[0], 4, 0)
    e.set_request(1, synth.make_requests(1, 5, 5, V, seed=4)[0], 4, 0)
    assert e.step()
    base = hip.nvllm_engine_logits(e.h)
    rows = np.ctypeslib.as_array(base, shape=(2, V))            # a view of the library's host buffer
    untouched = rows.copy()
    row_ptr = lambda k: C.cast(C.addressof(base.contents) + k * V * 4, C.POINTER(C.c_float))
    smp = capi.Sampler()
    hip.build_sampler(C.byref(smp), V, 0.0, 0.9, 7)
    natural = int(np.argmax(untouched[1]))
    assert hip.sample(C.byref(smp), row_ptr(1)) == natural
    banned = natural
    rows[1, banned] = -np.inf                                   # ban the winner ...
    forced = (natural + 17) % V
    rows[1, forced] = 1e9                                       # ... and force another token
    assert hip.sample(C.byref(smp), row_ptr(1)) == forced
    # with a temperature: same edited row through the oracle's sampler with the same seed
    rows[0, 3] += 25.0
    rows[0, 100] += 24.0
    edited = rows[0].copy()
    osmp = capi.Sampler()
    oracle.build_sampler(C.byref(osmp), V, 0.8, 0.9, 99)
    orow = edited.copy()
    want = [oracle.sample(C.byref(osmp), orow.ctypes.data_as(C.POINTER(C.c_float)))]
    hip.build_sampler(C.byref(smp), V, 0.8, 0.9, 99)            # srand(99) again
    got = [hip.sample(C.byref(smp), row_ptr(0))]
    assert got == want and got[0] in (3, 100)
    np.testing.assert_allclose(rows[0], orow, rtol=2e-6, atol=1e-12)   # the row now holds the probabilities
    assert abs(float(rows[0].sum()) - 1.0) < 1e-5
    # arg-max hand-off: the host buffer is never filled, the pointer only names the output row
    e.close()
    hip.nvllm_set_logits_mode(1)
    try:
        e = capi.Engine(hip, path, 2, temperature=0.0, chunk_size=8)
        e.set_request(0, synth.make_requests(1, 9, 9, V, seed=3)[0], 4, 0)
        e.set_request(1, synth.make_requests(1, 5, 5, V, seed=4)[0], 4, 0)
        assert e.step()
        base = hip.nvllm_engine_logits(e.h)
        n = C.c_int()
        am = hip.nvllm_last_argmax(C.byref(n))
        assert n.value == 2 and am[1] == natural
        hip.build_sampler(C.byref(smp), V, 0.0, 0.9, 7)
        assert hip.sample(C.byref(smp), C.cast(C.addressof(base.contents) + V * 4, C.POINTER(C.c_float))) == natural
        e.close()
    finally:
        hip.nvllm_set_logits_mode(0)


def _classifier_with_repeating_rows(src, dst, shape, period=37, nan_row=3):
    """Copy of a model file whose classifier rows repeat with `period` (every logits row then attains
    its maximum at several indices, in every 64-feature tile) and hold one row of NaN."""
    import shutil
    shutil.copyfile(src, dst)
    V, D = shape.vocab_size, shape.dim
    with open(dst, "r+b") as f:
        f.seek(-V * D * 2, 2)
        head = np.frombuffer(f.read(V * D * 2), np.float16).reshape(V, D).copy()
        head[:] = head[np.arange(V) % period]
        head[nan_row] = np.nan
        f.seek(-V * D * 2, 2)
        f.write(head.tobytes())


@pytest.mark.parametrize("n_seqs", [5, 130])
def test_greedy_pick_from_the_classifier_tiles_takes_the_first_maximum(hip, oracle, tmp_models, tmp_path, n_seqs):
    """The greedy token comes from (maximum, index) pairs that the classifier's tiles leave behind, merged
    per row (130 rows: the dense-tile classifier, rows scanned afterwards).  With classifier rows that repeat
    every 37 indices each maximum is attained about V / 37 times, inside every tile and across tiles: the
    pick must be the lowest index (sampler.c:86-97), a NaN logit never wins."""
    src, shape = tmp_models("tiny", seed=5, vocab_size=1000 if n_seqs > 5 else 777, max_seq_len=256)
    path = str(tmp_path / "ties.bin")
    _classifier_with_repeating_rows(src, path, shape)
    prompts = synth.make_requests(n_seqs, 1, 1 if n_seqs > 5 else 9, shape.vocab_size, seed=6)
    kw = dict(n_new=6, chunk_size=64, temperature=0.0, max_steps=100, collect_logits=True)
    from tests import util
    got = util.run_engine(hip, path, prompts, **kw)                              # rows to the host as well
    hip.nvllm_set_logits_mode(1)
    try:
        got2 = util.run_engine(hip, path, prompts, **{k: v for k, v in kw.items() if k != "collect_logits"})
    finally:
        hip.nvllm_set_logits_mode(0)
    for sg in got["steps"]:
        rows = sg["logits"]
        assert np.isnan(rows[:, 3]).all()
        first = [int(np.flatnonzero(r == np.nanmax(r))[0]) for r in rows]
        # the ties are there (index 3 is NaN: that class's first finite repeat is 3 + 37)
        assert all((i < 37 or i == 40) and np.count_nonzero(r == np.nanmax(r)) > 1 for i, r in zip(first, rows))
        np.testing.assert_array_equal(sg["sampled"], first)
    for a, b in zip(got2["histories"], got["histories"]):       # the pick from the tiles' pairs
        np.testing.assert_array_equal(a, b)
    # the CPU checker agrees wherever its own rows tie exactly (its row loop may round repeated rows differently)
    want = util.run_engine(oracle, path, prompts, **kw)
    if all(int(t) < 37 or int(t) == 40 for s in want["steps"] for t in s["sampled"]):
        for a, b in zip(got["histories"], want["histories"]):
            np.testing.assert_array_equal(a, b)
