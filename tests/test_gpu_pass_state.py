"""What sample() and nvllm_last_argmax() answer for when passes overlap (include/nvllm.h: the forward
pass in two halves), and sample() with a temperature on the host rows of a pass (reference
src/sampler.c:100-139): all rows prepared together, a row the caller changed handled alone."""
import ctypes as C

import numpy as np
import pytest

from nvllm_amd import capi, synth

pytestmark = pytest.mark.gpu


def _row_ptr(base, k, V):
    return C.cast(C.addressof(base.contents) + k * V * 4, C.POINTER(C.c_float))


def test_sample_and_last_argmax_answer_for_the_collected_pass(hip, tmp_models):
    """Between nvllm_forward_enqueue(n + 1) and its collect, the callers still see pass n: its row count,
    its indices, its row naming -- never a mixture of the two passes."""
    path, shape = tmp_models("tiny")
    V = shape.vocab_size
    w, cfg, st = capi.Weights(), capi.Config(), capi.RunState()
    hip.load_model(C.byref(w), C.byref(cfg), path.encode())
    hip.malloc_run_state(C.byref(st), C.byref(cfg))
    mgr = capi.KVCacheManager.in_dll(hip.dll, "g_kv_manager")
    hip.init_kv_cache_manager(C.byref(mgr), 16, 16, cfg.n_layers, cfg.n_kv_heads, cfg.head_dim)
    hip.nvllm_set_logits_mode(1)
    try:
        tables, keep = (C.POINTER(capi.BlockTable) * 5)(), []
        for i in range(5):
            arr = (C.c_int * 4)()
            arr[0] = hip.alloc_block(C.byref(mgr))
            bt = capi.BlockTable(C.cast(arr, capi.c_int_p), 1)
            keep.append((arr, bt))
            tables[i] = C.pointer(bt)

        def args(tokens, seqs):
            n = len(tokens)
            return capi.ints(tokens), n, capi.ints([0] * n), capi.ints(seqs), capi.ints(list(range(n))), n

        a_args, b_args = args([5, 9], [0, 1]), args([7, 11, 13], [2, 3, 4])
        tail = (C.byref(cfg), C.byref(st), C.byref(w), tables)
        n = C.c_int()
        hip.transformer_batch(*a_args, *tail)
        a = list(hip.nvllm_last_argmax(C.byref(n))[:2])
        assert n.value == 2
        hip.transformer_batch(*b_args, *tail)
        b = list(hip.nvllm_last_argmax(C.byref(n))[:3])
        assert n.value == 3 and b[:2] != a              # the two passes are told apart by their indices

        smp = capi.Sampler()
        hip.build_sampler(C.byref(smp), V, 0.0, 0.9, 1)
        row = lambda k: C.cast(st.logits + k * V * 4, C.POINTER(C.c_float))
        out = (C.c_int * 8)()
        ta = hip.nvllm_forward_enqueue(*a_args, None, *tail)
        assert hip.nvllm_forward_collect(ta, out, 8) == 2 and list(out[:2]) == a
        tb = hip.nvllm_forward_enqueue(*b_args, None, *tail)          # in flight, not collected
        seen = hip.nvllm_last_argmax(C.byref(n))
        assert n.value == 2 and list(seen[:2]) == a
        assert [hip.sample(C.byref(smp), row(k)) for k in range(2)] == a
        assert hip.nvllm_forward_collect(tb, out, 8) == 3 and list(out[:3]) == b
        seen = hip.nvllm_last_argmax(C.byref(n))
        assert n.value == 3 and list(seen[:3]) == b
        assert [hip.sample(C.byref(smp), row(k)) for k in range(3)] == b
        # PASS_RING further passes reuse pass b's slot: its indices are withdrawn, not left dangling
        tickets = [hip.nvllm_forward_enqueue(*a_args, None, *tail) for _ in range(4)]
        seen = hip.nvllm_last_argmax(C.byref(n))
        assert n.value == 0 and not seen
        for t in tickets:
            assert hip.nvllm_forward_collect(t, out, 8) == 2 and list(out[:2]) == a
    finally:
        hip.nvllm_set_logits_mode(0)
        hip.free_kv_cache_manager(C.byref(mgr))
        hip.free_run_state(C.byref(st))
        hip.nvllm_free_model(C.byref(w))


@pytest.mark.parametrize("topp", [0.9, 1.0])
def test_host_rows_with_a_temperature_together_and_alone(hip, oracle, tmp_models, topp):
    """Default hand-off mode: the first sample() on a row of the pass uploads all its rows at once; rows the
    caller edits afterwards, or samples a second time (they hold probabilities by then), go alone.  Tokens
    and the rows' final contents are those of the reference's sampler on the same sequence of calls."""
    path, shape = tmp_models("tiny")
    V = shape.vocab_size
    assert hip.nvllm_get_logits_mode() == 0
    e = capi.Engine(hip, path, 6, temperature=0.0, chunk_size=64)
    for i, p in enumerate(synth.make_requests(6, 3, 8, V, seed=91)):
        e.set_request(i, p, 4, 0)
    assert e.step()
    assert len(e.last_batch()["output_indices"]) == 6
    base = hip.nvllm_engine_logits(e.h)
    rows = np.ctypeslib.as_array(base, shape=(6, V))
    mine = rows.copy()                                            # the oracle's rows: same values, own memory

    def calls(lib, ptr, arr):
        smp = capi.Sampler()
        lib.build_sampler(C.byref(smp), V, 0.9, topp, 4321)      # srand(4321)
        out = [lib.sample(C.byref(smp), ptr(0))]
        arr[2, 7] += 6.0                                          # edited after the rows were taken together
        arr[2, 1] = -np.inf
        out.append(lib.sample(C.byref(smp), ptr(2)))
        out.append(lib.sample(C.byref(smp), ptr(3)))
        out.append(lib.sample(C.byref(smp), ptr(0)))              # probabilities by now: softmax of those
        out.append(lib.sample(C.byref(smp), ptr(5)))
        return out

    want = calls(oracle, lambda k: mine[k].ctypes.data_as(C.POINTER(C.c_float)), mine)
    got = calls(hip, lambda k: _row_ptr(base, k, V), rows)
    assert got == want
    np.testing.assert_allclose(rows, mine, rtol=3e-6, atol=1e-12)
    assert np.array_equal(rows[1], mine[1]) and np.array_equal(rows[4], mine[4])   # never sampled: still logits
    e.close()
