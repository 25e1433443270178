"""The driver's two-halves mode (greedy; arg-max hand-off or logits rows on the host): pass n+1 is enqueued before pass n is
collected and decode rows take their token from the previous pass's arg-max on the GPU.  Its
histories, allocation traces and per-step records must be those of the step-by-step loop and of
the CPU oracle -- including the reference's quirk that a sequence whose chunk boundary leaves one
prompt token continues with a SAMPLED token (src/main.c:292,353-355,402)."""
import os

import numpy as np
import pytest

from nvllm_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _requests(shape, n, seed):
    prompts = synth.make_requests(n, 2, 70, shape.vocab_size, seed=seed)
    prompts[1] = prompts[1][:1].repeat(17) if len(prompts[1]) < 17 else prompts[1][:17]   # 16-token chunk + ONE token left
    prompts[2] = prompts[2][:1]                                                         # a one-token prompt
    return prompts


def _run_blind(lib, path, prompts, n_new, arrivals, pause_at=(), **opts):
    """Steps without looking at the sampled tokens in between (except at pause_at), so passes overlap."""
    e = capi.Engine(lib, path, len(prompts), **opts)
    for i, p in enumerate(prompts):
        e.set_request(i, p, n_new[i], arrivals[i])
    steps, seen = 0, {}
    while e.step():
        steps += 1
        if steps in pause_at:
            seen[steps] = (e.last_batch(), [e.history(i).copy() for i in range(len(prompts))])
    out = dict(histories=[e.history(i) for i in range(len(prompts))], trace=e.alloc_trace(), steps=steps, seen=seen,
               tables=[e.block_table(i) for i in range(len(prompts))], last=e.last_batch())
    e.close()
    return out


@pytest.mark.parametrize("kv_bytes,recycle", [(2, 0), (4, 1)])
def test_two_halves_mode_equals_the_stepwise_loop_and_the_oracle(hip, oracle, tmp_models, kv_bytes, recycle):
    path, shape = tmp_models("stories15M", max_seq_len=256, vocab_size=2000)
    n = 9
    prompts = _requests(shape, n, seed=51)
    n_new = [5 + 3 * (i % 4) for i in range(n)]
    arrivals = [(4 * i) % 7 for i in range(n)]
    opts = dict(chunk_size=16, temperature=0.0, max_steps=400, recycle_blocks=recycle,
                num_blocks=(n * 16 if not recycle else 40))
    want = util.run_engine(oracle, path, prompts, n_new, arrivals=arrivals, **opts)
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(kv_bytes)
    hip.nvllm_set_logits_mode(1)
    try:
        blind = _run_blind(hip, path, prompts, n_new, arrivals, pause_at=(3, 11), **opts)
        os.environ["NVLLM_ENGINE_STEPWISE"] = "1"
        try:
            stepwise = _run_blind(hip, path, prompts, n_new, arrivals, pause_at=(3, 11), **opts)
        finally:
            del os.environ["NVLLM_ENGINE_STEPWISE"]
        watched = util.run_engine(hip, path, prompts, n_new, arrivals=arrivals, **opts)   # queries every step
    finally:
        hip.nvllm_set_logits_mode(0)
        hip.nvllm_set_kv_elem_bytes(before)
    assert blind["steps"] == stepwise["steps"] == len(want["steps"])
    for got in (blind, stepwise, watched):
        np.testing.assert_array_equal(got["trace"], want["trace"])
        for a, b in zip(got["histories"], want["histories"]):
            np.testing.assert_array_equal(a, b)
    for k in (3, 11):                                # what an observer sees mid-run is the same in both modes
        (lb_a, h_a), (lb_b, h_b) = blind["seen"][k], stepwise["seen"][k]
        for key in lb_a:
            np.testing.assert_array_equal(lb_a[key], lb_b[key])
            np.testing.assert_array_equal(lb_a[key], want["steps"][k - 1][key])
        for a, b in zip(h_a, h_b):
            np.testing.assert_array_equal(a, b)
    for key in blind["last"]:
        np.testing.assert_array_equal(blind["last"][key], stepwise["last"][key])


def test_two_halves_mode_with_logits_rows_on_the_host(hip, oracle, tmp_models):
    """The reference's own hand-off (logits rows in host memory, the sampler reads them): passes still
    overlap -- two RunStates alternate, the tokens come from sample() on the collected pass's rows -- and
    an observer who reads the rows mid-run finds those of the pass just collected."""
    path, shape = tmp_models("stories15M", max_seq_len=256, vocab_size=2000)
    n = 7
    prompts = _requests(shape, n, seed=77)
    n_new = [6 + 2 * (i % 3) for i in range(n)]
    arrivals = [(3 * i) % 5 for i in range(n)]
    opts = dict(chunk_size=16, temperature=0.0, max_steps=400)
    want = util.run_engine(oracle, path, prompts, n_new, arrivals=arrivals, collect_logits=True, **opts)
    assert hip.nvllm_get_logits_mode() == 0
    blind = _run_blind(hip, path, prompts, n_new, arrivals, pause_at=(4, 9), **opts)
    os.environ["NVLLM_ENGINE_STEPWISE"] = "1"
    try:
        stepwise = _run_blind(hip, path, prompts, n_new, arrivals, pause_at=(4, 9), **opts)
    finally:
        del os.environ["NVLLM_ENGINE_STEPWISE"]
    watched = util.run_engine(hip, path, prompts, n_new, arrivals=arrivals, collect_logits=True, **opts)
    assert blind["steps"] == stepwise["steps"] == len(want["steps"])
    for got in (blind, stepwise, watched):
        np.testing.assert_array_equal(got["trace"], want["trace"])
        for a, b in zip(got["histories"], want["histories"]):
            np.testing.assert_array_equal(a, b)
    for sg, so in zip(watched["steps"], want["steps"]):            # rows read after every step: the step's own
        assert np.abs(sg["logits"] - so["logits"]).max() < 2e-4
    for k in (4, 9):
        (lb_a, h_a), (lb_b, h_b) = blind["seen"][k], stepwise["seen"][k]
        for key in lb_a:
            np.testing.assert_array_equal(lb_a[key], lb_b[key])


def test_forward_in_two_halves_through_the_c_abi(hip, tmp_models):
    """nvllm_forward_enqueue / nvllm_forward_collect directly: three chained decode passes, the
    second and third fed from the GPU, against three transformer_batch calls fed from the host."""
    import ctypes as C
    path, shape = tmp_models("tiny")
    hip.nvllm_set_logits_mode(1)
    try:
        res = {}
        for mode in ("sync", "chained"):
            e = capi.Engine(hip, path, 2, temperature=0.0, chunk_size=64)   # owns model, workspace and pool
            e.set_request(0, synth.make_requests(1, 6, 6, shape.vocab_size, seed=1)[0], 8, 0)
            e.set_request(1, synth.make_requests(1, 9, 9, shape.vocab_size, seed=2)[0], 8, 0)
            e.step()                                                       # prefill of both prompts
            res[mode] = [e.history(0).copy(), e.history(1).copy()]
            if mode == "sync":
                for _ in range(3):
                    e.step()
                    e.history(0)
            else:
                e.run(3)
            res[mode] += [e.history(0).copy(), e.history(1).copy()]
            e.close()
        for a, b in zip(res["sync"], res["chained"]):
            np.testing.assert_array_equal(a, b)
        assert hip.nvllm_forward_enqueue(None, 0, None, None, None, 0, None, None, None, None, None) == -1
    finally:
        hip.nvllm_set_logits_mode(0)
    assert hip.nvllm_forward_enqueue(None, 1, None, None, None, 0, None, None, None, None, None) == -1   # wrong hand-off mode


def test_two_halves_mode_on_an_fp32_resident_model(hip, oracle, tmp_models):
    """The fp32 operator path takes its chained tokens through the patch kernel."""
    path, shape = tmp_models("tiny", seed=14, dtype="f32", shared=True, rope_tail=True, exact16=False)
    prompts = synth.make_requests(4, 3, 30, shape.vocab_size, seed=61)
    n_new, arrivals = [9, 6, 12, 7], [0, 0, 2, 5]
    opts = dict(chunk_size=16, temperature=0.0, max_steps=200)
    want = util.run_engine(oracle, path, prompts, n_new, arrivals=arrivals, **opts)
    hip.nvllm_set_logits_mode(1)
    try:
        got = _run_blind(hip, path, prompts, n_new, arrivals, **opts)
    finally:
        hip.nvllm_set_logits_mode(0)
    np.testing.assert_array_equal(got["trace"], want["trace"])
    for a, b in zip(got["histories"], want["histories"]):
        np.testing.assert_array_equal(a, b)
