"""Edge cases of the driver + forward pass against the CPU oracle, in both hand-off modes (host
logits rows: the step-by-step loop; arg-max: the two-halves loop): one-token prompts, requests that
generate nothing, a pass that fills the whole batch capacity while another sequence has to wait
(src/main.c:315-318), steps on which nobody is scheduled (src/main.c:358-362), the step cap
(src/main.c:416-418), and an empty forward call."""
import ctypes as C

import numpy as np
import pytest

from nvllm_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _both_modes(hip, oracle, path, prompts, n_new, arrivals, **opts):
    want = util.run_engine(oracle, path, prompts, n_new, arrivals=arrivals, **opts)
    outs = []
    for mode in (0, 1):
        hip.nvllm_set_logits_mode(mode)
        try:
            e = capi.Engine(hip, path, len(prompts), **opts)
            for i, p in enumerate(prompts):
                e.set_request(i, p, n_new[i], arrivals[i])
            steps = 0
            while e.step():
                steps += 1
            outs.append(dict(histories=[e.history(i) for i in range(len(prompts))], trace=e.alloc_trace(), steps=steps,
                             global_step=hip.nvllm_engine_global_step(e.h)))
            e.close()
        finally:
            hip.nvllm_set_logits_mode(0)
    for got in outs:
        np.testing.assert_array_equal(got["trace"], want["trace"])
        for a, b in zip(got["histories"], want["histories"]):
            np.testing.assert_array_equal(a, b)
    return want, outs


def test_one_token_prompts_and_requests_that_generate_nothing(hip, oracle, tmp_models):
    path, shape = tmp_models("tiny")
    prompts = synth.make_requests(5, 1, 9, shape.vocab_size, seed=81)
    prompts[0] = prompts[0][:1]
    prompts[3] = prompts[3][:1]
    n_new = [6, 0, 1, 0, 3]           # sequences 1 and 3 end with their prompt
    want, _ = _both_modes(hip, oracle, path, prompts, n_new, [0, 0, 1, 2, 2], chunk_size=4, temperature=0.0, max_steps=100)
    assert [len(h) for h in want["histories"]] == [len(p) + n for p, n in zip(prompts, n_new)]


def test_a_pass_that_fills_the_batch_capacity(hip, oracle, tmp_models):
    """max_seq_len 128 tokens per pass: four 32-token prompts fill it exactly, the fifth sequence is passed
    over until there is room (main.c:315-318)."""
    path, shape = tmp_models("tiny")
    prompts = synth.make_requests(5, 32, 32, shape.vocab_size, seed=82)
    want, outs = _both_modes(hip, oracle, path, prompts, [4] * 5, [0] * 5, temperature=0.0, max_steps=100)
    assert len(want["steps"][0]["tokens"]) == 128 and set(want["steps"][0]["seq_ids"].tolist()) == {0, 1, 2, 3}


def test_idle_steps_and_the_step_cap(hip, oracle, tmp_models):
    path, shape = tmp_models("tiny")
    prompts = synth.make_requests(3, 4, 12, shape.vocab_size, seed=83)
    # nobody is there before step 3 (idle iterations advance the step counter); the cap ends the run early
    want, outs = _both_modes(hip, oracle, path, prompts, [30, 30, 30], [3, 5, 9], chunk_size=8, temperature=0.0, max_steps=20)
    assert all(o["global_step"] == 21 for o in outs)
    assert all(len(h) < len(p) + 30 + 1 for h, p in zip(want["histories"], prompts))


def test_empty_forward_call_is_a_no_op(hip, tmp_models):
    path, shape = tmp_models("tiny")
    w, cfg, st = capi.Weights(), capi.Config(), capi.RunState()
    hip.load_model(C.byref(w), C.byref(cfg), path.encode())
    hip.malloc_run_state(C.byref(st), C.byref(cfg))
    try:
        z = capi.ints([0])
        hip.transformer_batch(z, 0, z, z, z, 0, C.byref(cfg), C.byref(st), C.byref(w), None)   # model.c:347: nothing to do
        hip.device_sync()
    finally:
        hip.free_run_state(C.byref(st))
        hip.nvllm_free_model(C.byref(w))
