"""Tensor parallelism of the forward pass: the per-rank slices (heads, hidden columns),
the partial-sum projections and the all-reduce hook, against the CPU oracle run on the
whole model."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from nvllm_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIGHT_TOL = 2e-4     # as for the single-process forward pass
LOGITS_TOL = 2e-2

GQA = dict(dim=256, hidden_dim=512, n_layers=3, n_heads=8, n_kv_heads=4, vocab_size=600, max_seq_len=160)
# one layer at the widths of BASELINE.json configs[4] (Llama-2-7B: dim 4096, 32 heads of 128, hidden 11008);
# a short vocabulary keeps the file and the oracle's classifier small
LLAMA7B_LAYER = dict(dim=4096, hidden_dim=11008, n_layers=1, n_heads=32, n_kv_heads=32, vocab_size=4000, max_seq_len=256)


def _run_and_compare(hip, oracle, path, shape, tol):
    prompts = synth.make_requests(4, 3, 40, shape.vocab_size, seed=17)
    kw = dict(n_new=20, arrivals=[0, 1, 1, 4], collect_logits=True, chunk_size=16, temperature=0.0, max_steps=400)
    o = util.run_engine(oracle, path, prompts, **kw)
    g = util.run_engine(hip, path, prompts, **kw)
    np.testing.assert_array_equal(g["trace"], o["trace"])
    for a, b in zip(g["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)
    worst = max(float(np.abs(a["logits"] - b["logits"]).max()) for a, b in zip(g["steps"], o["steps"]))
    assert worst < tol, worst
    return o


def test_partial_sum_path_with_a_single_rank(hip, oracle, tmp_models):
    """world = 1 with a caller-supplied exchange: the projections that return to the
    residual width leave plain sums, the exchange (here: nothing to add) runs between
    them and the residual kernel, once after wo and once after w_down per layer."""
    path, shape = tmp_models("tiny", seed=3, **GQA)
    calls = []

    def allreduce(ctx, buf, count, stream):
        calls.append(count)

    cb = capi.ALLREDUCE_FN(allreduce)
    assert hip.nvllm_tp_init_custom(0, 1, cb, None) == 0
    try:
        o = _run_and_compare(hip, oracle, path, shape, TIGHT_TOL)
    finally:
        hip.nvllm_tp_finalize()
    assert len(calls) == 2 * shape.n_layers * len(o["steps"])
    assert sum(calls) == 2 * shape.n_layers * shape.dim * sum(len(s["tokens"]) for s in o["steps"])
    # and the fused single-process path is back afterwards
    _run_and_compare(hip, oracle, path, shape, TIGHT_TOL)


@pytest.mark.parametrize("kv_bytes", [2, 4])
def test_7b_width_layer_with_a_single_rank(hip, oracle, tmp_models, kv_bytes):
    """The tensor-parallel forward pass (partial sums, exchange hook, residual kernel, vocabulary-split
    classifier) at the 7B widths: head_dim 128, K = 4096 and 11008, 12288-row fused q/k/v."""
    path, shape = tmp_models("tiny", seed=9, **LLAMA7B_LAYER)
    cb = capi.ALLREDUCE_FN(lambda ctx, buf, count, stream: None)
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(kv_bytes)
    assert hip.nvllm_tp_init_custom(0, 1, cb, None) == 0
    try:
        _run_and_compare(hip, oracle, path, shape, TIGHT_TOL if kv_bytes == 4 else LOGITS_TOL / 10)
    finally:
        hip.nvllm_tp_finalize()
        hip.nvllm_set_kv_elem_bytes(before)


@pytest.mark.parametrize("tensor_parallel", [False, True])
def test_7b_width_layer_at_batch_128(hip, oracle, tmp_models, tensor_parallel):
    """BASELINE.json configs[4]'s batch: 128 decode rows per pass.  At that width the wide projections
    leave the streaming kernel for 128 x 128 tiles (gate/up whole, q/k/v with K cut in two and the halves
    met through a slab); fused epilogues (single process) and plain sums + exchange hook (one rank)."""
    path, shape = tmp_models("tiny", seed=9, **LLAMA7B_LAYER)
    prompts = synth.make_requests(128, 2, 3, shape.vocab_size, seed=41)
    kw = dict(n_new=4, collect_logits=True, chunk_size=64, temperature=0.0, max_steps=50)
    o = util.run_engine(oracle, path, prompts, **kw)
    assert sum(len(s_["tokens"]) == 128 for s_ in o["steps"]) >= 3
    cb = capi.ALLREDUCE_FN(lambda ctx, buf, count, stream: None)
    before = hip.nvllm_get_kv_elem_bytes()
    hip.nvllm_set_kv_elem_bytes(2)
    if tensor_parallel:
        assert hip.nvllm_tp_init_custom(0, 1, cb, None) == 0
    try:
        g = util.run_engine(hip, path, prompts, **kw)
    finally:
        if tensor_parallel:
            hip.nvllm_tp_finalize()
        hip.nvllm_set_kv_elem_bytes(before)
    np.testing.assert_array_equal(g["trace"], o["trace"])
    for a, b in zip(g["histories"], o["histories"]):
        np.testing.assert_array_equal(a, b)
    worst = max(float(np.abs(a["logits"] - b["logits"]).max()) for a, b in zip(g["steps"], o["steps"]))
    assert worst < LOGITS_TOL / 10, worst


def test_rccl_communicator_of_one_rank(hip, oracle, tmp_models):
    """The RCCL plumbing end to end on the one GPU of the box: unique id, communicator,
    ncclAllReduce on the library stream between the projection and the residual kernel."""
    path, shape = tmp_models("tiny", seed=3, **GQA)
    ident = (C.c_char * capi.TP_ID_BYTES)()
    assert hip.nvllm_tp_unique_id(ident) == 0
    assert hip.nvllm_tp_init(0, 1, ident) == 0
    try:
        o = _run_and_compare(hip, oracle, path, shape, TIGHT_TOL)      # ncclAllReduce + ncclAllGather of logits slices
        hip.nvllm_set_logits_mode(1)                                   # ncclAllGather of (max, index) pairs
        prompts = synth.make_requests(4, 3, 40, shape.vocab_size, seed=17)
        g = util.run_engine(hip, path, prompts, n_new=20, arrivals=[0, 1, 1, 4], chunk_size=16, temperature=0.0, max_steps=400)
        for a, b in zip(g["histories"], o["histories"]):
            np.testing.assert_array_equal(a, b)
    finally:
        hip.nvllm_set_logits_mode(0)
        hip.nvllm_tp_finalize()


def test_vocabulary_split_hand_offs_with_a_single_rank(hip, oracle, tmp_models):
    """The classifier path of the tensor-parallel forward pass (slice -> pairs -> merge, and
    slice -> gathered rows for sampling) in one process."""
    path, shape = tmp_models("tiny", seed=3, **GQA)
    cb = capi.ALLREDUCE_FN(lambda ctx, buf, count, stream: None)
    prompts = synth.make_requests(3, 3, 30, shape.vocab_size, seed=29)
    res = {}
    for temperature in (0.0, 0.9):
        kw = dict(n_new=14, chunk_size=16, temperature=temperature, topp=0.8, seed=5)
        o = util.run_engine(oracle, path, prompts, **kw)
        assert hip.nvllm_tp_init_custom(0, 1, cb, None) == 0
        hip.nvllm_set_logits_mode(1)
        try:
            g = util.run_engine(hip, path, prompts, **kw)
        finally:
            hip.nvllm_set_logits_mode(0)
            hip.nvllm_tp_finalize()
        for a, b in zip(g["histories"], o["histories"]):
            np.testing.assert_array_equal(a, b)


def test_setup_is_refused_twice_and_for_bad_placements(hip):
    cb = capi.ALLREDUCE_FN(lambda ctx, buf, count, stream: None)
    assert hip.nvllm_tp_init_custom(2, 2, cb, None) == -1
    assert hip.nvllm_tp_init_custom(0, 1, cb, None) == 0
    try:
        assert hip.nvllm_tp_init_custom(0, 1, cb, None) == -1
    finally:
        hip.nvllm_tp_finalize()
    assert (hip.nvllm_tp_rank(), hip.nvllm_tp_world()) == (0, 1)


@pytest.mark.parametrize("world,kv_bytes,over,exchange", [
    (2, 4, dict(hidden_dim=192), "host"),  # "tiny": 4 heads / 2 kv heads of width 16
    (2, 2, GQA, "host"),
    (2, 4, LLAMA7B_LAYER, "host"),         # 16 heads of 128 and 5504 hidden columns per rank (fp32 rows: the sampled-token
                                           # comparison draws from probabilities that must agree to ~1e-6)
    (2, 2, GQA, "peer"),                   # the library's own exchange: kernels that store into each other's segments
    (4, 4, GQA, "peer"),                   # one kv head (two query heads) per rank
    (2, 4, LLAMA7B_LAYER, "peer"),
    (8, 4, dict(dim=512, hidden_dim=512, n_layers=2, n_heads=8, n_kv_heads=8, vocab_size=600, max_seq_len=160), "peer"),
                                           # BASELINE.json configs[4]'s degree: eight ranks, one head of 64 and 64 hidden columns each
])
def test_ranks_sharing_one_gpu_match_the_oracle(tmp_path, tmp_models, world, kv_bytes, over, exchange):
    """`world` processes, each with its slice of the weights and of the KV pool; the exchange
    staged through the host (gloo), or the library's all-reduce over peer-mapped exchange
    segments (hipIpc handles; here all segments lie in the one GPU's memory, on a node each in
    its own GPU's).  Every rank must reproduce the oracle's tokens, allocation trace and logits."""
    path, shape = tmp_models("tiny", seed=5, **over)
    big = over.get("dim", 0) >= 4096
    small = big or world > 4          # (eight processes on the box's CPU share: a shorter run each)
    spec = dict(name="tiny", over=over, model_path=path, n_seqs=3 if small else 5, seed=23, n_new=8 if small else 16, chunk=12,
                kv_bytes=kv_bytes, out_dir=str(tmp_path), exchange=exchange)
    if world > 4:
        spec["probe_counts"] = [4, 1000, 128 * 4096]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8" if big else ("1" if world > 4 else "2"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29620 + world + kv_bytes + (40 if big else 0) + (100 if exchange == "peer" else 0)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "tp_worker.py"), json.dumps(spec)],
                         env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    for rank in range(world):
        r = json.load(open(tmp_path / f"rank{rank}.json"))
        assert r["same_trace"] and r["same_tokens"], r
        assert r["worst"] < (TIGHT_TOL if kv_bytes == 4 else LOGITS_TOL / 10), r
        assert r["same_argmax"], r      # (max, index) pairs instead of logits rows
        assert r["same_sampled"], r     # temperature / top-p from rows gathered on demand
        # per step: two partial sums per layer, plus the gather of the logits slices (the test's
        # exchange only sums, so a gather is a sum over zero-padded slices)
        assert r["allreduce_calls"] == (2 * r["n_layers"] + 1) * r["steps"]
        assert r["allreduce_floats"] == 2 * r["n_layers"] * r["dim"] * r["tokens_forwarded"] + world * r["vocab_slice"] * r["output_rows"]
        if exchange == "peer":
            print(f"peer all-reduce of 128 x 4096 floats, {world} ranks on one GPU: {r['allreduce_us_128x4096']:.1f} us (rank {rank})")


def test_a_rank_left_alone_by_its_peers_stops_with_a_message(tmp_path, tmp_models):
    """Rank 0 makes three all-reduce calls that rank 1 never makes.  The kernels' waits are bounded (2 s here, 30 s
    by default) and only the first one is sat out; the launches end, and the next call stops the process with a message."""
    path, shape = tmp_models("tiny", seed=5, **GQA)
    spec = dict(name="tiny", over=GQA, model_path=path, n_seqs=1, seed=23, n_new=1, chunk=12, kv_bytes=2, out_dir=str(tmp_path),
                exchange="peer", lonely_call=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", NVLLM_TP_PEER_WAIT_S="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29791",
                          os.path.join(ROOT, "tests", "tp_worker.py"), json.dumps(spec)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    text = out.stdout + out.stderr
    assert "LONELY waited" in text and "LONELY still alive" not in text, text[-3000:]
    waited = float(text.split("LONELY waited")[1].split()[0])
    assert 1.5 < waited < 5.5, waited          # one wait of 2 s (phase A); three calls x two phases sat out in full would be 12 s
    assert "waited 2 s for a flag" in text, text[-3000:]


def test_residual_tail_inside_the_peer_exchange_gives_the_same_bits(tmp_path, tmp_models):
    """The kernel that follows each all-reduce (x += sum, the next projection's input tiles, sums of squares) rides in the
    peer exchange's last phases by default; NVLLM_TP_PEER_FUSED=0 keeps it a kernel of its own.  Two ranks, both ways:
    every logit of every step must be the same bits."""
    path, shape = tmp_models("tiny", seed=5, **GQA)
    crcs = []
    for fused in ("1", "0"):
        out_dir = tmp_path / f"fused{fused}"
        out_dir.mkdir()
        spec = dict(name="tiny", over=GQA, model_path=path, n_seqs=4, seed=23, n_new=12, chunk=12, kv_bytes=2, out_dir=str(out_dir),
                    exchange="peer", probe_counts=[1000])
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", NVLLM_TP_PEER_FUSED=fused)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(29795 + int(fused)),
                              os.path.join(ROOT, "tests", "tp_worker.py"), json.dumps(spec)],
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        rs = [json.load(open(out_dir / f"rank{rank}.json")) for rank in range(2)]
        assert all(r["same_tokens"] and r["same_trace"] for r in rs), rs
        assert rs[0]["logits_crc"] == rs[1]["logits_crc"]          # the ranks hold the same logits
        crcs.append(rs[0]["logits_crc"])
    assert crcs[0] == crcs[1], crcs
