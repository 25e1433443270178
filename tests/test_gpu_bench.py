"""bench.py end to end on the GPU at small sizes: the one-line JSON contract, for the data-parallel
workload (with its roofline, variants and CPU-baseline objects) and for the tensor-parallel one run
as a single rank (RCCL communicator of one: the same code path as N ranks)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--model", "stories15M", "--batch", "8", "--prompt-len", "40", "--chunk", "32", "--max-seq-len", "256",
          "--steps", "6", "--warmup", "2", "--profile-steps", "2"]


def _run(extra, tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model-dir", str(tmp_path)] + COMMON + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                 # stdout carries exactly one line
    return json.loads(lines[0])


def _common_fields(d, steps=6):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "tokens/s" and d["n_gpus"] == 1 and d["steps"] == steps and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_data_parallel_line(tmp_path):
    d = _run(["--cpu-steps", "2"], tmp_path)
    _common_fields(d)
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "single"
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    lo, hi = c["context_tokens"]                     # the CPU leg decodes at the GPU run's context
    assert hi - lo == 2 and lo > 40
    v = d["variants"]
    assert set(v) == {"fp32_kv_pool", "logits_rows_to_host", "fp32_kv_pool_and_logits_rows_to_host"}
    assert all(x["value"] > 0 for x in v.values())
    p = d["prefill_roofline"]
    assert abs(p["frac_issued"] - 2 * p["frac"]) < 1e-3          # hi + lo planes: twice the useful flops are issued


def test_tensor_parallel_line_with_one_rank(tmp_path):
    d = _run(["--tp", "--cpu-steps", "0"], tmp_path)
    _common_fields(d)
    assert d["scaling"] == "strong" and d["config"]["parallelism"] == "tp1" and "TP=1" in d["metric"]
    assert d["cpu_baseline"] is None and d["variants"] is None


def test_two_data_parallel_ranks_rehearsed_on_one_gpu(tmp_path):
    """The launcher contract with two ranks (`python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2`): both replicas share this box's one GPU and meet through gloo, so the throughput means
    nothing -- what is checked is the plumbing an 8-GPU run depends on: one shared model file, per-rank
    request shards, the MAX-over-ranks timing and one JSON line from rank 0 with the whole-job aggregate."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model-dir", str(tmp_path)] + COMMON + \
          ["--cpu-steps", "0", "--no-variants"]
    env = dict(os.environ, NVLLM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 2 * 8 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3


def test_two_tensor_parallel_ranks_with_the_peer_exchange_on_one_gpu(tmp_path):
    """`bench.py --tp --tp-exchange peer` under the launcher with two ranks: ONE model split over both, the two
    partial sums per layer all-reduced by the library's own kernel through exchange segments the ranks map from
    each other (hipIpc handles, gathered over gloo).  Both ranks share this box's GPU, so the throughput means
    nothing; the line must be the strong-scaling one with every request counted once."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tp", "--tp-exchange", "peer",
           "--model-dir", str(tmp_path), "--model", "tiny-gqa", "--batch", "8", "--prompt-len", "40", "--chunk", "32",
           "--max-seq-len", "160", "--steps", "6", "--warmup", "2", "--profile-steps", "2", "--cpu-steps", "0"]
    env = dict(os.environ, NVLLM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "tp2 (peer exchange)"
    assert d["value"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
