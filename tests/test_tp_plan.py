"""Host side of tensor parallelism, on CPU: which heads and hidden columns a rank owns
(nvllm_tp_shard), and that the partition plus a sum all-reduce reproduces the
unsharded projections (two gloo ranks, the oracle's operators)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

from nvllm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shard():
    dll = C.CDLL(capi.HIP_LIB, mode=C.RTLD_LOCAL)   # dlopen only; nvllm_tp_shard is plain host arithmetic
    dll.nvllm_tp_shard.argtypes = [C.POINTER(capi.Config), C.c_int, C.c_int, C.POINTER(capi.TpShard)]
    dll.nvllm_tp_shard.restype = C.c_int

    def call(rank, world, **shape):
        cfg = capi.Config(vocab_size=32000, max_seq_len=2048, n_layers=2, **shape)
        out = capi.TpShard()
        rc = dll.nvllm_tp_shard(C.byref(cfg), rank, world, C.byref(out))
        return rc, out
    return call


LLAMA7B = dict(dim=4096, hidden_dim=11008, n_heads=32, n_kv_heads=32)
TINYLLAMA = dict(dim=2048, hidden_dim=5632, n_heads=32, n_kv_heads=4)


@pytest.mark.parametrize("shape,world", [(LLAMA7B, 8), (LLAMA7B, 2), (TINYLLAMA, 4), (TINYLLAMA, 2), (TINYLLAMA, 1)])
def test_slices_partition_heads_and_hidden_columns(shard, shape, world):
    group = shape["n_heads"] // shape["n_kv_heads"]
    heads, kvs, cols = [], [], []
    for rank in range(world):
        rc, s = shard(rank, world, **shape)
        assert rc == 0 and (s.rank, s.world) == (rank, world)
        assert s.head0 == s.kv_head0 * group and s.n_heads == s.n_kv_heads * group   # query heads stay with their kv head
        assert s.hidden % 32 == 0
        heads += range(s.head0, s.head0 + s.n_heads)
        kvs += range(s.kv_head0, s.kv_head0 + s.n_kv_heads)
        cols += range(s.hidden0, s.hidden0 + s.hidden)
    rows = []
    for rank in range(world):
        _, s = shard(rank, world, **shape)
        assert s.vocab_slice % 16 == 0 and 0 <= s.vocab <= s.vocab_slice
        assert s.vocab0 == min(rank * s.vocab_slice, 32000)
        rows += range(s.vocab0, s.vocab0 + s.vocab)
    assert rows == list(range(32000))
    assert heads == list(range(shape["n_heads"]))
    assert kvs == list(range(shape["n_kv_heads"]))
    assert cols == list(range(shape["hidden_dim"]))


def test_shapes_that_do_not_divide_are_refused(shard):
    assert shard(0, 8, **TINYLLAMA)[0] == -1                  # 4 kv heads over 8 ranks
    assert shard(0, 3, **LLAMA7B)[0] == -1
    assert shard(0, 2, dim=64, hidden_dim=160, n_heads=4, n_kv_heads=2)[0] == -1   # 80 hidden columns: not whole 32-groups
    assert shard(2, 2, **LLAMA7B)[0] == -1                    # rank outside the world
    assert shard(0, 0, **LLAMA7B)[0] == -1


def test_two_gloo_ranks_reproduce_the_unsharded_projections(tmp_path):
    spec = dict(dim=128, hidden_dim=320, n_heads=8, n_kv_heads=4, tokens=9, seed=7, out_dir=str(tmp_path))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641",
                          os.path.join(ROOT, "tests", "tp_plan_worker.py"), json.dumps(spec)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    for rank in range(2):
        r = json.load(open(tmp_path / f"rank{rank}.json"))
        assert r["heads"] == [4 * rank, 4] and r["kv"] == [2 * rank, 2] and r["hidden"] == [160 * rank, 160]
        # same terms, summed in a different order (slice by slice): fp32 rounding only
        assert r["err_o"] < 1e-5 and r["err_m"] < 1e-5, r
