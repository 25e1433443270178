"""One rank of tests/test_tp_plan.py: the tensor-parallel partition evaluated with the CPU
oracle's operators and a gloo all_reduce (no GPU involved)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from nvllm_amd import capi  # noqa: E402
from tests import util  # noqa: E402


def main():
    spec = json.loads(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D, F, H, KVH, T = spec["dim"], spec["hidden_dim"], spec["n_heads"], spec["n_kv_heads"], spec["tokens"]
    hd = D // H
    cfg = capi.Config(dim=D, hidden_dim=F, n_layers=1, n_heads=H, n_kv_heads=KVH, vocab_size=64, max_seq_len=64)
    sh = capi.TpShard()
    dll = C.CDLL(capi.HIP_LIB, mode=C.RTLD_LOCAL)            # host-side planning only: no GPU call
    dll.nvllm_tp_shard.argtypes = [C.POINTER(capi.Config), C.c_int, C.c_int, C.POINTER(capi.TpShard)]
    assert dll.nvllm_tp_shard(C.byref(cfg), rank, world, C.byref(sh)) == 0
    q0, ql = sh.head0 * hd, sh.n_heads * hd
    f0, fl = sh.hidden0, sh.hidden

    r = np.random.Generator(np.random.PCG64(spec["seed"]))    # same numbers on every rank
    att = r.standard_normal((T, H * hd)).astype(np.float32)   # attention output, all heads
    x = r.standard_normal((T, D)).astype(np.float32)          # normalised MLP input
    wo = (r.standard_normal((D, H * hd)) * 0.1).astype(np.float32)
    wg = (r.standard_normal((F, D)) * 0.1).astype(np.float32)
    wu = (r.standard_normal((F, D)) * 0.1).astype(np.float32)
    wd = (r.standard_normal((D, F)) * 0.1).astype(np.float32)

    from tests import checkers
    oracle = checkers.load_oracle()
    mm = lambda a, w: util.op_matmul(oracle, np.ascontiguousarray(a), np.ascontiguousarray(w))
    # the whole model on one rank
    full_o = mm(att, wo)
    full_m = mm(util.op_swiglu(oracle, mm(x, wg), mm(x, wu), inplace=False), wd)
    # this rank's slice, then the sum over the ranks
    part_o = mm(att[:, q0:q0 + ql], wo[:, q0:q0 + ql])
    part_m = mm(util.op_swiglu(oracle, mm(x, wg[f0:f0 + fl]), mm(x, wu[f0:f0 + fl]), inplace=False), wd[:, f0:f0 + fl])
    to, tm = torch.from_numpy(part_o.copy()), torch.from_numpy(part_m.copy())
    dist.all_reduce(to)
    dist.all_reduce(tm)
    err_o = float(np.abs(to.numpy() - full_o).max() / np.abs(full_o).max())
    err_m = float(np.abs(tm.numpy() - full_m).max() / np.abs(full_m).max())
    with open(os.path.join(spec["out_dir"], f"rank{rank}.json"), "w") as f:
        json.dump(dict(rank=rank, err_o=err_o, err_m=err_m, heads=[sh.head0, sh.n_heads], kv=[sh.kv_head0, sh.n_kv_heads],
                       hidden=[sh.hidden0, sh.hidden]), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
