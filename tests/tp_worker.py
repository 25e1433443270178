"""One tensor-parallel rank for tests/test_gpu_tp.py (started by torch.distributed.run).

Every rank loads its slice of the same model file into the HIP library, with the
all-reduce either supplied by this script -- the partial sums go through the host and a
gloo all_reduce -- or the library's own exchange over peer-mapped memory (spec "exchange":
"peer"; the handles travel by gloo).  Both let several ranks share the one GPU of a test box
(RCCL itself refuses two ranks on one device).  Every rank also runs the CPU oracle on the
whole model and compares."""
import time
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from nvllm_amd import capi, synth  # noqa: E402
from tests import util  # noqa: E402


def main():
    spec = json.loads(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape = synth.ModelShape.named(spec["name"], **spec["over"])
    path = spec["model_path"]
    prompts = synth.make_requests(spec["n_seqs"], 3, 40, shape.vocab_size, seed=spec["seed"])
    kw = dict(n_new=spec["n_new"], arrivals=[(2 * i) % 3 for i in range(spec["n_seqs"])], collect_logits=True,
              chunk_size=spec["chunk"], temperature=0.0, max_steps=600)

    from tests import checkers
    oracle = checkers.load_oracle()
    want = util.run_engine(oracle, path, prompts, **kw)

    hip = capi.load_hip()
    hip.init_ops()
    hip.nvllm_set_kv_elem_bytes(spec["kv_bytes"])
    calls = {"n": 0, "floats": 0}

    def allreduce(ctx, buf, count, stream):
        hip.device_sync()                                     # the partial sums are complete
        host = np.empty(count, np.float32)
        hip.device_memcpy(host.ctypes.data, buf, host.nbytes, capi.DEVICE_TO_HOST)
        t = torch.from_numpy(host)
        dist.all_reduce(t)                                    # sum over the ranks
        hip.device_memcpy(buf, host.ctypes.data, host.nbytes, capi.HOST_TO_DEVICE)
        calls["n"] += 1
        calls["floats"] += count

    peer = spec.get("exchange") == "peer"
    timing = None
    if peer:
        mine = (C.c_ubyte * capi.TP_PEER_HANDLE_BYTES)()
        assert hip.nvllm_tp_peer_handle(mine) == 0
        gathered = [torch.zeros(capi.TP_PEER_HANDLE_BYTES, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor(list(mine), dtype=torch.uint8))
        handles = (C.c_ubyte * (capi.TP_PEER_HANDLE_BYTES * world))(*[int(b) for t in gathered for b in t])
        assert hip.nvllm_tp_init_peer(rank, world, handles) == 0
        if spec.get("lonely_call"):
            # rank 0 all-reduces alone: its kernel must give up after NVLLM_TP_PEER_WAIT_S, the launch must end,
            # and the next call on the host must stop the process with a message (the test reads stderr)
            dev = C.c_void_p()
            assert hip.device_malloc(C.byref(dev), 4096) == 0
            if rank == 0:
                t0 = time.perf_counter()
                hip.nvllm_tp_allreduce(dev, 1024)
                hip.nvllm_tp_allreduce(dev, 1024)      # behind a wait that ran out nothing waits again
                hip.nvllm_tp_allreduce(dev, 1024)
                hip.device_sync()
                print(f"LONELY waited {time.perf_counter() - t0:.1f} s", flush=True)
                hip.nvllm_tp_allreduce(dev, 1024)      # aborts here
                print("LONELY still alive", flush=True)
            return
        # the exchange alone, on known values: rank r contributes (r + 1) * pattern
        for n in spec.get("probe_counts", [4, 1000, 128 * 4096, (1 << 20) + 2052]):
            pattern = ((np.arange(n) % 251) - 125).astype(np.float32) / 64
            host = (pattern * (rank + 1)).astype(np.float32)
            dev = C.c_void_p()
            assert hip.device_malloc(C.byref(dev), host.nbytes) == 0
            hip.device_memcpy(dev, host.ctypes.data, host.nbytes, capi.HOST_TO_DEVICE)
            assert hip.nvllm_tp_allreduce(dev, n) == 0
            hip.device_sync()
            back = np.empty_like(host)
            hip.device_memcpy(back.ctypes.data, dev, host.nbytes, capi.DEVICE_TO_HOST)
            want_sum = np.zeros(n, np.float32)
            for r in range(world):                      # the exchange's own order: rank 0 first
                want_sum = want_sum + pattern * np.float32(r + 1)
            assert np.array_equal(back, want_sum), (rank, n, int((back != want_sum).sum()))
            if n == 128 * 4096:                         # BASELINE.json configs[4]'s message: 128 rows x 4096 fp32
                dist.barrier()
                reps = 200
                t0 = time.perf_counter()
                for _ in range(reps):
                    hip.nvllm_tp_allreduce(dev, n)
                hip.device_sync()
                timing = (time.perf_counter() - t0) / reps * 1e6
            hip.device_free(dev)
        c0, f0 = C.c_ulonglong(), C.c_ulonglong()
        hip.nvllm_tp_peer_stats(C.byref(c0), C.byref(f0))
    else:
        cb = capi.ALLREDUCE_FN(allreduce)
        assert hip.nvllm_tp_init_custom(rank, world, cb, None) == 0
    assert (hip.nvllm_tp_rank(), hip.nvllm_tp_world()) == (rank, world)
    dist.barrier()    # the ranks start the forward passes together (the oracle runs above took different times)
    got = util.run_engine(hip, path, prompts, **kw)
    if peer:
        c1, f1 = C.c_ulonglong(), C.c_ulonglong()
        hip.nvllm_tp_peer_stats(C.byref(c1), C.byref(f1))
        reduce_calls, reduce_floats = c1.value - c0.value, f1.value - f0.value
    else:
        reduce_calls, reduce_floats = calls["n"], calls["floats"]
    # arg-max hand-off: the ranks exchange one (max, index) pair per row instead of logits
    hip.nvllm_set_logits_mode(1)
    kw2 = dict(kw, collect_logits=False)
    got_argmax = util.run_engine(hip, path, prompts, **kw2)
    # temperature / top-p sampling from rows that exist only as vocabulary slices on the ranks
    kw3 = dict(kw2, temperature=0.8, topp=0.9, seed=77)
    want_sampled = util.run_engine(oracle, path, prompts, **kw3)
    got_sampled = util.run_engine(hip, path, prompts, **kw3)
    hip.nvllm_set_logits_mode(0)
    hip.nvllm_tp_finalize()
    cfg = capi.Config(dim=shape.dim, hidden_dim=shape.hidden_dim, n_layers=shape.n_layers, n_heads=shape.n_heads,
                      n_kv_heads=shape.n_kv_heads, vocab_size=shape.vocab_size, max_seq_len=shape.max_seq_len)
    sh = capi.TpShard()
    assert hip.nvllm_tp_shard(C.byref(cfg), rank, world, C.byref(sh)) == 0

    same_tokens = all(np.array_equal(a, b) for a, b in zip(got["histories"], want["histories"]))
    same_trace = np.array_equal(got["trace"], want["trace"])
    worst = max(float(np.abs(a["logits"] - b["logits"]).max()) for a, b in zip(got["steps"], want["steps"]))
    tokens_forwarded = int(sum(len(s["tokens"]) for s in want["steps"]))
    same_argmax = all(np.array_equal(a, b) for a, b in zip(got_argmax["histories"], want["histories"]))
    same_sampled = all(np.array_equal(a, b) for a, b in zip(got_sampled["histories"], want_sampled["histories"]))
    rows = int(sum(len(s["output_indices"]) for s in want["steps"]))
    out = dict(rank=rank, world=world, same_tokens=bool(same_tokens), same_trace=bool(same_trace), worst=worst,
               same_argmax=bool(same_argmax), same_sampled=bool(same_sampled),
               allreduce_calls=reduce_calls, allreduce_floats=reduce_floats, steps=len(want["steps"]),
               tokens_forwarded=tokens_forwarded, n_layers=shape.n_layers, dim=shape.dim,
               output_rows=rows, vocab_slice=sh.vocab_slice, vocab=[sh.vocab0, sh.vocab], allreduce_us_128x4096=timing,
               logits_crc=int(sum(int(np.frombuffer(a["logits"].tobytes(), np.uint32).astype(np.uint64).sum()) for a in got["steps"]) % (1 << 61)))
    with open(os.path.join(spec["out_dir"], f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
