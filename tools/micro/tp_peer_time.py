"""Times the library's peer-memory all-reduce (nvllm_tp_init_peer) between `world` processes.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \
      tools/micro/tp_peer_time.py [floats ...]

On a box with one GPU all ranks share it (every exchange segment lies in the same HBM, the kernels of
the ranks run side by side): the numbers then show the cost of the protocol -- launches, flag hops,
uncached accesses -- not of xGMI.  On a node each rank takes the GPU of its LOCAL_RANK.
Every size is checked against the exact sum before it is timed."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from nvllm_amd import capi  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [int(a) for a in sys.argv[1:]] or [64 * 2048, 128 * 4096, 1 << 20]
    hip = capi.load_hip()
    n_gpus = torch.cuda.device_count()
    hip.nvllm_set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(n_gpus, 1))
    hip.init_ops()
    mine = (C.c_ubyte * capi.TP_PEER_HANDLE_BYTES)()
    assert hip.nvllm_tp_peer_handle(mine) == 0
    gathered = [torch.zeros(capi.TP_PEER_HANDLE_BYTES, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor(list(mine), dtype=torch.uint8))
    handles = (C.c_ubyte * (capi.TP_PEER_HANDLE_BYTES * world))(*[int(b) for t in gathered for b in t])
    assert hip.nvllm_tp_init_peer(rank, world, handles) == 0
    for n in sizes:
        pattern = ((np.arange(n) % 251) - 125).astype(np.float32) / 64
        host = (pattern * (rank + 1)).astype(np.float32)
        dev = C.c_void_p()
        assert hip.device_malloc(C.byref(dev), host.nbytes) == 0
        hip.device_memcpy(dev, host.ctypes.data, host.nbytes, capi.HOST_TO_DEVICE)
        assert hip.nvllm_tp_allreduce(dev, n) == 0
        hip.device_sync()
        back = np.empty_like(host)
        hip.device_memcpy(back.ctypes.data, dev, host.nbytes, capi.DEVICE_TO_HOST)
        want = np.zeros(n, np.float32)
        for r in range(world):
            want = want + pattern * np.float32(r + 1)
        assert np.array_equal(back, want), (rank, n)
        hip.device_memcpy(dev, np.zeros(n, np.float32).ctypes.data, host.nbytes, capi.HOST_TO_DEVICE)   # sums stay finite over the loop
        best = None
        for _ in range(3):
            dist.barrier()
            reps = 300
            t0 = time.perf_counter()
            for _ in range(reps):
                hip.nvllm_tp_allreduce(dev, n)
            hip.device_sync()
            us = (time.perf_counter() - t0) / reps * 1e6
            best = us if best is None else min(best, us)
        hip.device_free(dev)
        if rank == 0:
            print(f"world {world} gpus {n_gpus} wgs {os.environ.get('NVLLM_TP_PEER_WGS', 'default')}: "
                  f"{n} floats ({n * 4 / 2**20:.2f} MiB) {best:.1f} us per all-reduce", flush=True)
    hip.nvllm_tp_finalize()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
