#!/usr/bin/env python3
"""Does decode attention run faster when its K/V pages sit in the memory-side cache?  The benchmark's
attention call (64 sequences at context ~1040, 32 / 4 heads of 64, fp16 pages, 68 MB of pages) timed
cold (after 512 MiB of other traffic) and again right after itself."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nvllm_amd import capi  # noqa: E402
from tests import util  # noqa: E402


def attn_us(lib):
    for c in range(lib.nvllm_prof_num_classes()):
        if lib.nvllm_prof_name(c) == b"attn":
            n, ms, by, fl = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
            lib.nvllm_prof_get(c, C.byref(n), C.byref(ms), C.byref(by), C.byref(fl))
            return ms.value * 1e3 / max(1, n.value), by.value / max(1, n.value)


def main():
    lib = capi.load_hip()
    lib.init_ops()
    lib.nvllm_set_kv_elem_bytes(2)
    hd, H, KVH, n_seqs, bs, S = 64, 32, 4, 64, 16, 2048
    r = np.random.Generator(np.random.PCG64(1))
    ctx = [int(c) for c in r.integers(1040, 1101, size=n_seqs)]
    nb = sum((c + bs - 1) // bs for c in ctx) + 3
    pool = util.Pool(lib, bs, nb, 1, KVH, hd)
    perm = r.permutation(nb)
    cur = 0
    for s, c in enumerate(ctx):
        pool.new_table(S // bs)
        need = (c + bs - 1) // bs
        pool.set_table(s, perm[cur:cur + need].tolist())
        cur += need
    pos = [c - 1 for c in ctx]
    sid = list(range(n_seqs))
    q = r.standard_normal((n_seqs, H * hd)).astype(np.float32)
    big = lib.empty_device(512 << 20)
    lib.nvllm_prof_enable(1)
    for rep in range(3):
        lib.device_memcpy(big, big, 512 << 20, capi.DEVICE_TO_DEVICE)
        lib.device_sync()
        out = []
        for k in range(4):
            lib.nvllm_prof_reset()
            util.op_paged_attention_batch(lib, pool, q, 0, pos, sid, S, H, KVH, hd)
            out.append(attn_us(lib))
        print("cold, then warm x3 (us):", " ".join(f"{u:6.2f}" for u, _ in out), f"| {out[0][1] / 1e6:.1f} MB per call")
    lib.nvllm_prof_enable(0)


if __name__ == "__main__":
    main()
