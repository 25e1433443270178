"""Child process of test_gpu_dense_tile.py: runs ops.h matmul() on the HIP library for every projection
handle of a model file at the given token counts and stores the results.  The environment of the child
selects the dense-tile kernel's form (tile height, wave form, K split), which the library reads once."""
import ctypes as C
import sys

import numpy as np

from nvllm_amd import capi


def cases_of(w, cfg):
    D, F, V = cfg.dim, cfg.hidden_dim, cfg.vocab_size
    QD, KD = cfg.n_heads * cfg.head_dim, cfg.n_kv_heads * cfg.head_dim
    lw = w.layers[0]
    return [("wq", lw.wq, D, QD), ("wk", lw.wk, D, KD), ("wo", lw.wo, QD, D), ("w_gate", lw.w_gate, D, F),
            ("w_down", lw.w_down, F, D), ("lm_head", w.lm_head, D, V)]


def run(lib, path, tokens):
    w, cfg = capi.Weights(), capi.Config()
    lib.load_model(C.byref(w), C.byref(cfg), path.encode())
    out = {}
    for T in tokens:
        rr = np.random.Generator(np.random.PCG64(100 + T))
        for name, handle, K, N in cases_of(w, cfg):
            x = rr.standard_normal((T, K)).astype(np.float32)
            dx, do = lib.to_device(x), lib.empty_device(T * N * 4)
            lib.matmul(do, dx, handle, T, K, N)
            out[f"{name}_{T}"] = lib.to_host(do, (T, N))
            lib.free(dx)
            lib.free(do)
    return out


if __name__ == "__main__":
    model, dst = sys.argv[1], sys.argv[2]
    tokens = [int(t) for t in sys.argv[3].split(",")]
    lib = capi.load_hip()
    lib.init_ops()
    np.savez(dst, **run(lib, model, tokens))
