#!/usr/bin/env python3
"""Timeline inside the decode projection kernel, from a diagnostic build of the library
(make EXTRA=-DNVLLM_STAMPS; NVLLM_HIP_LIB points at it): wave 0 of every workgroup reads the
100 MHz constant-rate clock at seven points.  Runs the benchmark's batch (64 decode rows) on a
TinyLlama-width model with few layers and prints, per launch of one decode step, where the
workgroups are in time (microseconds after the launch's first workgroup started):
  0 start   1 loads of the first K-steps issued   2 first K-step multiplied (its data arrived)
  3 K loop done   4 waves' sums added through LDS   5 K split closed   6 stores drained
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nvllm_amd import capi, synth  # noqa: E402

STAMP_WGS = 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt-len", type=int, default=64)
    args = ap.parse_args()
    lib = capi.load_hip()
    lib.init_ops()
    lib.nvllm_set_kv_elem_bytes(2)
    lib.nvllm_set_logits_mode(1)
    set_stamps = lib.dll.nvllm_debug_set_gemm_stamps
    set_stamps.argtypes = [C.c_void_p, C.c_int]
    shape = synth.ModelShape.named("tinyllama-1.1b", n_layers=args.layers, max_seq_len=512)
    path = "/tmp/nvllm_stamps_model.bin"
    synth.write_model(path, shape, seed=0)
    prompts = synth.make_requests(args.batch, args.prompt_len, args.prompt_len, shape.vocab_size, seed=1)
    eng = capi.Engine(lib, path, args.batch, chunk_size=512, temperature=0.0, max_seq_len=512, max_steps=1 << 30)
    for i, p in enumerate(prompts):
        eng.set_request(i, p, 64, 0)
    for _ in range(12):          # prefill and a few decode steps
        eng.step()
    cap = 4 * args.layers + 1
    nwords = cap * STAMP_WGS * 8
    buf = C.c_void_p()
    assert lib.device_malloc(C.byref(buf), nwords * 8) == 0
    zeros = np.zeros(nwords, np.uint64)
    results = []
    for rep in range(5):
        lib.device_memcpy(buf, zeros.ctypes.data, zeros.nbytes, capi.HOST_TO_DEVICE)
        set_stamps(buf, cap)
        eng.step()
        lib.device_sync()
        set_stamps(None, 0)
        results.append(lib.to_host(buf.value, (cap, STAMP_WGS, 8), np.uint64).astype(np.int64))
    names = (["qkv", "wo", "gate/up", "down"] * args.layers) + ["head"]
    st = results[-1]
    print(f"{'launch':10s} {'wgs':>4s} | stamp: median (min..max) us after the launch's first workgroup started")
    for i in range(cap):
        s = st[i]
        live = s[:, 0] > 0
        if not live.any():
            continue
        s = s[live].astype(np.float64)
        t0 = s[:, 0].min()
        rel = (s - t0) / 100.0       # 100 MHz ticks -> us
        cols = []
        for k in range(7):
            v = rel[:, k]
            v = v[s[:, k] > 0]
            cols.append(f"{k}: {np.median(v):5.2f} ({v.min():5.2f}..{v.max():5.2f})" if v.size else f"{k}: -")
        print(f"{names[i]:10s} {int(live.sum()):4d} | " + "  ".join(cols))
        # per-workgroup durations of the phases (median over workgroups)
        # (a stamp a workgroup never wrote -- those that leave before the K split is closed write no 5 and 6 -- is
        # zero in the buffer: it takes no part in any figure)
        rel = np.where(s > 0, rel, np.nan)
        d = np.diff(rel, axis=1)
        med = lambda a: f"{np.nanmedian(a):.2f}" if np.isfinite(a).any() else "-"
        print(f"{'':15s} | phase medians: issue {med(d[:,0])}  first-data {med(d[:,1])}  loop {med(d[:,2])}  "
              f"lds-sum {med(d[:,3])}  close {med(d[:,4])}  epilogue+drain {med(d[:,5])}  | wg life {med(rel[:,6]-rel[:,0])}")
    eng.close()
    # ---- the same projection four times in a row on the same weights: from the second launch on
    # they come from the caches (an upper bound on what warming them ahead of a launch could buy)
    w, cfg = capi.Weights(), capi.Config()
    lib.load_model(C.byref(w), C.byref(cfg), path.encode())
    r = np.random.Generator(np.random.PCG64(1))
    for name, handle, K, N in (("wo", w.layers[0].wo, cfg.dim, cfg.dim), ("w_down", w.layers[0].w_down, cfg.hidden_dim, cfg.dim),
                               ("w_gate", w.layers[0].w_gate, cfg.dim, cfg.hidden_dim)):
        x = r.standard_normal((args.batch, K)).astype(np.float32)
        dx, do = lib.to_device(x), lib.empty_device(args.batch * N * 4)
        big = lib.empty_device(512 << 20)
        lib.device_memcpy(big, big, 512 << 20, capi.DEVICE_TO_DEVICE)       # push the weights out of the caches
        lib.device_sync()
        lib.device_memcpy(buf, zeros.ctypes.data, zeros.nbytes, capi.HOST_TO_DEVICE)
        set_stamps(buf, cap)
        for _ in range(4):
            lib.matmul(do, dx, handle, args.batch, K, N)
        lib.device_sync()
        set_stamps(None, 0)
        st = lib.to_host(buf.value, (cap, STAMP_WGS, 8), np.uint64).astype(np.int64)
        for i in range(4):
            sl = st[i]
            live = sl[:, 0] > 0
            sl = sl[live].astype(np.float64)
            rel = np.where(sl > 0, (sl - sl[:, 0].min()) / 100.0, np.nan)      # unwritten stamps take no part
            med = [float(np.nanmedian(rel[:, k])) if np.isfinite(rel[:, k]).any() else float("nan") for k in range(7)]
            print(f"matmul {name:7s} launch {i} wgs {int(live.sum()):3d} | " + "  ".join(f"{k}: {m:5.2f}" for k, m in enumerate(med))
                  + f" | end {np.nanmax(rel):5.2f}")
        for q in (dx, do, big):
            lib.free(q)


if __name__ == "__main__":
    main()
