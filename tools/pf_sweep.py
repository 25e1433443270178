#!/usr/bin/env python3
"""Sweep the token count of one projection through the library's matmul entry point, for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/pf_sweep.py --which lm_head --tokens 1024,1408,1591
    python3 tools/pf_sweep.py --summarise OUT/*/*_kernel_trace.csv
The summary groups the projection kernels' launches by grid size (one group per token count).
The launches of one token count run back to back on the same weights: a matrix of tens of MB is then
served by the memory-side cache, which a forward pass (2 GB of weights per step) never sees -- fine for
prefill sizes, where the kernel is not bound by HBM; decode sizes must be confirmed in bench.py."""
import argparse
import csv
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(path):
    g = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "k_gemm" not in r["Kernel_Name"]:
            continue
        key = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_X"]))
        g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (wgs, wg), d in sorted(g.items()):
        d = sorted(d)
        print(f"workgroups {wgs:6d} x {wg:4d} threads  launches {len(d):3d}  median {d[len(d) // 2]:8.1f} us  min {d[0]:8.1f} us"
              f"  per workgroup-slot-round {d[len(d) // 2] / -(-wgs // 512):7.1f} us")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--summarise")
    ap.add_argument("--which", default="w_gate")
    ap.add_argument("--tokens", default="1024,1152,1280,1408,1536,1591,1664,2048")
    ap.add_argument("--model", default="tinyllama-1.1b")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--stamps", action="store_true",
                    help="diagnostic build (-DNVLLM_STAMPS): print where the workgroups of the last launch are in time")
    args = ap.parse_args()
    if args.summarise:
        return summarise(args.summarise)
    import numpy as np
    from nvllm_amd import capi, synth
    shape = synth.ModelShape.named(args.model, max_seq_len=2048)
    path = f"/tmp/nvllm_{args.model}_s2048_seed0.bin"
    if not os.path.exists(path):
        synth.write_model(path, shape, seed=0)
    lib = capi.load_hip()
    lib.nvllm_set_device(0)
    lib.init_ops()
    w, cfg = capi.Weights(), capi.Config()
    lib.load_model(C.byref(w), C.byref(cfg), path.encode())
    if args.which == "lm_head":
        handle, N, K = w.lm_head, cfg.vocab_size, cfg.dim
    elif args.which == "w_down":
        handle, N, K = w.layers[0].w_down, cfg.dim, cfg.hidden_dim
    elif args.which in ("wq", "wo"):
        handle, N, K = getattr(w.layers[0], args.which), cfg.n_heads * cfg.head_dim, cfg.dim
    elif args.which not in ("w_gate", "w_up"):
        raise SystemExit(f"unknown projection {args.which}")
    else:
        handle, N, K = getattr(w.layers[0], args.which), cfg.hidden_dim, cfg.dim
    rng = np.random.default_rng(0)
    STAMP_WGS = 1024
    if args.stamps:
        set_stamps = lib.dll.nvllm_debug_set_gemm_stamps
        set_stamps.argtypes = [C.c_void_p, C.c_int]
        sbuf = lib.empty_device(STAMP_WGS * 8 * 8)
        zeros = np.zeros(STAMP_WGS * 8, np.uint64)
    for T in [int(t) for t in args.tokens.split(",")]:
        x = rng.standard_normal((T, K), dtype=np.float32)
        dx, do = lib.to_device(x), lib.empty_device(T * N * 4)
        for _ in range(args.reps):
            lib.matmul(do, dx, handle, T, K, N)
        out = lib.to_host(do, (T, N))
        print(T, N, K, float(np.abs(out).mean()), flush=True)
        if args.stamps:
            lib.device_memcpy(sbuf, zeros.ctypes.data, zeros.nbytes, capi.HOST_TO_DEVICE)
            set_stamps(sbuf, 1)
            lib.matmul(do, dx, handle, T, K, N)
            lib.to_host(do, (T, N))
            set_stamps(None, 0)
            st = lib.to_host(sbuf, (STAMP_WGS, 8), np.uint64).astype(np.int64)
            wg_ids = np.nonzero(st[:, 0] > 0)[0]
            st = st[st[:, 0] > 0]
            t0 = st[:, 0].min()
            life = (st[:, 3] - st[:, 0]) / 100.0
            print("  loop-done percentiles (own clock) 5/25/50/75/95/100: " + " ".join(f"{np.percentile(life, q):.1f}" for q in (5, 25, 50, 75, 95, 100)))
            print("  median by blockIdx % 8: " + " ".join(f"{np.median(life[wg_ids % 8 == x]):.1f}" for x in range(8)))
            print("  max by blockIdx % 8:    " + " ".join(f"{life[wg_ids % 8 == x].max():.1f}" for x in range(8)))
            order = np.argsort(-life)[:12]
            print("  slowest: " + " ".join(f"{wg_ids[i]}:{life[i]:.0f}" for i in order))
            print(f"  {len(st)} workgroups; us after the first workgroup started: median (min..max); then per workgroup since its own start")
            for k, name in enumerate(["start", "prologue issued", "first stage landed", "loop done", "all waves done", "epilogue issued", "stores drained", "split closed / left"]):
                v = st[:, k]
                ok = v > 0
                if not ok.any():
                    continue
                a = (v[ok] - t0) / 100.0
                b = (v[ok] - st[ok, 0]) / 100.0
                print(f"  {k} {name:20s} {np.median(a):7.2f} ({a.min():6.2f}..{a.max():6.2f})   own {np.median(b):7.2f} ({b.min():6.2f}..{b.max():6.2f})")
        lib.free(dx)
        lib.free(do)


if __name__ == "__main__":
    main()
