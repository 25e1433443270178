#!/usr/bin/env python3
"""Timeline inside the one-launch decode step, from a diagnostic build of the library
(make BUILD=build_stamps OUT=variants/libnvllm_stamps.so EXTRA=-DNVLLM_STAMPS; NVLLM_HIP_LIB points
at it).  Thread 0 of every workgroup reads the 100 MHz constant-rate clock at eight points of every
phase of every layer:
  projection phases  0 dependency met, tile starts   1 activations requested, 1/rms gathered (first ring landed)
                     2 first K-step multiplied   3 K loop done (next phase's weights requested right after)
                     4 waves' sums added through LDS   5 K split closed   6 epilogue done, stores drained
                     7 arrival published
  attention          0 dependency met   1 q loaded, first pages requested   3 page loop done
                     6 merged and stored   7 arrival published
Prints, per phase of one middle layer, medians over workgroups in microseconds after the first
workgroup of the layer started its q/k/v tile, the time workgroups spend waiting for the phase's
dependency (stamp 0 minus the previous phase's stamp 7), and the layer's span.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nvllm_amd import capi, synth  # noqa: E402

PHASES = ["qkv", "attention", "wo", "gate/up", "down"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--context", type=int, default=1040)
    ap.add_argument("--model", default="tinyllama-1.1b")
    ap.add_argument("--show-layer", type=int, default=3)
    args = ap.parse_args()
    lib = capi.load_hip()
    lib.init_ops()
    lib.nvllm_set_kv_elem_bytes(2)
    lib.nvllm_set_logits_mode(1)
    set_stamps = lib.dll.nvllm_debug_set_chain_stamps
    set_stamps.argtypes = [C.c_void_p]
    msl = max(1024, args.context + 128)
    shape = synth.ModelShape.named(args.model, n_layers=args.layers, max_seq_len=msl)
    path = f"/tmp/nvllm_chain_stamps_{args.model}_{args.layers}.bin"
    synth.write_model(path, shape, seed=0)
    prompts = synth.make_requests(args.batch, args.context + 8, args.context + 8, shape.vocab_size, seed=1)
    eng = capi.Engine(lib, path, args.batch, chunk_size=512, temperature=0.0, max_seq_len=msl, max_steps=1 << 30)
    for i, p in enumerate(prompts):
        eng.set_request(i, p, 64, 0)
        eng.skip_prefill(i, args.context)
    for _ in range(12):
        eng.step()
    G = 256
    nwords = args.layers * 5 * G * 8
    buf = C.c_void_p()
    assert lib.device_malloc(C.byref(buf), nwords * 8) == 0
    zeros = np.zeros(nwords, np.uint64)
    lib.nvllm_set_decode_chain(1)
    before = lib.nvllm_decode_chain_launches()
    for rep in range(4):
        lib.device_memcpy(buf, zeros.ctypes.data, zeros.nbytes, capi.HOST_TO_DEVICE)
        set_stamps(buf)
        eng.step()
        eng.step()      # (the driver collects a pass one step later)
        lib.device_sync()
        set_stamps(None)
    assert lib.nvllm_decode_chain_launches() > before, "the steps did not take the one-launch form"
    st = lib.to_host(buf.value, (args.layers, 5, G, 8), np.uint64).astype(np.int64).astype(np.float64) / 100.0   # us
    st[st == 0] = np.nan
    L = min(args.show_layer, args.layers - 1)
    t0 = np.nanmin(st[L, 0, :, 0])
    print(f"model {args.model}, {args.batch} rows at context {args.context}; layer {L} of {args.layers}; "
          f"us after the layer's first q/k/v tile started; median (min..max) over workgroups")
    prev7 = st[L - 1, 4, :, 7] if L > 0 else None
    for ph, name in enumerate(PHASES):
        s = st[L, ph] - t0
        s[s < -20.0] = np.nan     # left by the step before (a workgroup that does not close a K split writes no 5 and 6)
        live = ~np.isnan(s[:, 0])
        cols = []
        for k in range(8):
            v = s[live, k]
            v = v[~np.isnan(v)]
            cols.append(f"{k}: {np.median(v):6.2f} ({v.min():6.2f}..{v.max():6.2f})" if v.size else f"{k}:   -")
        print(f"{name:10s} {int(live.sum()):4d} wgs | " + "  ".join(cols))
        if prev7 is not None:
            w = st[L, ph, :, 0] - prev7
            w = w[~np.isnan(w)]
            if w.size:
                print(f"{'':19s}| waited for the dependency (own previous arrival -> this tile's start): median {np.median(w):5.2f}  max {w.max():5.2f}")
        prev7 = st[L, ph, :, 7]
    spans = []
    for l in range(1, args.layers):
        spans.append(np.nanmax(st[l, 4, :, 7]) - np.nanmax(st[l - 1, 4, :, 7]))
    print("layer spans (last down arrival to last down arrival):", " ".join(f"{x:.1f}" for x in spans), "us")
    # when do a phase's first weight bytes land relative to the previous phase's last store?
    for ph in (2, 3, 4):
        first_data = st[L, ph, :, 1]
        prev_end = np.nanmax(st[L, ph - 1, :, 6])
        print(f"{PHASES[ph]:10s}: first ring landed {np.nanmedian(first_data) - prev_end:+.2f} us (median) relative to the LAST store drain of {PHASES[ph - 1]}")
    eng.close()


if __name__ == "__main__":
    main()
