#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DGEMM_STAMPS): run a few decode steps and
print, for one projection class, where its waves spend their cycles."""
import ctypes as C, os, sys, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvllm_amd import capi, synth
cls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
os.environ["NVLLM_STAMP_CLASS"] = str(cls)
lib = capi.load_hip()
lib.nvllm_set_logits_mode(1)
shape = synth.ModelShape.named("tinyllama-1.1b", max_seq_len=2048)
path = "/tmp/nvllm_tinyllama-1.1b_s2048_seed0.bin"
if not os.path.exists(path): synth.write_model(path, shape, seed=0)
B = 64
eng = capi.Engine(lib, path, B, chunk_size=512, temperature=0.0, max_seq_len=2048, max_steps=1 << 30)
for i, p in enumerate(synth.make_requests(B, 64, 64, shape.vocab_size, seed=11)): eng.set_request(i, p, 40, 0)
for _ in range(12): eng.step()
fn = lib.dll.nvllm_debug_dump_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_char_p]
n = fn(b"/tmp/stamps.bin")
raw = open("/tmp/stamps.bin", "rb").read()
wgs = struct.unpack("<i", raw[:4])[0]
st = np.frombuffer(raw[4:], np.uint64).reshape(wgs, 8, 8).astype(np.int64)
names = ["entry", "loads issued", "main loop done", "lds reduce done", "slab drained", "ticket known", "epilogue start", "end"]
print(f"class {cls}: {wgs} workgroups; per-wave phase durations in shader clocks (s_memtime ticks are per XCD, so only differences inside one workgroup are meaningful)")
for k in range(1, 8):
    ok = (st[:, :, k] > 0) & (st[:, :, k - 1] > 0)
    d = (st[:, :, k] - st[:, :, k - 1])[ok]
    if d.size: print(f"  {names[k-1]:16s} -> {names[k]:16s} n={d.size:5d}  med {np.median(d):8.0f}  p10 {np.percentile(d,10):8.0f}  p90 {np.percentile(d,90):8.0f}  max {d.max():8.0f}")
life = []
for w in range(wgs):
    e = st[w, :, 0]; ends = st[w].max(axis=1)
    life.append(ends.max() - e[e > 0].min())
life = np.array(life)
print(f"  workgroup lifetime: med {np.median(life):.0f}  p90 {np.percentile(life,90):.0f}  max {life.max():.0f} clocks")
