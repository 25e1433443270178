"""Synthetic model files, tokenizer files and request sets in the reference's formats.

No real checkpoints exist offline, so every test and benchmark runs on random-init
files of the published shapes.  The formats are the reference's:

* ``model.bin``: 7 int32 header (dim, hidden_dim, n_layers, n_heads, n_kv_heads,
  vocab_size, max_seq_len) followed by the tensors grouped by kind across layers
  in the order the loader reads them -- embed, rms_att[L], wq[L], wk[L], wv[L],
  wo[L], rms_ffn[L], w_gate[L], w_down[L], w_up[L], rms_final, (lm_head)
  (reference src/model.c:118,230-279).
* ``tokenizer.bin``: int32 max_token_length, then per token float32 score, int32
  length, bytes (reference src/tokenizer.c:43-77).

Values are small integers over a power of two, hence exactly representable in
fp16: an fp16 file and the fp32 expansion the reference makes of it hold the same
numbers.  Generation is seeded and chunked so that a 1.1 B-parameter file needs
little memory.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

SHAPES = {
    # name: dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, max_seq_len
    "tiny": (64, 160, 2, 4, 2, 512, 128),
    "tiny-mha48": (96, 224, 2, 2, 2, 384, 96),        # head_dim 48, like stories15M
    "tiny-gqa": (256, 512, 3, 8, 4, 600, 160),        # grouped heads that divide over 2 or 4 tensor-parallel ranks
    "stories15M": (288, 768, 6, 6, 6, 32000, 256),
    "stories42M": (512, 1376, 8, 8, 8, 32000, 1024),
    "tinyllama-1.1b": (2048, 5632, 22, 32, 4, 32000, 2048),
    "llama2-7b": (4096, 11008, 32, 32, 32, 32000, 2048),
}


@dataclass
class ModelShape:
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    max_seq_len: int

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_heads

    @classmethod
    def named(cls, name: str, **over) -> "ModelShape":
        s = cls(*SHAPES[name])
        for k, v in over.items():
            setattr(s, k, v)
        return s

    def header(self) -> bytes:
        return struct.pack("<7i", self.dim, self.hidden_dim, self.n_layers, self.n_heads,
                           self.n_kv_heads, self.vocab_size, self.max_seq_len)

    def n_params(self, shared: bool = False) -> int:
        D, F, L, V = self.dim, self.hidden_dim, self.n_layers, self.vocab_size
        QD, KD = self.n_heads * self.head_dim, self.n_kv_heads * self.head_dim
        body = V * D + L * (2 * D + D * QD + 2 * D * KD + QD * D + 3 * D * F) + D
        return body if shared else body + V * D


_CHUNK = 1 << 24


def _write_random(f, n: int, rng: np.random.Generator, span: int, denom: float, dtype, noise=None) -> None:
    """n values k/denom with k uniform in [-span, span], streamed in chunks.  With a
    `noise` generator every value also gets a relative perturbation of about 1e-4, which
    takes it off the fp16 grid (fp32 files only)."""
    done = 0
    while done < n:
        m = min(_CHUNK, n - done)
        k = rng.integers(-span, span + 1, size=m, dtype=np.int16)
        v = k.astype(np.float32) / np.float32(denom)
        if noise is not None:
            v = v + noise.standard_normal(m).astype(np.float32) * np.float32(span / denom * 1e-4)
        v.astype(dtype).tofile(f)
        done += m


def _write_norm(f, n: int, rng: np.random.Generator, dtype) -> None:
    """RMSNorm gains near one: 1 + k/64, k in [-8, 8]."""
    k = rng.integers(-8, 9, size=n, dtype=np.int16)
    (1.0 + k.astype(np.float32) / 64.0).astype(dtype).tofile(f)


def write_model(path: str, shape: ModelShape, seed: int = 0, dtype: str = "f16",
                shared: bool = False, rope_tail: bool = False, progress=None, exact16: bool = True) -> int:
    """Write a random-init model file; returns its size in bytes.

    Matrices: k/4096 with |k| <= 82 (about +-0.02); embedding (and a separate
    classifier): k/128 with |k| <= 64 (about +-0.5); norm gains 1 + k/64.
    ``rope_tail`` appends the legacy llama2.c RoPE tables (fp32 shared files only).
    ``progress(name)`` is called after every tensor kind (large files take a while).
    ``exact16=False`` (fp32 files) perturbs every matrix value so that fp16 cannot hold it,
    like the weights of a trained fp32 checkpoint.
    """
    progress = progress or (lambda name: None)
    np_dtype = {"f16": np.float16, "f32": np.float32}[dtype]
    D, F, L, V = shape.dim, shape.hidden_dim, shape.n_layers, shape.vocab_size
    QD, KD = shape.n_heads * shape.head_dim, shape.n_kv_heads * shape.head_dim
    rng = np.random.Generator(np.random.PCG64(seed))
    noise = None
    if not exact16:
        assert dtype == "f32", "only an fp32 file can hold values off the fp16 grid"
        noise = np.random.Generator(np.random.PCG64(seed + 104729))
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(shape.header())
        _write_random(f, V * D, rng, 64, 128.0, np_dtype, noise)       # token_embedding_table
        for _ in range(L):
            _write_norm(f, D, rng, np_dtype)                           # rms_att_weight
        for name, n in (("wq", D * QD), ("wk", D * KD), ("wv", D * KD), ("wo", QD * D)):
            for _ in range(L):
                _write_random(f, n, rng, 82, 4096.0, np_dtype, noise)
            progress(name)
        for _ in range(L):
            _write_norm(f, D, rng, np_dtype)                           # rms_ffn_weight
        for name, n in (("w_gate", D * F), ("w_down", F * D), ("w_up", D * F)):
            for _ in range(L):
                _write_random(f, n, rng, 82, 4096.0, np_dtype, noise)
            progress(name)
        _write_norm(f, D, rng, np_dtype)                               # rms_final_weight
        if not shared:
            _write_random(f, V * D, rng, 64, 2048.0, np_dtype, noise)  # lm_head
        if rope_tail:
            assert dtype == "f32" and shared
            np.zeros(shape.max_seq_len * shape.head_dim, np.float32).tofile(f)
    os.replace(tmp, path)
    return os.path.getsize(path)


def write_tokenizer(path: str, vocab_size: int) -> None:
    """A tokenizer file the reference's BPE encoder can digest: ids 0-2 specials,
    3-258 the byte tokens '<0xNN>', then ' ' and printable single characters, then
    unique filler strings (so decoded text is invertible to ids)."""
    pieces = ["<unk>", "<s>", "</s>"] + [f"<0x{b:02X}>" for b in range(256)]
    singles = [" "] + [chr(c) for c in range(33, 127)]
    pieces += singles
    i = 0
    while len(pieces) < vocab_size:
        pieces.append(f"▁t{i}")
        i += 1
    pieces = pieces[:vocab_size]
    enc = [p.encode("utf-8") for p in pieces]
    with open(path, "wb") as f:
        f.write(struct.pack("<i", max(len(e) for e in enc)))
        for e in enc:
            f.write(struct.pack("<fi", 0.0, len(e)))
            f.write(e)


def make_requests(n: int, prompt_lo: int, prompt_hi: int, vocab_size: int, seed: int = 0,
                  bos: int = 1):
    """n prompts of token ids in [3, vocab) with BOS first (reference
    src/tokenizer.c:136), lengths uniform in [prompt_lo, prompt_hi]."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    out = []
    for _ in range(n):
        ln = int(rng.integers(prompt_lo, prompt_hi + 1))
        p = rng.integers(3, vocab_size, size=ln, dtype=np.int64).astype(np.int32)
        p[0] = bos
        out.append(p)
    return out
