"""ctypes view of the C ABI declared in include/*.h.

``Lib`` wraps one shared library that exports those symbols.  The product is the
HIP library (``load_hip``): its tensors live in HBM and are moved with
device_memcpy.  The same declarations also fit a host-memory build of the same
interface (kind other than 'hip': a "device pointer" is then just a numpy buffer);
the test suite uses that to drive its CPU checkers, which live outside this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

c_int_p = C.POINTER(C.c_int)
c_float_p = C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [("dim", C.c_int), ("hidden_dim", C.c_int), ("n_layers", C.c_int),
                ("n_heads", C.c_int), ("n_kv_heads", C.c_int), ("vocab_size", C.c_int),
                ("max_seq_len", C.c_int), ("head_dim", C.c_int), ("rope_theta", C.c_float)]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rms_att_weight", "wq", "wk", "wv", "wo",
                                          "rms_ffn_weight", "w_gate", "w_up", "w_down")]


class Weights(C.Structure):
    _fields_ = [("token_embedding_table", C.c_void_p), ("layers", C.POINTER(LayerWeights)),
                ("rms_final_weight", C.c_void_p), ("lm_head", C.c_void_p),
                ("_data_buffer", C.c_void_p)]


class RunState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v", "att",
                                          "logits", "key_cache", "value_cache")]


class KVCacheManager(C.Structure):
    _fields_ = [("block_size", C.c_int), ("num_blocks", C.c_int), ("free_blocks_count", C.c_int),
                ("free_block_indices", c_int_p), ("pool_k", C.c_void_p), ("pool_v", C.c_void_p)]


class BlockTable(C.Structure):
    _fields_ = [("block_indices", c_int_p), ("num_blocks", C.c_int)]


class ProbIndex(C.Structure):
    _fields_ = [("prob", C.c_float), ("index", C.c_int)]


class Sampler(C.Structure):
    _fields_ = [("state", C.c_ulonglong), ("temperature", C.c_float), ("topp", C.c_float),
                ("probindex", C.POINTER(ProbIndex)), ("vocab_size", C.c_int)]


class TpShard(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("rank", "world", "n_heads", "head0", "n_kv_heads", "kv_head0", "hidden", "hidden0",
                                         "vocab", "vocab0", "vocab_slice")]


# void fn(void* ctx, float* device_buf, size_t count, void* hip_stream)
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
TP_ID_BYTES = 128
TP_PEER_HANDLE_BYTES = 64


class EngineOptions(C.Structure):
    _fields_ = [("chunk_size", C.c_int), ("block_size", C.c_int), ("num_blocks", C.c_int),
                ("max_seq_len", C.c_int), ("raise_to_1024", C.c_int), ("recycle_blocks", C.c_int),
                ("max_steps", C.c_int), ("temperature", C.c_float), ("topp", C.c_float),
                ("seed", C.c_ulonglong)]


class EngineStats(C.Structure):
    _fields_ = [("steps", C.c_longlong), ("tokens_forwarded", C.c_longlong),
                ("prefill_tokens", C.c_longlong), ("decode_tokens", C.c_longlong),
                ("outputs", C.c_longlong), ("forward_seconds", C.c_double),
                ("sample_seconds", C.c_double), ("prefill_step_seconds", C.c_double),
                ("decode_step_seconds", C.c_double), ("decode_step_tokens", C.c_longlong),
                ("prefill_step_tokens", C.c_longlong)]


_V, _I, _F, _P, _L = None, C.c_int, C.c_float, C.c_void_p, C.c_long

# name -> (restype, argtypes); present in all three libraries
_COMMON = {
    "init_ops": (_V, []),
    "free_ops": (_V, []),
    "rms_norm": (_V, [_P, _P, _P, _I, _I, _F]),
    "matmul": (_V, [_P, _P, _P, _I, _I, _I]),
    "apply_rope": (_V, [_P, _P, _I, _F, _I, _I, _I, _I]),
    "apply_rope_batch": (_V, [_P, _P, c_int_p, _F, _I, _I, _I, _I]),
    "swiglu": (_V, [_P, _P, _P, _I, _I]),
    "softmax": (_V, [_P, _I]),
    "accum": (_V, [_P, _P, _I]),
    "update_kv_cache": (_V, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "multi_head_attention": (_V, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "init_kv_cache_manager": (_V, [C.POINTER(KVCacheManager), _I, _I, _I, _I, _I]),
    "free_kv_cache_manager": (_V, [C.POINTER(KVCacheManager)]),
    "alloc_block": (_I, [C.POINTER(KVCacheManager)]),
    "free_block": (_V, [C.POINTER(KVCacheManager), _I]),
    "get_physical_offset": (_L, [C.POINTER(KVCacheManager), _I, _I, _I, _I, _I]),
    "update_kv_cache_paged": (_V, [C.POINTER(KVCacheManager), C.POINTER(BlockTable), _P, _P,
                                   _I, _I, _I, _I, _I]),
    "update_kv_cache_paged_batch": (_V, [C.POINTER(KVCacheManager), C.POINTER(C.POINTER(BlockTable)),
                                         c_int_p, _P, _P, _I, c_int_p, _I, _I, _I]),
    "paged_attention": (_V, [_P, _P, C.POINTER(KVCacheManager), C.POINTER(BlockTable), _P,
                             _I, _I, _I, _I, _I, _I, _I]),
    "paged_attention_batch": (_V, [_P, _P, C.POINTER(KVCacheManager),
                                   C.POINTER(C.POINTER(BlockTable)), c_int_p, _P, _I, c_int_p,
                                   _I, _I, _I, _I, _I]),
    "load_model": (_V, [C.POINTER(Weights), C.POINTER(Config), C.c_char_p]),
    "malloc_run_state": (_V, [C.POINTER(RunState), C.POINTER(Config)]),
    "free_run_state": (_V, [C.POINTER(RunState)]),
    "transformer_batch": (_V, [c_int_p, _I, c_int_p, c_int_p, c_int_p, _I, C.POINTER(Config),
                               C.POINTER(RunState), C.POINTER(Weights),
                               C.POINTER(C.POINTER(BlockTable))]),
    "build_sampler": (_V, [C.POINTER(Sampler), _I, _F, _F, C.c_ulonglong]),
    "sample": (_I, [C.POINTER(Sampler), _P]),
    # N-sequence driver (host/engine.c is compiled into every library)
    "nvllm_engine_create": (_P, [C.c_char_p, _I, C.POINTER(EngineOptions)]),
    "nvllm_engine_destroy": (_V, [_P]),
    "nvllm_engine_set_request": (_I, [_P, _I, c_int_p, _I, _I, _I]),
    "nvllm_engine_skip_prefill": (_I, [_P, _I, _I]),
    "nvllm_engine_step": (_I, [_P]),
    "nvllm_engine_run": (_I, [_P, _I]),
    "nvllm_engine_global_step": (_I, [_P]),
    "nvllm_engine_get_stats": (_V, [_P, C.POINTER(EngineStats)]),
    "nvllm_engine_reset_stats": (_V, [_P]),
    "nvllm_engine_config": (C.POINTER(Config), [_P]),
    "nvllm_engine_seq_state": (_I, [_P, _I, c_int_p, c_int_p, c_int_p, c_int_p]),
    "nvllm_engine_history": (_I, [_P, _I, c_int_p, _I]),
    "nvllm_engine_block_table": (_I, [_P, _I, c_int_p, _I]),
    "nvllm_engine_last_batch": (_I, [_P, c_int_p, c_int_p, c_int_p, _I, c_int_p, c_int_p, _I, c_int_p]),
    "nvllm_engine_logits": (c_float_p, [_P]),
    "nvllm_engine_alloc_trace": (_I, [_P, c_int_p, _I]),
}

# only in the HIP library (include/backend.h, include/nvllm.h)
_HIP_ONLY = {
    "device_malloc": (_I, [C.POINTER(C.c_void_p), C.c_size_t]),
    "device_free": (_V, [_P]),
    "device_memcpy": (_I, [_P, _P, C.c_size_t, _I]),
    "device_sync": (_V, []),
    "transformer": (_V, [c_int_p, _I, _I, C.POINTER(Config), C.POINTER(RunState),
                         C.POINTER(Weights), C.POINTER(BlockTable)]),
    "nvllm_set_device": (_I, [_I]),
    "nvllm_stream": (_P, []),
    "nvllm_set_logits_mode": (_V, [_I]),
    "nvllm_get_logits_mode": (_I, []),
    "nvllm_last_argmax": (c_int_p, [c_int_p]),
    "nvllm_argmax_rows": (_I, [c_int_p, _P, _I, _I, _I]),
    "nvllm_forward_enqueue": (_L, [c_int_p, _I, c_int_p, c_int_p, c_int_p, _I, c_int_p, C.POINTER(Config),
                                   C.POINTER(RunState), C.POINTER(Weights), C.POINTER(C.POINTER(BlockTable))]),
    "nvllm_forward_collect": (_I, [_L, c_int_p, _I]),
    "nvllm_set_kv_elem_bytes": (_V, [_I]),
    "nvllm_get_kv_elem_bytes": (_I, []),
    "nvllm_prof_enable": (_V, [_I]),
    "nvllm_prof_reset": (_V, []),
    "nvllm_prof_num_classes": (_I, []),
    "nvllm_prof_name": (C.c_char_p, [_I]),
    "nvllm_prof_get": (_V, [_I, C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                            C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nvllm_free_model": (_V, [C.POINTER(Weights)]),
    "nvllm_matmul_noalloc": (_I, [_P, _P, _P, _I, _I, _I]),
    "nvllm_set_decode_chain": (_V, [_I]),
    "nvllm_get_decode_chain": (_I, []),
    "nvllm_decode_chain_launches": (_L, []),
    "nvllm_tp_shard": (_I, [C.POINTER(Config), _I, _I, C.POINTER(TpShard)]),
    "nvllm_tp_unique_id": (_I, [_P]),
    "nvllm_tp_init": (_I, [_I, _I, _P]),
    "nvllm_tp_init_custom": (_I, [_I, _I, ALLREDUCE_FN, _P]),
    "nvllm_tp_peer_handle": (_I, [_P]),
    "nvllm_tp_init_peer": (_I, [_I, _I, _P]),
    "nvllm_tp_peer_stats": (_V, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "nvllm_tp_allreduce": (_I, [_P, C.c_size_t]),
    "nvllm_tp_finalize": (_V, []),
    "nvllm_tp_rank": (_I, []),
    "nvllm_tp_world": (_I, []),
}

DEVICE_TO_HOST, DEVICE_TO_DEVICE, HOST_TO_DEVICE = 0, 1, 2

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
HIP_LIB = os.environ.get("NVLLM_HIP_LIB") or os.path.join(PKG_DIR, "libnvllm_hip.so")   # the override serves A/B builds of the same library


def declared_symbols(hip: bool = True):
    names = list(_COMMON)
    if hip:
        names += list(_HIP_ONLY)
    return names


class Lib:
    """One loaded library plus tensor helpers.  kind 'hip' = tensors in HBM; any other
    kind names a host-memory library with the same symbols."""

    def __init__(self, path: str, kind: str):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.kind = kind
        self.path = path
        self.dll = C.CDLL(path, mode=C.RTLD_LOCAL)
        table = dict(_COMMON)
        if kind == "hip":
            table.update(_HIP_ONLY)
        for name, (res, args) in table.items():
            fn = getattr(self.dll, name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        self._keep = []

    # ---- tensors -------------------------------------------------------------
    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        if self.kind != "hip":
            buf = arr.copy()
            self._keep.append(buf)
            return buf.ctypes.data
        p = C.c_void_p()
        if self.device_malloc(C.byref(p), max(arr.nbytes, 4)) != 0:
            raise MemoryError("device_malloc failed")
        if arr.nbytes:
            self.device_memcpy(p, arr.ctypes.data, arr.nbytes, HOST_TO_DEVICE)
        return p.value

    def empty_device(self, nbytes: int) -> int:
        return self.to_device(np.zeros(max(nbytes, 4), np.uint8))

    def to_host(self, ptr: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype)
        if self.kind != "hip":
            C.memmove(out.ctypes.data, ptr, out.nbytes)
        elif out.nbytes:
            self.device_memcpy(out.ctypes.data, ptr, out.nbytes, DEVICE_TO_HOST)
        return out

    def free(self, ptr: int) -> None:
        if self.kind == "hip":
            self.device_free(ptr)

    def release(self) -> None:
        self._keep.clear()


def ints(values) -> "C.Array":
    a = np.ascontiguousarray(values, dtype=np.int32)
    return (C.c_int * len(a))(*a.tolist())


def load_hip() -> Lib:
    """The product library.  Raises when it has not been built: there is no CPU
    fall-back for the forward pass."""
    return Lib(HIP_LIB, "hip")


class Engine:
    """Thin object wrapper over the nvllm_engine_* entry points."""

    def __init__(self, lib: Lib, model_path: str, n_seqs: int, **opts):
        self.lib = lib
        o = EngineOptions()
        for k, v in opts.items():
            setattr(o, k, v)
        self.h = lib.nvllm_engine_create(model_path.encode(), n_seqs, C.byref(o))
        if not self.h:
            raise RuntimeError("nvllm_engine_create failed")
        self.n_seqs = n_seqs
        self.config = lib.nvllm_engine_config(self.h).contents

    def set_request(self, seq, prompt, n_new, arrival_step=0):
        p = np.ascontiguousarray(prompt, dtype=np.int32)
        rc = self.lib.nvllm_engine_set_request(self.h, seq, p.ctypes.data_as(c_int_p), len(p),
                                               n_new, arrival_step)
        if rc != 0:
            raise ValueError(f"bad request for seq {seq}")

    def skip_prefill(self, seq, upto) -> int:
        """Measurement aid: start `seq` at position `upto` with its blocks allocated and the pool rows as they are."""
        rc = self.lib.nvllm_engine_skip_prefill(self.h, seq, upto)
        if rc < 0:
            raise ValueError(f"cannot skip the prefill of seq {seq}")
        return rc

    def step(self) -> int:
        return self.lib.nvllm_engine_step(self.h)

    def run(self, n: int) -> int:
        return self.lib.nvllm_engine_run(self.h, n)

    def run_all(self, cap: int = 100000) -> int:
        return self.lib.nvllm_engine_run(self.h, cap)

    def stats(self) -> EngineStats:
        s = EngineStats()
        self.lib.nvllm_engine_get_stats(self.h, C.byref(s))
        return s

    def reset_stats(self):
        self.lib.nvllm_engine_reset_stats(self.h)

    def history(self, seq) -> np.ndarray:
        cap = self.config.max_seq_len + 1
        buf = (C.c_int * cap)()
        n = self.lib.nvllm_engine_history(self.h, seq, buf, cap)
        return np.array(buf[:min(n, cap)], dtype=np.int32)

    def block_table(self, seq) -> np.ndarray:
        cap = self.config.max_seq_len
        buf = (C.c_int * cap)()
        n = self.lib.nvllm_engine_block_table(self.h, seq, buf, cap)
        return np.array(buf[:min(n, cap)], dtype=np.int32)

    def seq_state(self, seq):
        st, cur, npr, sl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        pos = self.lib.nvllm_engine_seq_state(self.h, seq, C.byref(st), C.byref(cur), C.byref(npr),
                                              C.byref(sl))
        return dict(pos=pos, status=st.value, current_token=cur.value, n_prompt=npr.value,
                    seq_len=sl.value)

    def last_batch(self):
        cap = self.config.max_seq_len
        tok, pos, sid = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
        oi, smp = (C.c_int * self.n_seqs)(), (C.c_int * self.n_seqs)()
        no = C.c_int()
        n = self.lib.nvllm_engine_last_batch(self.h, tok, pos, sid, cap, oi, smp, self.n_seqs,
                                             C.byref(no))
        return dict(tokens=np.array(tok[:n]), pos=np.array(pos[:n]), seq_ids=np.array(sid[:n]),
                    output_indices=np.array(oi[:no.value]), sampled=np.array(smp[:no.value]))

    def logits(self, num_outputs: int) -> np.ndarray:
        p = self.lib.nvllm_engine_logits(self.h)
        V = self.config.vocab_size
        return np.ctypeslib.as_array(p, shape=(num_outputs, V)).copy()

    def alloc_trace(self) -> np.ndarray:
        n = self.lib.nvllm_engine_alloc_trace(self.h, None, 0)
        buf = (C.c_int * (3 * max(n, 1)))()
        self.lib.nvllm_engine_alloc_trace(self.h, buf, n)
        return np.array(buf[:3 * n], dtype=np.int32).reshape(n, 3)

    def close(self):
        if self.h:
            self.lib.nvllm_engine_destroy(self.h)
            self.h = None
