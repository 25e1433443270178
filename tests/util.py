"""Helpers shared by the tests: call the operators of include/ops.h on numpy data
through any of the three libraries (HIP product, CPU oracle, reference build)."""
import ctypes as C

import numpy as np

from nvllm_amd import capi


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Pool:
    """A KVCacheManager plus per-sequence block tables living in Python."""

    def __init__(self, lib, block_size, num_blocks, n_layers, n_kv_heads, head_dim):
        self.lib = lib
        self.mgr = capi.KVCacheManager()
        self.geom = (n_layers, n_kv_heads, head_dim)
        lib.init_kv_cache_manager(C.byref(self.mgr), block_size, num_blocks, n_layers, n_kv_heads, head_dim)
        self.tables = []

    def new_table(self, max_blocks):
        arr = (C.c_int * max_blocks)()
        bt = capi.BlockTable(C.cast(arr, capi.c_int_p), 0)
        self.tables.append((bt, arr))
        return len(self.tables) - 1

    def grow(self, seq, n_positions):
        """Allocate blocks so that positions [0, n_positions) are covered."""
        bt, arr = self.tables[seq]
        bs = self.mgr.block_size
        while bt.num_blocks * bs < n_positions:
            b = self.lib.alloc_block(C.byref(self.mgr))
            assert b >= 0
            arr[bt.num_blocks] = b
            bt.num_blocks += 1

    def set_table(self, seq, blocks):
        bt, arr = self.tables[seq]
        for i, b in enumerate(blocks):
            arr[i] = b
        bt.num_blocks = len(blocks)

    def table_ptrs(self):
        n = len(self.tables)
        ptrs = (C.POINTER(capi.BlockTable) * n)()
        for i, (bt, _) in enumerate(self.tables):
            ptrs[i] = C.pointer(bt)
        return ptrs

    def elems(self):
        L, KVH, hd = self.geom
        return L * self.mgr.num_blocks * self.mgr.block_size * KVH * hd

    # ---- fp16 pools of the HIP library are stored in page order: inside a block one
    # page per kv head, elements in matrix-core operand order (include/nvllm.h,
    # nvllm_set_kv_elem_bytes).  These helpers translate to / from the reference's
    # logical layout [layer][block][position][kv head][dim].
    def _page_offsets(self, which):
        L, KVH, hd = self.geom
        key = np.arange(16)[:, None]
        d = np.arange(hd)[None, :]
        if which == "k":
            return (d >> 4) * 256 + ((((d >> 2) & 3) * 16 + key) << 2) + (d & 3)
        return (d >> 4) * 256 + (((key >> 2) * 16 + (d & 15)) << 2) + (key & 3)

    def _is_fp16(self):
        return self.lib.kind == "hip" and self.lib.nvllm_get_kv_elem_bytes() == 2

    def dump(self, which="k"):
        """Pool contents as fp32 in the reference's logical layout."""
        p = self.mgr.pool_k if which == "k" else self.mgr.pool_v
        if not self._is_fp16():
            return self.lib.to_host(p, (self.elems(),), np.float32)
        L, KVH, hd = self.geom
        nb, bs = self.mgr.num_blocks, self.mgr.block_size
        assert bs == 16
        raw = self.lib.to_host(p, (L, nb, KVH, 16 * hd), np.float16)
        off = self._page_offsets(which)                       # [16, hd]
        logical = raw[:, :, :, off]                           # [L, nb, KVH, 16, hd]
        return np.ascontiguousarray(logical.transpose(0, 1, 3, 2, 4)).astype(np.float32).reshape(-1)

    def fill(self, k_vals, v_vals):
        """Overwrite both pools with host data given in the logical layout."""
        if self.lib.kind != "hip":
            C.memmove(self.mgr.pool_k, f32(k_vals).ctypes.data, k_vals.size * 4)
            C.memmove(self.mgr.pool_v, f32(v_vals).ctypes.data, v_vals.size * 4)
            return
        if not self._is_fp16():
            self.lib.device_memcpy(self.mgr.pool_k, f32(k_vals).ctypes.data, k_vals.size * 4, capi.HOST_TO_DEVICE)
            self.lib.device_memcpy(self.mgr.pool_v, f32(v_vals).ctypes.data, v_vals.size * 4, capi.HOST_TO_DEVICE)
            return
        L, KVH, hd = self.geom
        nb = self.mgr.num_blocks
        for which, vals, dst in (("k", k_vals, self.mgr.pool_k), ("v", v_vals, self.mgr.pool_v)):
            logical = np.asarray(vals, dtype=np.float16).reshape(L, nb, 16, KVH, hd).transpose(0, 1, 3, 2, 4)
            raw = np.zeros((L, nb, KVH, 16 * hd), np.float16)
            raw[:, :, :, self._page_offsets(which)] = logical
            self.lib.device_memcpy(dst, raw.ctypes.data, raw.nbytes, capi.HOST_TO_DEVICE)

    def close(self):
        self.lib.free_kv_cache_manager(C.byref(self.mgr))


def op_rms_norm(lib, x, w, eps=1e-5, inplace=False):
    T, D = x.shape
    dx, dw = lib.to_device(f32(x)), lib.to_device(f32(w))
    do = dx if inplace else lib.empty_device(x.nbytes)
    lib.rms_norm(do, dx, dw, D, T, eps)
    out = lib.to_host(do, (T, D))
    for p in {dx, dw, do}:
        lib.free(p)
    return out


def op_matmul(lib, x, w):
    T, K = x.shape
    N = w.shape[0]
    dx, dw, do = lib.to_device(f32(x)), lib.to_device(f32(w)), lib.empty_device(T * N * 4)
    lib.matmul(do, dx, dw, T, K, N)
    out = lib.to_host(do, (T, N))
    for p in (dx, dw, do):
        lib.free(p)
    return out


def op_rope_batch(lib, q, k, pos, theta, head_dim, n_heads, n_kv_heads):
    T = q.shape[0]
    dq, dk = lib.to_device(f32(q)), lib.to_device(f32(k))
    lib.apply_rope_batch(dq, dk, capi.ints(pos), theta, head_dim, T, n_heads, n_kv_heads)
    oq, ok = lib.to_host(dq, q.shape), lib.to_host(dk, k.shape)
    lib.free(dq)
    lib.free(dk)
    return oq, ok


def op_rope(lib, q, k, pos0, theta, head_dim, n_heads, n_kv_heads):
    T = q.shape[0]
    dq, dk = lib.to_device(f32(q)), lib.to_device(f32(k))
    lib.apply_rope(dq, dk, pos0, theta, head_dim, T, n_heads, n_kv_heads)
    oq, ok = lib.to_host(dq, q.shape), lib.to_host(dk, k.shape)
    lib.free(dq)
    lib.free(dk)
    return oq, ok


def op_swiglu(lib, g, u, inplace=True):
    T, F = g.shape
    dg, du = lib.to_device(f32(g)), lib.to_device(f32(u))
    do = dg if inplace else lib.empty_device(g.nbytes)
    lib.swiglu(do, dg, du, F, T)
    out = lib.to_host(do, (T, F))
    for p in {dg, du, do}:
        lib.free(p)
    return out


def op_accum(lib, a, b):
    da, db = lib.to_device(f32(a)), lib.to_device(f32(b))
    lib.accum(da, db, a.size)
    out = lib.to_host(da, a.shape)
    lib.free(da)
    lib.free(db)
    return out


def op_softmax(lib, x):
    dx = lib.to_device(f32(x))
    lib.softmax(dx, x.size)
    out = lib.to_host(dx, x.shape)
    lib.free(dx)
    return out


def op_kv_write_batch(lib, pool, k, v, layer, pos, seq_ids, n_kv_heads, head_dim):
    T = k.shape[0]
    dk, dv = lib.to_device(f32(k)), lib.to_device(f32(v))
    lib.update_kv_cache_paged_batch(C.byref(pool.mgr), pool.table_ptrs(), capi.ints(seq_ids), dk, dv, layer,
                                    capi.ints(pos), n_kv_heads, head_dim, T)
    if lib.kind == "hip":
        lib.device_sync()
    lib.free(dk)
    lib.free(dv)


def op_paged_attention_batch(lib, pool, q, layer, pos, seq_ids, max_seq_len, n_heads, n_kv_heads, head_dim):
    T = q.shape[0]
    dq, do = lib.to_device(f32(q)), lib.empty_device(q.nbytes)
    datt = lib.empty_device(n_heads * max_seq_len * 4)
    lib.paged_attention_batch(do, dq, C.byref(pool.mgr), pool.table_ptrs(), capi.ints(seq_ids), datt, layer,
                              capi.ints(pos), max_seq_len, n_heads, n_kv_heads, head_dim, T)
    out = lib.to_host(do, q.shape)
    for p in (dq, do, datt):
        lib.free(p)
    return out


def engine_kv_rows(lib, e, n_seqs):
    """K and V of every position each sequence has been through, read out of the driver's
    block pool (the library's global g_kv_manager) through the sequences' block tables, in
    the reference's logical layout: per sequence two arrays [layers, positions, kv_heads * head_dim]."""
    mgr = capi.KVCacheManager.in_dll(lib.dll, "g_kv_manager")
    cfg = e.config
    view = Pool.__new__(Pool)
    view.lib, view.mgr, view.tables = lib, mgr, []
    view.geom = (cfg.n_layers, cfg.n_kv_heads, cfg.head_dim)
    row = cfg.n_kv_heads * cfg.head_dim
    bs = mgr.block_size
    pools = [view.dump(which).reshape(cfg.n_layers, mgr.num_blocks * bs, row) for which in ("k", "v")]
    out = []
    for i in range(n_seqs):
        n_pos = e.seq_state(i)["pos"]
        table = e.block_table(i)
        rows = np.array([table[p // bs] * bs + p % bs for p in range(n_pos)], dtype=np.int64)
        out.append(tuple(pl[:, rows, :].copy() for pl in pools))
    return out


def run_engine(lib, model_path, prompts, n_new, arrivals=None, collect_logits=False, kv_rows=False, **opts):
    """Run the N-sequence driver to completion; returns histories, allocation trace
    and (optionally) the logits of every step and the K/V rows left in the pool."""
    n = len(prompts)
    e = capi.Engine(lib, model_path, n, **opts)
    for i, p in enumerate(prompts):
        e.set_request(i, p, n_new if np.isscalar(n_new) else n_new[i], 0 if arrivals is None else arrivals[i])
    steps = []
    while e.step():
        lb = e.last_batch()
        rec = dict(lb)
        if collect_logits:
            rec["logits"] = e.logits(len(lb["output_indices"]))
        steps.append(rec)
    out = dict(histories=[e.history(i) for i in range(n)], trace=e.alloc_trace(), steps=steps,
               tables=[e.block_table(i) for i in range(n)], stats=e.stats())
    if kv_rows:
        out["kv"] = engine_kv_rows(lib, e, n)
    e.close()
    return out
