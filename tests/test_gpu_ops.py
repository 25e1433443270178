"""GPU parity of every operator of include/ops.h against the CPU oracle, on the
same seeded inputs, through the C ABI.  Integer / copy work must match bit for
bit; floating-point work within the tolerance written next to each assert."""
import ctypes as C

import numpy as np
import pytest

from nvllm_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


@pytest.mark.parametrize("T,D", [(1, 64), (7, 288), (64, 2048), (3, 100), (130, 512)])
def test_rms_norm(hip, oracle, T, D):
    r = rng(T * 1000 + D)
    x = r.standard_normal((T, D)).astype(np.float32) * r.choice([1e-3, 1.0, 50.0], size=(T, 1)).astype(np.float32)
    w = (1 + 0.1 * r.standard_normal(D)).astype(np.float32)
    want = util.op_rms_norm(oracle, x, w)
    got = util.op_rms_norm(hip, x, w)
    # only the order of the sum of squares differs: a few ulp
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)
    got_inplace = util.op_rms_norm(hip, x, w, inplace=True)  # reference model.c:396 aliases in and out
    np.testing.assert_array_equal(got_inplace, got)


@pytest.mark.parametrize("T,K,N", [(1, 64, 48), (7, 288, 288), (64, 256, 100), (33, 130, 17)])
def test_matmul_raw_fp32_weights(hip, oracle, T, K, N):
    r = rng(T + K + N)
    x = r.standard_normal((T, K)).astype(np.float32)
    w = (r.standard_normal((N, K)) * 0.05).astype(np.float32)
    want = util.op_matmul(oracle, x, w)
    got = util.op_matmul(hip, x, w)
    # fp32 products, different summation order: error ~ sqrt(K) * eps * |x||w|
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * np.sqrt(K))


@pytest.mark.parametrize("hd,H,KVH", [(64, 32, 4), (48, 6, 6), (128, 4, 4), (16, 4, 2)])
def test_rope_batch_bit_exact(hip, oracle, hd, H, KVH):
    r = rng(hd + H)
    pos = np.array([0, 1, 15, 16, 17, 511, 2047, 4095, 3, 1000], dtype=np.int32)
    T = len(pos)
    q = r.standard_normal((T, H * hd)).astype(np.float32)
    k = r.standard_normal((T, KVH * hd)).astype(np.float32)
    wq, wk = util.op_rope_batch(oracle, q, k, pos, 10000.0, hd, H, KVH)
    gq, gk = util.op_rope_batch(hip, q, k, pos, 10000.0, hd, H, KVH)
    # angles come from the same libm calls on the host; the rotation uses the
    # reference's two-roundings form, so the result is bit-identical
    np.testing.assert_array_equal(gq, wq)
    np.testing.assert_array_equal(gk, wk)
    wq, wk = util.op_rope(oracle, q, k, 37, 10000.0, hd, H, KVH)
    gq, gk = util.op_rope(hip, q, k, 37, 10000.0, hd, H, KVH)
    np.testing.assert_array_equal(gq, wq)
    np.testing.assert_array_equal(gk, wk)


def test_swiglu_accum_softmax(hip, oracle):
    r = rng(5)
    g = (r.standard_normal((9, 160)) * r.choice([0.1, 1, 30], size=(9, 1))).astype(np.float32)
    g[0, :4] = [-100.0, 100.0, 0.0, -0.0]
    u = r.standard_normal((9, 160)).astype(np.float32)
    want = util.op_swiglu(oracle, g, u)
    got = util.op_swiglu(hip, g, u)
    # device expf vs glibc expf: <= 2 ulp of the result
    np.testing.assert_allclose(got, want, rtol=5e-7, atol=1e-30)
    a, b = r.standard_normal(1000).astype(np.float32), r.standard_normal(1000).astype(np.float32)
    np.testing.assert_array_equal(util.op_accum(hip, a, b), util.op_accum(oracle, a, b))
    for n in (1, 17, 1000, 32000):
        x = (r.standard_normal(n) * 4).astype(np.float32)
        # the reference adds the n exponentials one after another in fp32, so ITS
        # denominator carries an error that grows with n; the GPU sums as a tree
        np.testing.assert_allclose(util.op_softmax(hip, x), util.op_softmax(oracle, x), rtol=2e-6 + 4e-9 * n,
                                   atol=1e-12)


def _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, max_blocks, scramble_seed=None):
    ph = util.Pool(hip, bs, nb, L, KVH, hd)
    po = util.Pool(oracle, bs, nb, L, KVH, hd)
    for _ in range(n_seqs):
        ph.new_table(max_blocks)
        po.new_table(max_blocks)
    return ph, po


def test_allocator_and_offsets_bit_exact(hip, oracle):
    """Host bookkeeping of reference memory.c: LIFO stack, first pops are 0,1,2.."""
    ph, po = _pools(hip, oracle, 16, 9, 2, 2, 16, 0, 0)
    seq_h = [hip.alloc_block(C.byref(ph.mgr)) for _ in range(11)]
    seq_o = [oracle.alloc_block(C.byref(po.mgr)) for _ in range(11)]
    assert seq_h == seq_o == list(range(9)) + [-1, -1]
    for b in (4, 2, 7):
        hip.free_block(C.byref(ph.mgr), b)
        oracle.free_block(C.byref(po.mgr), b)
    assert [hip.alloc_block(C.byref(ph.mgr)) for _ in range(4)] == [7, 2, 4, -1]
    r = rng(3)
    for _ in range(50):
        args = (int(r.integers(0, 2)), int(r.integers(0, 9)), int(r.integers(0, 16)), 2, 16)
        assert hip.get_physical_offset(C.byref(ph.mgr), *args) == oracle.get_physical_offset(C.byref(po.mgr), *args)
    ph.close()
    po.close()


@pytest.mark.parametrize("hd,H,KVH", [(64, 32, 4), (48, 6, 6), (128, 8, 8), (16, 4, 2), (64, 8, 1)])
def test_kv_write_and_paged_attention(hip, oracle, hd, H, KVH):
    """Ragged batch: one 21-token prefill chunk that straddles block boundaries,
    decode rows at context lengths 1, 16, 17, 100 and 1000, scattered block tables."""
    r = rng(hd * 7 + H)
    L, bs, S = 2, 16, 4096
    ctx = [40, 1, 15, 16, 17, 100, 1000, 4096]  # keys visible to the LAST token of each sequence
    n_seqs, maxb = len(ctx), S // bs
    nb = sum((c + bs - 1) // bs for c in ctx) + 5
    ph, po = _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, maxb)
    perm = r.permutation(nb)
    cur = 0
    for s, c in enumerate(ctx):                # non-monotone physical blocks, same on both sides
        need = (c + bs - 1) // bs
        blocks = perm[cur:cur + need].tolist()
        cur += need
        ph.set_table(s, blocks)
        po.set_table(s, blocks)
    # history: random K/V everywhere in the pool (identical), then this step's rows on top
    kfill = r.standard_normal(ph.elems()).astype(np.float32)
    vfill = r.standard_normal(ph.elems()).astype(np.float32)
    ph.fill(kfill, vfill)
    po.fill(kfill, vfill)
    pos, sid = [], []
    for p in range(ctx[0] - 21, ctx[0]):       # prefill chunk of sequence 0
        pos.append(p)
        sid.append(0)
    for s in range(1, n_seqs):
        pos.append(ctx[s] - 1)
        sid.append(s)
    T = len(pos)
    k = r.standard_normal((T, KVH * hd)).astype(np.float32)
    v = r.standard_normal((T, KVH * hd)).astype(np.float32)
    q = r.standard_normal((T, H * hd)).astype(np.float32)
    for layer in (1,):
        util.op_kv_write_batch(oracle, po, k, v, layer, pos, sid, KVH, hd)
        util.op_kv_write_batch(hip, ph, k, v, layer, pos, sid, KVH, hd)
        np.testing.assert_array_equal(ph.dump("k"), po.dump("k"))   # copies: bit-exact
        np.testing.assert_array_equal(ph.dump("v"), po.dump("v"))
        want = util.op_paged_attention_batch(oracle, po, q, layer, pos, sid, S, H, KVH, hd)
        got = util.op_paged_attention_batch(hip, ph, q, layer, pos, sid, S, H, KVH, hd)
        # fp32 throughout; online softmax and tiled sums reorder the additions
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-6 * np.abs(want).max() + 2e-6)
    ph.close()
    po.close()


@pytest.mark.parametrize("hd,H,KVH", [(64, 32, 4), (48, 6, 6), (128, 8, 8), (16, 4, 2), (64, 8, 1), (96, 2, 2)])
def test_fp16_pool_kv_write_and_paged_attention(hip, oracle, hd, H, KVH):
    """The fp16 pool (pages in matrix-core operand order).  Written rows must be the
    fp16 roundings of what the reference stores, at the reference's logical
    coordinates; attention over the pool must match the oracle run on those rounded
    values to fp32 accuracy."""
    r = rng(hd * 11 + H)
    L, bs, S = 2, 16, 4096
    ctx = [40, 1, 15, 16, 17, 100, 1000, 33, 4096]
    n_seqs, maxb = len(ctx), S // bs
    nb = sum((c + bs - 1) // bs for c in ctx) + 5
    hip.nvllm_set_kv_elem_bytes(2)
    try:
        ph, po = _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, maxb)
    finally:
        hip.nvllm_set_kv_elem_bytes(4)
    hip.nvllm_set_kv_elem_bytes(2)   # Pool helpers ask the library for the element size
    try:
        perm = r.permutation(nb)
        cur = 0
        for s, c in enumerate(ctx):
            need = (c + bs - 1) // bs
            blocks = perm[cur:cur + need].tolist()
            cur += need
            ph.set_table(s, blocks)
            po.set_table(s, blocks)
        h16 = lambda a: a.astype(np.float16).astype(np.float32)
        kfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        vfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        ph.fill(kfill, vfill)
        po.fill(kfill, vfill)
        np.testing.assert_array_equal(ph.dump("k"), kfill)          # the helpers' view of the layout round-trips
        pos, sid = [], []
        for p in range(ctx[0] - 21, ctx[0]):
            pos.append(p)
            sid.append(0)
        for s in range(1, n_seqs):
            pos.append(ctx[s] - 1)
            sid.append(s)
        T = len(pos)
        k = r.standard_normal((T, KVH * hd)).astype(np.float32)
        v = r.standard_normal((T, KVH * hd)).astype(np.float32)
        k[0, 0], v[0, 1] = 1e6, -1e6                                # saturate instead of overflowing to inf
        q = r.standard_normal((T, H * hd)).astype(np.float32)
        sat = lambda a: h16(np.clip(a, -65504, 65504))
        layer = 1
        util.op_kv_write_batch(oracle, po, sat(k), sat(v), layer, pos, sid, KVH, hd)
        util.op_kv_write_batch(hip, ph, k, v, layer, pos, sid, KVH, hd)
        np.testing.assert_array_equal(ph.dump("k"), po.dump("k"))
        np.testing.assert_array_equal(ph.dump("v"), po.dump("v"))
        want = util.op_paged_attention_batch(oracle, po, q, layer, pos, sid, S, H, KVH, hd)
        got = util.op_paged_attention_batch(hip, ph, q, layer, pos, sid, S, H, KVH, hd)
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-6 * np.abs(want).max() + 2e-6)
    finally:
        hip.nvllm_set_kv_elem_bytes(4)
        ph.close()
        po.close()


@pytest.mark.parametrize("hd,H,KVH,n_seqs", [(128, 16, 16, 160), (64, 32, 32, 96), (48, 12, 6, 400)])
def test_fp16_pool_attention_many_short_sequences(hip, oracle, hd, H, KVH, n_seqs):
    """Thousands of (sequence, head group) items of at most 16 pages each -- the shape of a decode batch
    of a multi-head model at short context -- which the library serves with one wave per item instead
    of four waves per item.  Contexts cover 1 key, page boundaries and the 16-page limit."""
    r = rng(hd * 13 + H + n_seqs)
    L, bs, S = 2, 16, 256
    ctx = [int(c) for c in r.integers(1, 257, size=n_seqs)]
    ctx[:6] = [1, 15, 16, 17, 255, 256]
    maxb = S // bs
    nb = sum((c + bs - 1) // bs for c in ctx) + 3
    hip.nvllm_set_kv_elem_bytes(2)
    try:
        ph, po = _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, maxb)
        perm = r.permutation(nb)
        cur = 0
        for s, c in enumerate(ctx):
            need = (c + bs - 1) // bs
            blocks = perm[cur:cur + need].tolist()
            cur += need
            ph.set_table(s, blocks)
            po.set_table(s, blocks)
        h16 = lambda a: a.astype(np.float16).astype(np.float32)
        kfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        vfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        ph.fill(kfill, vfill)
        po.fill(kfill, vfill)
        pos = [c - 1 for c in ctx]
        sid = list(range(n_seqs))
        q = r.standard_normal((n_seqs, H * hd)).astype(np.float32)
        want = util.op_paged_attention_batch(oracle, po, q, 1, pos, sid, S, H, KVH, hd)
        got = util.op_paged_attention_batch(hip, ph, q, 1, pos, sid, S, H, KVH, hd)
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-6 * np.abs(want).max() + 2e-6)
    finally:
        hip.nvllm_set_kv_elem_bytes(4)
        ph.close()
        po.close()


@pytest.mark.parametrize("hd,H,KVH", [(64, 32, 4), (48, 6, 6), (128, 8, 2)])
def test_fp16_pool_attention_over_prefill_chunks(hip, oracle, hd, H, KVH):
    """Hundreds of tokens per call (chunks of several sequences at different depths plus decode
    rows): thousands of (token run, head group) items, the regime in which a workgroup takes
    four consecutive items, one per wave."""
    r = rng(hd * 7 + H)
    L, bs, S = 1, 16, 1024
    # (first position, tokens) per sequence: chunks that start mid-sequence, one whole prompt, decode rows
    spans = [(300, 400), (0, 333), (700, 250), (512, 1), (17, 1), (1000, 24)]
    n_seqs, maxb = len(spans), S // bs
    nb = sum((p0 + n + bs - 1) // bs for p0, n in spans) + 3
    hip.nvllm_set_kv_elem_bytes(2)
    try:
        ph, po = _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, maxb)
        perm = r.permutation(nb)
        cur = 0
        for s_, (p0, n) in enumerate(spans):
            need = (p0 + n + bs - 1) // bs
            blocks = perm[cur:cur + need].tolist()
            cur += need
            ph.set_table(s_, blocks)
            po.set_table(s_, blocks)
        h16 = lambda a: a.astype(np.float16).astype(np.float32)
        kfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        vfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        ph.fill(kfill, vfill)
        po.fill(kfill, vfill)
        pos = [p for p0, n in spans for p in range(p0, p0 + n)]
        sid = [s_ for s_, (p0, n) in enumerate(spans) for _ in range(n)]
        T = len(pos)
        assert T > 1000
        q = r.standard_normal((T, H * hd)).astype(np.float32)
        want = util.op_paged_attention_batch(oracle, po, q, 0, pos, sid, S, H, KVH, hd)
        got = util.op_paged_attention_batch(hip, ph, q, 0, pos, sid, S, H, KVH, hd)
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-6 * np.abs(want).max() + 2e-6)
    finally:
        hip.nvllm_set_kv_elem_bytes(4)
        ph.close()
        po.close()


@pytest.mark.parametrize("kv_bytes", [2, 4])
@pytest.mark.parametrize("lo,hi", [(1040, 1100), (3900, 3960)])
def test_attention_at_the_benchmark_shape(hip, oracle, kv_bytes, lo, hi):
    """The decode batch the benchmark times, as one operator call: 64 sequences, one new token each, at
    contexts of 1,040 to 1,100 keys (65 to 69 pages per item) and of about 3,900; 32 query heads over
    4 kv heads of 64 -- 256 items of four waves, one per CU.  fp16 pages and the reference's fp32 rows."""
    hd, H, KVH, n_seqs = 64, 32, 4, 64
    r = rng(lo + kv_bytes)
    L, bs, S = 1, 16, 4096
    ctx = [int(c) for c in r.integers(lo, hi + 1, size=n_seqs)]
    ctx[:4] = [lo, hi, lo + 15, lo + 16]
    maxb = S // bs
    nb = sum((c + bs - 1) // bs for c in ctx) + 7
    hip.nvllm_set_kv_elem_bytes(kv_bytes)
    try:
        ph, po = _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, maxb)
        perm = r.permutation(nb)
        cur = 0
        for s, c in enumerate(ctx):
            need = (c + bs - 1) // bs
            blocks = perm[cur:cur + need].tolist()
            cur += need
            ph.set_table(s, blocks)
            po.set_table(s, blocks)
        h16 = lambda a: a.astype(np.float16).astype(np.float32)
        kfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        vfill = h16(r.standard_normal(ph.elems()).astype(np.float32))
        ph.fill(kfill, vfill)
        po.fill(kfill, vfill)
        pos = [c - 1 for c in ctx]
        sid = list(range(n_seqs))
        q = r.standard_normal((n_seqs, H * hd)).astype(np.float32)
        want = util.op_paged_attention_batch(oracle, po, q, 0, pos, sid, S, H, KVH, hd)
        got = util.op_paged_attention_batch(hip, ph, q, 0, pos, sid, S, H, KVH, hd)
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-6 * np.abs(want).max() + 2e-6)
    finally:
        hip.nvllm_set_kv_elem_bytes(4)
        ph.close()
        po.close()


@pytest.mark.parametrize("n_seqs,H,KVH,hd,ctx_lo,ctx_hi", [
    (160, 8, 4, 64, 40, 300),      # 640 items: three key slices per item -> the four waves of a workgroup, merged in LDS
    (96, 6, 6, 48, 17, 200),       # 576 items, head width 48 (dims past 48 idle in the merge), contexts from two pages
    (40, 32, 8, 128, 100, 700),    # 320 items: six slices -> eight waves per workgroup, two 64-dim halves per lane
    (70, 4, 2, 32, 1, 64),         # 140 items of one to four pages: slices mostly empty
    (3, 8, 2, 64, 2000, 2100),     # 6 items: more than eight slices each -> partials in memory and the merge launch
])
def test_fp32_row_attention_in_every_slice_form(hip, oracle, n_seqs, H, KVH, hd, ctx_lo, ctx_hi):
    """The fp32-row attention cuts an item's keys into slices until the chip is full; 2-8 slices of an item share a
    workgroup (4 or 8 waves) and meet in LDS, more than 8 leave partials in memory for a second launch.  One decode
    token per sequence over scattered tables, against the oracle."""
    r = rng(n_seqs * 1000 + hd)
    L, bs, S = 1, 16, 2112
    ctx = [int(c) for c in r.integers(ctx_lo, ctx_hi + 1, size=n_seqs)]
    ctx[0], ctx[-1] = ctx_lo, ctx_hi
    maxb = S // bs
    nb = sum((c + bs - 1) // bs for c in ctx) + 5
    ph, po = _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, maxb)
    try:
        perm = r.permutation(nb)
        cur = 0
        for s_, c in enumerate(ctx):
            need = (c + bs - 1) // bs
            blocks = perm[cur:cur + need].tolist()
            cur += need
            ph.set_table(s_, blocks)
            po.set_table(s_, blocks)
        kfill = r.standard_normal(ph.elems()).astype(np.float32)
        vfill = r.standard_normal(ph.elems()).astype(np.float32)
        ph.fill(kfill, vfill)
        po.fill(kfill, vfill)
        pos = [c - 1 for c in ctx]
        sid = list(range(n_seqs))
        q = r.standard_normal((n_seqs, H * hd)).astype(np.float32)
        want = util.op_paged_attention_batch(oracle, po, q, 0, pos, sid, S, H, KVH, hd)
        got = util.op_paged_attention_batch(hip, ph, q, 0, pos, sid, S, H, KVH, hd)
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-6 * np.abs(want).max() + 2e-6)
    finally:
        ph.close()
        po.close()


@pytest.mark.parametrize("bs", [32, 64])
def test_paged_attention_with_larger_blocks_of_fp32_rows(hip, oracle, bs):
    """The reference's pool takes any block size (memory.c:74-82); the library's fp32 rows take 16, 32, 64 ...:
    a block then holds several of the kernel's 16-position pages.  Scattered tables, contexts that end inside
    a block, at a page boundary inside a block and at a block boundary."""
    hd, H, KVH = 64, 8, 2
    r = rng(bs)
    L, S = 2, 1024
    ctx = [1, 15, 16, 17, bs - 1, bs, bs + 1, 2 * bs + 5, 300, 1000]
    n_seqs, maxb = len(ctx), S // bs
    nb = sum((c + bs - 1) // bs for c in ctx) + 3
    ph, po = _pools(hip, oracle, bs, nb, L, KVH, hd, n_seqs, maxb)
    perm = r.permutation(nb)
    cur = 0
    for s, c in enumerate(ctx):
        need = (c + bs - 1) // bs
        blocks = perm[cur:cur + need].tolist()
        cur += need
        ph.set_table(s, blocks)
        po.set_table(s, blocks)
    kfill = r.standard_normal(ph.elems()).astype(np.float32)
    vfill = r.standard_normal(ph.elems()).astype(np.float32)
    ph.fill(kfill, vfill)
    po.fill(kfill, vfill)
    pos = [c - 1 for c in ctx]
    sid = list(range(n_seqs))
    T = len(pos)
    k = r.standard_normal((T, KVH * hd)).astype(np.float32)
    v = r.standard_normal((T, KVH * hd)).astype(np.float32)
    q = r.standard_normal((T, H * hd)).astype(np.float32)
    util.op_kv_write_batch(oracle, po, k, v, 1, pos, sid, KVH, hd)
    util.op_kv_write_batch(hip, ph, k, v, 1, pos, sid, KVH, hd)
    np.testing.assert_array_equal(ph.dump("k"), po.dump("k"))
    np.testing.assert_array_equal(ph.dump("v"), po.dump("v"))
    want = util.op_paged_attention_batch(oracle, po, q, 1, pos, sid, S, H, KVH, hd)
    got = util.op_paged_attention_batch(hip, ph, q, 1, pos, sid, S, H, KVH, hd)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-6 * np.abs(want).max() + 2e-6)
    ph.close()
    po.close()


def test_kv_write_table_too_small_stops(hip, oracle, capfd):
    """Reference attention.c:159-162: the first token whose logical block is not in
    the table prints and ends the call; later tokens are not written."""
    L, bs, KVH, hd = 1, 16, 2, 16
    ph, po = _pools(hip, oracle, bs, 8, L, KVH, hd, 2, 8)
    for p_ in (ph, po):
        p_.set_table(0, [3])
        p_.set_table(1, [5])
    z = np.zeros(ph.elems(), np.float32)
    ph.fill(z, z)
    po.fill(z, z)
    r = rng(11)
    k = r.standard_normal((3, KVH * hd)).astype(np.float32)
    v = r.standard_normal((3, KVH * hd)).astype(np.float32)
    pos, sid = [2, 16, 4], [0, 0, 1]          # second token needs logical block 1 of a 1-block table
    util.op_kv_write_batch(oracle, po, k, v, 0, pos, sid, KVH, hd)
    util.op_kv_write_batch(hip, ph, k, v, 0, pos, sid, KVH, hd)
    np.testing.assert_array_equal(ph.dump("k"), po.dump("k"))
    np.testing.assert_array_equal(ph.dump("v"), po.dump("v"))
    assert np.count_nonzero(ph.dump("k")) == KVH * hd   # only the first token landed
    ph.close()
    po.close()


def test_contiguous_cache_ops(hip, oracle):
    """update_kv_cache / multi_head_attention (reference attention.c:11-103)."""
    r = rng(21)
    L, S, H, KVH, hd, T, pos0 = 2, 64, 4, 2, 16, 5, 9
    n = L * S * KVH * hd
    kc, vc = r.standard_normal(n).astype(np.float32), r.standard_normal(n).astype(np.float32)
    k = r.standard_normal((T, KVH * hd)).astype(np.float32)
    v = r.standard_normal((T, KVH * hd)).astype(np.float32)
    q = r.standard_normal((T, H * hd)).astype(np.float32)
    outs = {}
    for lib in (oracle, hip):
        dkc, dvc = lib.to_device(kc), lib.to_device(vc)
        dk, dv, dq = lib.to_device(k), lib.to_device(v), lib.to_device(q)
        do, datt = lib.empty_device(q.nbytes), lib.empty_device(H * S * 4)
        lib.update_kv_cache(dkc, dvc, dk, dv, 1, pos0, S, KVH, hd, T)
        lib.multi_head_attention(do, dq, dkc, dvc, datt, 1, pos0, S, H, KVH, hd, T)
        outs[lib.kind] = (lib.to_host(dkc, (n,)), lib.to_host(dvc, (n,)), lib.to_host(do, q.shape))
    np.testing.assert_array_equal(outs["hip"][0], outs["oracle"][0])
    np.testing.assert_array_equal(outs["hip"][1], outs["oracle"][1])
    np.testing.assert_allclose(outs["hip"][2], outs["oracle"][2], rtol=0, atol=1e-5)


def test_attention_over_a_table_that_is_too_short_ends_the_process(tmp_path):
    """The reference reads block ids past the end of the table (attention.c:285 has no bounds
    check).  On the GPU that would be a wild read of the pool: the library neither reads the
    caller's table past num_blocks nor launches; it reports and aborts."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "short_table.py"
    script.write_text(
        "import sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import numpy as np\n"
        "from nvllm_amd import capi\n"
        "from tests import util\n"
        "hip = capi.load_hip(); hip.init_ops()\n"
        "p = util.Pool(hip, 16, 8, 1, 2, 16); p.new_table(1); p.set_table(0, [3])\n"
        "q = np.zeros((1, 4 * 16), np.float32)\n"
        "util.op_paged_attention_batch(hip, p, q, 0, [20], [0], 64, 4, 2, 16)\n"   # position 20 needs logical block 1
        "print('survived')\n")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "survived" not in r.stdout
    assert "beyond its sequence's block table" in r.stderr
