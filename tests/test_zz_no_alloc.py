"""nvllm_matmul_noalloc (include/nvllm.h): matmul() for callers that cannot allocate.
Checked against matmul() itself through the C ABI -- same kernels, same K split, so the
results must be bit-identical -- on two calls in a row, the second with a smaller input
(the static scratch array keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from nvllm_amd import capi

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    """hipMemGetInfo through the HIP runtime the library itself is linked against."""
    rtl = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert rtl.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _both(hip, handle, x, K, N):
    T = x.shape[0]
    dx = hip.to_device(x)
    d_ref, d_new = hip.empty_device(T * N * 4), hip.to_device(np.full((T, N), np.nan, np.float32))
    hip.matmul(d_ref, dx, handle, T, K, N)
    hip.device_sync()
    before = _free_device_bytes()
    rc = hip.nvllm_matmul_noalloc(d_new, dx, handle, T, K, N)
    assert rc == 0
    hip.device_sync()
    assert _free_device_bytes() == before      # nothing was allocated (or freed) on the GPU
    want, got = hip.to_host(d_ref, (T, N)), hip.to_host(d_new, (T, N))
    for p in (dx, d_ref, d_new):
        hip.free(p)
    return want, got


def test_matmul_noalloc_matches_matmul_twice_in_a_row(hip, tmp_models):
    # hidden_dim 4096: w_down runs with a K split (partial tiles through the slab + arrival counters)
    path, shape = tmp_models("tiny", seed=71, dim=256, hidden_dim=4096, n_layers=1, n_heads=4, n_kv_heads=2,
                             vocab_size=777, max_seq_len=128)
    w, cfg = capi.Weights(), capi.Config()
    hip.load_model(C.byref(w), C.byref(cfg), path.encode())
    try:
        D, F, V = cfg.dim, cfg.hidden_dim, cfg.vocab_size
        lw = w.layers[0]
        r = np.random.Generator(np.random.PCG64(72))
        # first call: the widest batch; second call: fewer tokens AND a shorter K, right after it
        for handle, K, N, tokens in ((lw.w_down, F, D, (64, 5)), (lw.w_gate, D, F, (33, 1)), (w.lm_head, D, V, (64, 17))):
            for T in tokens:
                x = r.standard_normal((T, K)).astype(np.float32)
                want, got = _both(hip, handle, x, K, N)
                assert np.isfinite(got).all()
                np.testing.assert_array_equal(got, want)
        x_big = r.standard_normal((64, F)).astype(np.float32)
        x_small = r.standard_normal((3, D)).astype(np.float32)
        for handle, x, K, N in ((lw.w_down, x_big, F, D), (lw.wq, x_small, D, D)):
            want, got = _both(hip, handle, x, K, N)
            np.testing.assert_array_equal(got, want)
        # a second call on another host thread (HIP's current device is per thread)
        import threading
        res = {}
        th = threading.Thread(target=lambda: res.update(r=_both(hip, lw.w_down, x_big, F, D)))
        th.start()
        th.join()
        np.testing.assert_array_equal(res["r"][1], res["r"][0])
        # limits and bad handles are refused with an error code, nothing is enqueued
        dx, do = hip.to_device(np.zeros((65, D), np.float32)), hip.empty_device(65 * D * 4)
        assert hip.nvllm_matmul_noalloc(do, dx, lw.wq, 65, D, D) == -3
        assert hip.nvllm_matmul_noalloc(do, dx, lw.wq, 4, D, D + 16) == -2
        assert hip.nvllm_matmul_noalloc(do, dx, dx, 4, D, D) == -2
        hip.free(dx)
        hip.free(do)
    finally:
        hip.nvllm_free_model(C.byref(w))


def test_matmul_noalloc_keeps_the_k_split_of_a_wide_projection(hip, tmp_models):
    """170 tiles x 2 K slices x 16 KiB of partial tile = 5.3 MiB: the widest partial-tile slab a
    self-chosen split makes on 256 CUs still fits the static array, so the bits stay matmul()'s."""
    path, shape = tmp_models("tiny", seed=73, dim=4096, hidden_dim=10880, n_layers=1, n_heads=32, n_kv_heads=4,
                             vocab_size=64, max_seq_len=128)
    w, cfg = capi.Weights(), capi.Config()
    hip.load_model(C.byref(w), C.byref(cfg), path.encode())
    try:
        r = np.random.Generator(np.random.PCG64(74))
        for T in (64, 7):
            x = r.standard_normal((T, cfg.dim)).astype(np.float32)
            want, got = _both(hip, w.layers[0].w_gate, x, cfg.dim, cfg.hidden_dim)
            np.testing.assert_array_equal(got, want)
    finally:
        hip.nvllm_free_model(C.byref(w))


def test_matmul_noalloc_is_capturable_into_a_hip_graph(hip, tmp_models):
    """One of the callers the entry point exists for: under stream capture an allocation is an error.  The
    call is captured on the library's stream, instantiated and replayed; the replay's result is matmul()'s."""
    path, shape = tmp_models("tiny", seed=71, dim=256, hidden_dim=4096, n_layers=1, n_heads=4, n_kv_heads=2,
                             vocab_size=777, max_seq_len=128)
    w, cfg = capi.Weights(), capi.Config()
    hip.load_model(C.byref(w), C.byref(cfg), path.encode())
    rtl = C.CDLL("libamdhip64.so")
    try:
        T, K, N = 64, cfg.hidden_dim, cfg.dim
        handle = w.layers[0].w_down
        x = np.random.Generator(np.random.PCG64(75)).standard_normal((T, K)).astype(np.float32)
        dx = hip.to_device(x)
        d_ref, d_new = hip.empty_device(T * N * 4), hip.empty_device(T * N * 4)
        hip.matmul(d_ref, dx, handle, T, K, N)
        assert hip.nvllm_matmul_noalloc(d_new, dx, handle, T, K, N) == 0      # first use outside the capture
        hip.device_sync()
        blank = np.full((T, N), np.nan, np.float32)
        hip.device_memcpy(d_new, blank.ctypes.data, blank.nbytes, capi.HOST_TO_DEVICE)
        stream = C.c_void_p(hip.nvllm_stream())
        graph, executable = C.c_void_p(), C.c_void_p()
        assert rtl.hipStreamBeginCapture(stream, 0) == 0
        rc = hip.nvllm_matmul_noalloc(d_new, dx, handle, T, K, N)
        assert rtl.hipStreamEndCapture(stream, C.byref(graph)) == 0 and rc == 0 and graph.value
        hip.device_sync()
        assert np.isnan(hip.to_host(d_new, (T, N))).all()                     # captured, not run
        assert rtl.hipGraphInstantiate(C.byref(executable), graph, None, None, 0) == 0
        for _ in range(2):
            hip.device_memcpy(d_new, blank.ctypes.data, blank.nbytes, capi.HOST_TO_DEVICE)
            assert rtl.hipGraphLaunch(executable, stream) == 0
            hip.device_sync()
            np.testing.assert_array_equal(hip.to_host(d_new, (T, N)), hip.to_host(d_ref, (T, N)))
        rtl.hipGraphExecDestroy(executable)
        rtl.hipGraphDestroy(graph)
        for p in (dx, d_ref, d_new):
            hip.free(p)
    finally:
        hip.nvllm_free_model(C.byref(w))
