"""The dense-tile projection kernel (csrc/gemm_prefill.hip) in every form it can take -- tile heights
of 128, 96 and 64 tokens, the four- and the eight-wave form, K cut across workgroups -- against the
oracle's fp32 matmul (reference src/kernels/layers.c:44-67) on ragged token counts and on widths that
are not multiples of the tile.  The form is read from the environment once per process, hence a child
process per form."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import dense_tile_child

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOKENS = (256, 257, 300, 511, 777)

FORMS = [
    {"NVLLM_GEMM_PREFILL_TW": "4", "NVLLM_GEMM_PREFILL_SHAPE": "3"},
    {"NVLLM_GEMM_PREFILL_TW": "3", "NVLLM_GEMM_PREFILL_SHAPE": "3"},
    {"NVLLM_GEMM_PREFILL_TW": "2", "NVLLM_GEMM_PREFILL_SHAPE": "3"},
    {"NVLLM_GEMM_PREFILL_TW": "3", "NVLLM_GEMM_PREFILL_SHAPE": "3", "NVLLM_GEMM_PREFILL_RING": "2"},
    {"NVLLM_GEMM_PREFILL_SHAPE": "2"},
    {"NVLLM_GEMM_PREFILL_SHAPE": "2", "NVLLM_GEMM_PREFILL_KSPLIT": "2"},
    {"NVLLM_GEMM_PREFILL_SHAPE": "3", "NVLLM_GEMM_PREFILL_TW": "2", "NVLLM_GEMM_PREFILL_KSPLIT": "3"},
    {},
]


@pytest.fixture(scope="module")
def wanted(oracle, tmp_models):
    path, _ = tmp_models("tiny-mha48")   # dim 96, hidden 224, vocab 384: no width is a multiple of 128
    return path, dense_tile_child.run(oracle, path, TOKENS)


@pytest.mark.gpu
@pytest.mark.parametrize("form", FORMS, ids=lambda f: ",".join(f"{k[19:]}={v}" for k, v in f.items()) or "default")
def test_every_form_of_the_dense_tile_kernel(wanted, tmp_path, form):
    path, want = wanted
    dst = str(tmp_path / "got.npz")
    env = dict(os.environ, **form)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dense_tile_child.py"), path, dst,
                        ",".join(str(t) for t in TOKENS)], env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(dst)
    assert set(got.files) == set(want)
    for key, w in want.items():
        # fp16 weights are exact in both; the activation enters as hi+lo fp16 (~2^-22 relative)
        np.testing.assert_allclose(got[key], w, rtol=0, atol=3e-6 * np.abs(w).max() + 1e-7, err_msg=key)
