"""The CPU oracle against the golden vectors recorded from the REFERENCE build
(tools/make_golden.py, run where /root/reference exists).  Bit-exact: the oracle
keeps the reference's operation order, libm calls and compiler flags.  With
-m gpu the same vectors check the HIP path, to the tolerance of each operator."""
import os
import tempfile

import numpy as np
import pytest

from tools import make_golden as mg

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def test_oracle_ops_match_reference_vectors_bit_exact(oracle):
    want = _load("ops_ref.npz")
    got = mg.run_ops(oracle, mg.golden_inputs_ops())
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


@pytest.mark.parametrize("case", sorted(mg.ENGINE_CASES))
def test_oracle_forward_pass_matches_reference_vectors_bit_exact(oracle, case):
    want = _load(f"engine_{case}_ref.npz")
    with tempfile.TemporaryDirectory() as d:
        got = mg.pack_engine(mg.run_engine_case(oracle, case, d))
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


@pytest.mark.gpu
def test_hip_ops_match_reference_vectors(hip):
    want = _load("ops_ref.npz")
    got = mg.run_ops(hip, mg.golden_inputs_ops())
    exact = {"pool_k", "pool_v"} | {k for k in want if k.startswith("rope_")}   # copies and RoPE: bit-exact
    for k in want:
        if k in exact:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        else:
            # fp32 arithmetic with a different summation order / device expf
            np.testing.assert_allclose(got[k], want[k], rtol=3e-6, atol=3e-6 * float(np.abs(want[k]).max()), err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(mg.ENGINE_CASES))
def test_hip_forward_pass_matches_reference_vectors(hip, case):
    want = _load(f"engine_{case}_ref.npz")
    with tempfile.TemporaryDirectory() as d:
        got = mg.pack_engine(mg.run_engine_case(hip, case, d))
    np.testing.assert_array_equal(got["trace"], want["trace"])          # allocation order: bit-exact
    np.testing.assert_array_equal(got["sampled"], want["sampled"])      # greedy tokens: identical
    for k in want:
        if k.startswith("hist"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        if k.startswith("logits_step"):
            assert float(np.abs(got[k] - want[k]).max()) < 2e-4, k      # bar for this path: 2e-2
    assert float(want["margins"].min()) > 1e-4   # the recorded top-2 margins dwarf the logits error
