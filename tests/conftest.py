import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import checkers
    return checkers.load_oracle()


@pytest.fixture(scope="session")
def ref():
    from tests import checkers
    if not os.path.exists(checkers.REF_LIB):
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return checkers.load_ref()


@pytest.fixture(scope="session")
def hip():
    """The product library.  No fall-back: a missing build is an error."""
    from nvllm_amd import capi
    lib = capi.load_hip()
    lib.init_ops()
    return lib


@pytest.fixture(scope="session")
def tmp_models(tmp_path_factory):
    """Synthetic model files, generated once per session."""
    from nvllm_amd import synth
    d = tmp_path_factory.mktemp("models")
    cache = {}

    def get(name, seed=1, dtype="f16", shared=False, rope_tail=False, exact16=True, **over):
        key = (name, seed, dtype, shared, rope_tail, exact16, tuple(sorted(over.items())))
        if key not in cache:
            shape = synth.ModelShape.named(name, **over)
            path = str(d / f"{name}_{seed}_{dtype}_{int(shared)}_{len(cache)}.bin")
            synth.write_model(path, shape, seed=seed, dtype=dtype, shared=shared, rope_tail=rope_tail, exact16=exact16)
            cache[key] = (path, shape)
        return cache[key]

    return get
