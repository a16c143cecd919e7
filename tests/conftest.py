import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; without a device they are skipped
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# the seven parameter sets of the reference's test/InfectionSimTest.jl:2-10
def reference_test_params():
    from epidemicmodeling_amd import make_pomdp
    return [
        make_pomdp(symptomatic_isolation_prob=1.0, asymptomatic_prob=0.0, LOD=5),
        make_pomdp(symptomatic_isolation_prob=0.0, asymptomatic_prob=1.0, LOD=5),
        make_pomdp(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.40, LOD=5),
        make_pomdp(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.40, LOD=5, test_delay=1),
        make_pomdp(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.40, LOD=5, test_delay=5),
        make_pomdp(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.65, LOD=3),
        make_pomdp(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.65, LOD=3),
    ]


def random_states(pomdp, n, rng, scale=10_000, with_tests=True):
    """Valid random states in the host layout: (S[n], I[n,H], R[n], T[n,H,D+1])."""
    H, D = pomdp.horizon, pomdp.test_delay
    I = rng.integers(0, scale, size=(n, H)).astype(np.int64)
    I[rng.random((n, H)) < 0.1] = 0                      # empty bins
    T = np.zeros((n, H, D + 1), dtype=np.int64)
    if D > 0 and with_tests:
        # pending tests: a random split of at most I[c] over rows 0..D-1 of columns 1..H-1
        frac = rng.random((n, H, D)) * 0.3 / D
        T[:, 1:, :D] = np.floor(frac[:, 1:, :] * I[:, 1:, None]).astype(np.int64)
    R = rng.integers(0, scale * 5, size=n).astype(np.int64)
    S = pomdp.N - I.sum(axis=1) - R
    assert (S >= 0).all()
    return S, I, R, T


@pytest.fixture
def testing_lib():
    """The -DEPI_TESTING build of the library as the one every context of this test uses (contexts are cached per
    library, so the cache is emptied on the way in and out)."""
    from epidemicmodeling_amd import _lib, interface
    saved = _lib._lib
    interface._contexts.clear()
    _lib._lib = _lib.load_testing()
    try:
        yield _lib._lib
    finally:
        _lib._lib.epi_debug_step_impl(1)
        _lib._lib.epi_debug_screen(1)
        _lib._lib.epi_debug_node_exchange(0)
        interface._contexts.clear()
        _lib._lib = saved
