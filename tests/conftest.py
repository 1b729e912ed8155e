import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# One ROCm stack per process: torch wheels bundle their own libamdhip64 / librccl, and whichever of torch and
# libgpk.so is loaded FIRST decides which HIP runtime the process runs on.  bench.py and the drivers load torch first; so does every test
# session, whatever subset of files it runs -- a session that opened libgpk.so (system ROCm), bound the system librccl in-process
# (tests/test_gpu_mg.py::test_rccl_binding_comes_up_with_one_rank) and only THEN imported torch ended with two runtimes and a double free at
# interpreter exit (round 6: `pytest tests/test_gpu_mg.py tests/test_gpu_sharded_ops.py`; the full suite never did, an earlier file imports torch).
try:
    import torch  # noqa: F401,E402
except Exception:                                                 # noqa: BLE001 -- CPU-only tiers without torch still collect
    pass

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REFERENCE = '/root/reference'          # exists only in the authoring container, never on the GPU box


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (authoring container only)')


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(REFERENCE)
    skip_ref = pytest.mark.skip(reason='/root/reference not present (GPU box)')
    for item in items:
        if 'reference' in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='module')
def dev_ctx():
    """A handle in libgpk_dev.so -- the development build that also holds the look-ahead factorisation measured in round 5 (gpk_tune key 54),
    the probes and the micro-benchmarks (include/gpk_dev.h).  The product library libgpk.so rejects those keys."""
    import gpk
    c = gpk.Context(0, dev=True)
    yield c
    c.close()
