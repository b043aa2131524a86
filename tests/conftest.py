import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def oracle():
    """CPU oracle (test infrastructure only)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope='session')
def gpu_ctx():
    from simple_mip_solver_amd import _ffi
    return _ffi.default_context()


# Host-logic tests run twice: against the CPU oracle (no GPU needed, `-m "not gpu"`) and against
# the real engine through the C ABI (`-m gpu`).  The product default is always the HIP backend.
@pytest.fixture(params=['oracle', pytest.param('hip', marks=pytest.mark.gpu)])
def engine(request):
    from simple_mip_solver_amd import lp
    if request.param == 'oracle':
        from tests.support.oracle_backend import OracleBackend
        backend = OracleBackend()
    else:
        from tests.support.compare_backend import CompareBackend
        backend = CompareBackend()  # the HIP engine, every solve checked against the oracle
    lp.set_backend(backend)
    yield backend
    lp.set_backend(None)
