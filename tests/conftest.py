import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session', autouse=True)
def _hip_library_built():
    """The shared library is a build artefact (git-ignored): compile it for gfx950 when it is missing or stale.
    hipcc cross-compiles without a GPU, so this also is the CPU suite's "does it build" check."""
    from wavthruvec_pytorch_amd import build
    if build.needs_build():
        build.build()
    return build.LIB_PATH
