import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from mp3common import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def product():
    """The real library.  Loading it does not need a GPU; computing with it does."""
    from mp3common import Mp3mi, PRODUCT_SO
    if not os.path.exists(PRODUCT_SO):
        pytest.fail("libmp3mi.so is not built: run python -c 'import __graft_entry__ as g; g.build()'")
    return Mp3mi(emu=False)


@pytest.fixture(scope="session")
def emu():
    from mp3common import Mp3mi
    return Mp3mi(emu=True)
