import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def encoder():
    from _common import product
    za = product()
    enc = za.Encoder(0)          # raises loudly when the HIP library or the GPU is missing
    yield enc
    enc.close()
