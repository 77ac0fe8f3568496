import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library; built on demand where hipcc exists (it cross-compiles without a GPU)."""
    from news_recommendation_model_amd import build, native
    if not os.path.exists(native.LIB_PATH):
        build.build()
    return native.load()
