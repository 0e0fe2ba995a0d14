import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def tune(monkeypatch):
    """A/B environment variables of the library for one test: the library reads them once (mobi_amd/csrc/tuning.h),
    so every change is followed by `mobi_tuning_reload()`; restored (and re-read) at teardown."""
    from mobi_amd import _lib

    class Tune:
        def setenv(self, key, value):
            monkeypatch.setenv(key, str(value))
            _lib.load().mobi_tuning_reload()

        def delenv(self, key):
            monkeypatch.delenv(key, raising=False)
            _lib.load().mobi_tuning_reload()

    yield Tune()
    monkeypatch.undo()
    _lib.load().mobi_tuning_reload()
