"""pytest configuration: registers the ``gpu`` marker and puts the product package
(``pytv-4d_amd/``) and the repo root (for ``oracle``) on ``sys.path``."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytv-4d_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCHEMES = ("upwind", "downwind", "central", "hybrid")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """Make sure the HIP library exists (hipcc cross-compiles gfx950 without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pytv4d_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not mod.up_to_date():
        mod.build(verbose=False)
    yield


@pytest.fixture
def tvopt(monkeypatch):
    """Set one of the library's tuning options for the duration of a test: ``tvopt("TV_ZCHUNK", 3)``.  The library
    reads the environment only once (when it is loaded), so the option goes through tv_set_option(); the environment
    variable is set as well for the ranks a test spawns."""
    from pytv import _native
    unset = -2 ** 31
    before = {}

    def set_(name, value):
        if name not in before:
            before[name] = _native.get_option(name, unset)
        monkeypatch.setenv(name, str(value))
        _native.set_option(name, int(value))
    yield set_
    for name, old in before.items():
        _native.set_option(name, None if old == unset else old)
