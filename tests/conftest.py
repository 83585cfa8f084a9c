"""pytest configuration: markers, import path and the hyphenated package loader."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_pkg():
    """Import the host package that lives in the (non-identifier) dir gpu-ntt_amd/."""
    if "gpu_ntt_amd" in sys.modules:
        return sys.modules["gpu_ntt_amd"]
    pkg_dir = os.path.join(ROOT, "gpu-ntt_amd")
    spec = importlib.util.spec_from_file_location(
        "gpu_ntt_amd", os.path.join(pkg_dir, "__init__.py"),
        submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gpu_ntt_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
