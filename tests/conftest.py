"""pytest configuration: markers and shared loaders."""
from __future__ import annotations

import importlib
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mf():
    """The product package (directory name has a hyphen, so import by string)."""
    return importlib.import_module("matrix-factorization-torch_amd")


def golden_files():
    return sorted(GOLDEN.glob("losses_*.npz"))
