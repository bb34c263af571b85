import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_once():
    """Make sure libdagr_hip.so and the oracle exist (no-op when prebuilt, e.g. on the GPU box)."""
    import subprocess
    lib = os.path.join(ROOT, "dagr_amd", "lib", "libdagr_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", ROOT, "dagr_amd/lib/libdagr_hip.so"])
    from oracle import graph as og
    og.build()
    yield
