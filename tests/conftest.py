"""pytest configuration: `gpu` marker, repo root on sys.path, package import helper."""
import importlib
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
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def vg():
    """The product package (directory `video-gpt_amd`)."""
    return importlib.import_module("video-gpt_amd")


@pytest.fixture(scope="session")
def ops(vg):
    return importlib.import_module("video-gpt_amd.ops")
