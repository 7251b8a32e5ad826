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


@pytest.fixture(params=[0, 1], ids=["gemm-four-wave", "gemm-eight-wave"])
def gemm_family(request):
    """Both GEMM kernel families under a model-level test (include/vgpt.h: vgpt_gemm_set_family): 0 = the four-wave kernel with
    the hand-scheduled loop wherever it applies (the default), 1 = the eight-wave LDS-DMA kernels only."""
    lib = importlib.import_module("video-gpt_amd._lib").load()
    prev = lib.vgpt_gemm_set_family(request.param)
    yield request.param
    lib.vgpt_gemm_set_family(prev)
