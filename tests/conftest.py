import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # PCFA_TEST_RANGE="a:b[,keep-substring]": collection indices a <= i < b plus every test whose name holds the substring,
    # in file order -- for bisecting an order-dependent failure (tools/dev)
    rng = os.environ.get("PCFA_TEST_RANGE")
    if rng:
        span, _, keep = rng.partition(",")
        a, b = (int(v) for v in span.split(":"))
        gpu_items = [it for it in items if "gpu" in it.keywords]     # (indices count the -m gpu tests only)
        chosen = [it for i, it in enumerate(gpu_items) if a <= i < b or (keep and keep in it.name)]
        config.hook.pytest_deselected(items=[it for it in items if it not in chosen])
        items[:] = chosen
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_ops():
    from oracle import ops as o
    return o


@pytest.fixture
def sepconv5_algo():
    """select("winograd" | "direct"): the process-wide switch of the SepConvGRU convolutions (pcfa_sepconv5_algo),
    restored after the test."""
    from pcfa_amd import _hip
    lib = _hip.load()
    prev = lib.pcfa_sepconv5_algo(-1)

    def select(name):
        lib.pcfa_sepconv5_algo({"winograd": 1, "direct": 0}[name])
    yield select
    lib.pcfa_sepconv5_algo(prev)
