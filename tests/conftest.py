import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


def rel_err(a, b):
    """||a-b||_2 / ||b||_2 in float64 (the parity gate of SURVEY §8(d))."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_tiny_decoder.pt")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_tiny_full.pt")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_c2_decoder.pt")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_dataset.pt")
