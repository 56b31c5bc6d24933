import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def bits(a):
    """bit pattern view of a float32/complex64 array for exact comparison"""
    return np.ascontiguousarray(a).view(np.uint32)


def same_values(a, b):
    """exact equality of float arrays where +0 == -0 (the sign of a zero never reaches a
    non-zero result in this path) and NaN == NaN"""
    a = np.ascontiguousarray(a).view(np.float32)
    b = np.ascontiguousarray(b).view(np.float32)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libloraref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def gpu():
    """torch + a gfx950 device + the in-tree HIP library; fails loudly when the library is missing"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lora_sdr_amd
    lora_sdr_amd.load()
    assert lora_sdr_amd.device_count() >= 1, "liblorahip.so sees no gfx950 device"
    return torch
