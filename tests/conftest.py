"""pytest configuration: registers the ``gpu`` marker and shared helpers.

``-m "not gpu"`` : oracle vs golden vectors, host logic, C-ABI symbol checks (CPU only).
``-m gpu``       : parity tests proper -- the HIP path, called through the C-ABI, vs the oracle.
"""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files():
    return sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name))
    return {k: g[k] for k in g.files}


def h_is_diagonal(H):
    H = np.asarray(H)
    off = H - np.einsum("...ii->...i", H)[..., None] * np.eye(H.shape[-1])
    return bool(np.all(off == 0))


def use_pytensor_or_standin():
    """The real PyTensor when it is installed, else tests/_pytensor_shim (a TEST-ONLY stand-in for the graph
    classes pymc_statespace_amd/op.py touches; see its _core.py).  Only tests call this."""
    try:
        import pytensor
    except ImportError:
        shim = os.path.join(ROOT, "tests", "_pytensor_shim")
        if shim not in sys.path:
            sys.path.insert(0, shim)
        import pytensor
    return pytensor


@pytest.fixture(scope="session")
def has_gpu():
    import torch

    return torch.cuda.is_available()
