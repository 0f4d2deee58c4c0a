import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture
def load_golden():
    return golden


def assert_close(a, b, rtol, atol, what="", scale_atol=0.0):
    """|a - b| <= atol + scale_atol * max|b| + rtol * |b| elementwise.
    `scale_atol` makes the absolute part relative to the tensor's magnitude: sums of many O(1) terms (scan outputs,
    attention rows) cancel to small elements whose error is set by the size of the terms, not of the element."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + scale_atol * (np.abs(b).max() if b.size else 0.0) + rtol * np.abs(b)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: max violation at {i}: got {a[i]!r} want {b[i]!r} (|err|={err[i]:.3e}, "
                             f"tol={tol[i]:.3e}); max abs err {err.max():.3e}")


@pytest.fixture
def allow_torch_sdpa(monkeypatch):
    """Tiny test models (hidden 64 / 128: head_dim 4 / 8) have no MFMA attention kernel (head_dim in {24, 32, 48, 64, 72});
    their attention core runs on torch's SDPA, which the product only does when asked to (dimsum_amd.utils.note_torch_path)."""
    monkeypatch.setenv("DIMSUM_ALLOW_TORCH_SDPA", "1")
