import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def assert_close(a, b, rtol=1e-4, atol=1e-6, what='', floor=0.05):
    """The parity bar of BASELINE.json: <= 1e-4 relative fp32.

    Element-wise |a-b| <= atol + rtol * max(|b|, floor * max|b|): relative to the element, except
    that elements smaller than `floor` (5 %) of the tensor's largest magnitude are judged against
    that floor (a 14-layer fp32 CNN cannot hold 1e-4 relative on outputs that cancel to ~0)."""
    import torch
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, '%s shape %s vs %s' % (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    err = np.abs(a - b)
    bound = atol + rtol * np.maximum(np.abs(b), floor * scale)
    bad = err > bound
    assert not bad.any(), '%s: %d/%d out of tolerance, max err %.3e (ref scale %.3e)' % (
        what, bad.sum(), bad.size, err.max(), scale)
