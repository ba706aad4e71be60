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


class ErrorBudget:
    """Measured fp32 error budget: a float64 evaluation of the same graph is the truth, the reference's own float32
    result shows what fp32 arithmetic costs on this quantity, and the HIP result may cost at most ``factor`` times
    that (+ atol x max(1, max|truth|)):   max|got - ref64| <= factor * max|ref32 - ref64| + atol * scale.
    Checks are collected; ``finish()`` fails with the full list (RISP_BUDGET_REPORT=1 prints every measured pair)."""

    def __init__(self, factor=2.0, atol=4e-6):
        self.factor, self.atol, self.rows = factor, atol, []

    def __call__(self, got, ref32, ref64, what='', atol=None):
        import torch
        tonp = lambda v: np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, np.float64)
        got, ref32, ref64 = tonp(got), tonp(ref32), tonp(ref64)
        assert got.shape == ref64.shape == ref32.shape, '%s shapes %s %s %s' % (what, got.shape, ref32.shape, ref64.shape)
        if not got.size:
            return
        scale = max(1.0, np.abs(ref64).max())
        e_got, e_ref = np.abs(got - ref64).max(), np.abs(ref32 - ref64).max()
        bound = self.factor * e_ref + (self.atol if atol is None else atol) * scale
        self.rows.append((what, e_got, e_ref, scale, e_got <= bound))

    def finish(self):
        if os.environ.get('RISP_BUDGET_REPORT') == '1':
            for what, e_got, e_ref, scale, ok in self.rows:
                print('BUDGET %-40s hip %.3e  ref32 %.3e  ratio %7.2f  hip/scale %.2e %s' % (
                    what, e_got, e_ref, e_got / max(e_ref, 1e-30), e_got / scale, '' if ok else '  <-- OVER'))
        bad = ['%s: |hip - fp64| = %.3e, |reference fp32 - fp64| = %.3e, scale %.3g' % r[:4] for r in self.rows if not r[4]]
        assert not bad, 'over the fp32 error budget (%.1f x reference + %.1e x scale):\n  ' % (self.factor, self.atol) + '\n  '.join(bad)
