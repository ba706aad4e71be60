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
    # the CPU oracle runs beside every GPU test; on a many-core host torch's default of one thread per core makes its
    # small tensor ops (unfold / sort / conv on 16 x 16 .. 70 x 64 images) crawl
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


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
    """Measured fp32 error budget.  A float64 evaluation of the same graph is the truth; the reference's (or the
    oracle's) own float32 result shows what fp32 arithmetic costs; the HIP result may cost at most ``factor`` times that:

        max|hip - fp64| / scale  <=  factor * E_ref(family) + atol,      scale = max(1, max|fp64|)

    E_ref(family) is the LARGEST relative fp32 error of the reference over the quantities of one family (slot outputs,
    architecture gradients, parameter gradients, ...) in the test.  Per family, not per tensor, because everything
    downstream of a ReLU mask is discontinuous in the pre-activations: a pre-activation within an ulp of zero is
    clipped by one fp32 implementation and not by another, and which tensor that lands in is arbitrary (measured: the
    reference is 1.2e-4 off on the second DARTS iteration's architecture gradients and 8e-7 on the first's; the HIP
    path the other way round).  ``atol`` = 4e-6 (64 ulp of the tensor's magnitude) covers the hardware exp2 / log2
    approximations and the fixed summation order of the reductions.  The north-star bar is 1e-4.
    Checks are collected; ``finish()`` fails with the full list (RISP_BUDGET_REPORT=1 prints every measured pair)."""

    def __init__(self, factor=2.0, atol=4e-6):
        self.factor, self.atol, self.rows = factor, atol, []

    def __call__(self, got, ref32, ref64, what='', family=None):
        import torch
        tonp = lambda v: np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, np.float64)
        got, ref32, ref64 = tonp(got), tonp(ref32), tonp(ref64)
        assert got.shape == ref64.shape == ref32.shape, '%s shapes %s %s %s' % (what, got.shape, ref32.shape, ref64.shape)
        if not got.size:
            return
        scale = max(1.0, np.abs(ref64).max())
        self.rows.append((what, np.abs(got - ref64).max() / scale, np.abs(ref32 - ref64).max() / scale, family or what))

    def finish(self):
        fam = {}
        for _, _, e_ref, family in self.rows:
            fam[family] = max(fam.get(family, 0.0), e_ref)
        bad = []
        for what, e_got, e_ref, family in self.rows:
            ok = e_got <= self.factor * fam[family] + self.atol
            if os.environ.get('RISP_BUDGET_REPORT') == '1':
                print('BUDGET %-40s hip %.2e  ref32 %.2e  family %-12s %.2e %s' % (what, e_got, e_ref, family, fam[family],
                                                                                  '' if ok else '  <-- OVER'))
            if not ok:
                bad.append('%s: |hip - fp64| = %.3e of scale; reference fp32: %.3e (family %s: %.3e)' % (
                    what, e_got, e_ref, family, fam[family]))
        assert not bad, 'over the fp32 error budget (%.1f x reference + %.1e):\n  ' % (self.factor, self.atol) + '\n  '.join(bad)
