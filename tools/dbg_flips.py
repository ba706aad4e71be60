"""Debug (GPU box): one DARTS golden iteration run twice - 9x9 first layers on risp_conv2d_toep_first / on risp_conv2d_k3 - recording
every convolution output; prints, launch by launch, how far the two runs are apart and how many outputs changed sign / zero-ness."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from conftest import load_golden  # noqa: E402
from test_host_logic import darts_opt, seed_darts  # noqa: E402
from reconfigisp_amd import convnets as CN  # noqa: E402
from reconfigisp_amd.codes.models import create_model  # noqa: E402

real_conv, real_small = CN.conv, CN.conv_small
log = []


def conv(x, pc, n, h, w, **kw):
    y = real_conv(x, pc, n, h, w, **kw)
    log.append(('conv k%d %d->%d%s epi %d' % (pc.k, pc.cin, pc.cout, ' T' if kw.get('transpose') else '', kw.get('epi', 0)), y.detach().clone()))
    return y


def conv_small(x, sc, n, h, w, **kw):
    y = real_small(x, sc, n, h, w, **kw)
    log.append(('small k%d %d->%d epi %d' % (sc.k, sc.cin, sc.cout, kw.get('epi', 0)), y.detach().clone()))
    return y


CN.conv, CN.conv_small = conv, conv_small
g = load_golden('darts_step')
runs = []
for first in (True, False):
    CN.TOEP_FIRST = os.environ.get('RISP_DBG_MODE', 'plain') if first else '0'
    log.clear()
    torch.manual_seed(0)
    model = create_model(darts_opt(torch.device('cuda')))
    seed_darts(model)
    data = tuple(torch.from_numpy(np.asarray(g[k])) for k in ('img', 'gt', 'val_img', 'val_gt'))
    model.feed_data(data)
    model.update_learning_rate(0, warmup_iter=-1)
    model.optimize_alphas()
    runs.append(list(log))
    print('alpha grads', [a.grad.flatten().tolist() for a in model.netG.alphas][1])
a, b = runs
print(len(a), len(b))
for i, ((na, ya), (nb, yb)) in enumerate(zip(a, b)):
    d = (ya - yb).abs().max().item() / (yb.abs().max().item() or 1.0)
    flips = ((ya > 0) != (yb > 0)).sum().item()
    if d > 0:
        extra = ''
        if flips:
            idx = torch.nonzero((ya > 0) != (yb > 0))[0].tolist()
            extra = ' first at %s: %.3e vs %.3e' % (idx, ya[tuple(idx)].item(), yb[tuple(idx)].item())
        print('%3d %-28s %-14s rel diff %.2e  sign changes %d%s' % (i, na, tuple(ya.shape), d, flips, extra))
