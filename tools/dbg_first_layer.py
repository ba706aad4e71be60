"""Debug (GPU box): inside the DARTS golden iteration, every 9x9 first-layer launch is run on risp_conv2d_toep_first AND on
risp_conv2d_k3 and compared with float64 (conv + border-case table + ReLU): largest error of each, and how many ReLU decisions
differ from float64's."""
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from conftest import load_golden  # noqa: E402
from test_host_logic import darts_opt, seed_darts  # noqa: E402
from reconfigisp_amd import convnets as CN  # noqa: E402
from reconfigisp_amd.codes.models import create_model  # noqa: E402


def border_case(v, L, P=4):
    return v if v < P else (2 * P - (L - 1 - v) if v >= L - P else P)


real_init, real_conv = CN.PackedConv.__init__, CN.conv


def init(self, weight, bias):
    real_init(self, weight, bias)
    self._w, self._b = weight.detach().clone(), bias.detach().clone()


def conv(x, pc, n, h, w, **kw):
    members = getattr(pc, 'members', [pc])
    if getattr(members[0], 'toep_first', None) is None or kw.get('transpose'):
        return real_conv(x, pc, n, h, w, **kw)
    y = real_conv(x, pc, n, h, w, **kw)
    CN.TOEP_FIRST = '0'
    y3 = real_conv(x, pc, n, h, w, **kw)
    CN.TOEP_FIRST = os.environ.get('RISP_DBG_MODE', 'plain')
    G = len(members)
    xs = x.double()
    if kw.get('load', 0) == CN.LOAD_UNSHUFFLE2:
        xs = TF.pixel_unshuffle(xs, 2)
    iy = torch.tensor([border_case(v, h) for v in range(h)], device='cuda')
    ix = torch.tensor([border_case(v, w) for v in range(w)], device='cuda')
    for g, m in enumerate(members):
        lin = TF.conv2d(xs, m._w.double()[:, :xs.shape[1]], m._b.double(), padding=4)
        if kw.get('cvals') is not None:
            t = kw['cvals'][g * n:(g + 1) * n].view(n, -1, 9, 9).double()
            lin = lin + t[:, :, iy][:, :, :, ix]
        ref = torch.relu(lin) if kw.get('epi', 0) & CN.EPI_RELU else lin
        s = slice(g * n, (g + 1) * n)
        mag = ref.abs().max().item()
        e, e3 = (y[s].double() - ref).abs(), (y3[s].double() - ref).abs()
        flips = ((y[s] > 0) != (lin > 0)).sum().item(), ((y3[s] > 0) != (lin > 0)).sum().item()
        pos = torch.nonzero(e == e.max())[0].tolist()
        dis = torch.nonzero((y[s] > 0) != (y3[s] > 0))
        for q in dis[:4].tolist():
            print('   sign differs at', q, 'toep %.4e k3 %.4e float64 %.6e' % (y[s][tuple(q)].item(), y3[s][tuple(q)].item(), lin[tuple(q)].item()), flush=True)
        print('first layer %dx%d cin %d member %d: toep max %.2e rms %.2e (at %s) flips %d | k3 max %.2e rms %.2e flips %d | mag %.2e, |x| max %.2e' % (
            h, w, xs.shape[1], g, e.max().item() / mag, e.pow(2).mean().sqrt().item() / mag, pos, flips[0], e3.max().item() / mag,
            e3.pow(2).mean().sqrt().item() / mag, flips[1], mag, xs.abs().max().item()), flush=True)
    return y


CN.PackedConv.__init__ = init
CN.TOEP_FIRST = os.environ.get('RISP_DBG_MODE', 'plain')
CN.conv = conv
g = load_golden('darts_step')
model = create_model(darts_opt(torch.device('cuda')))
seed_darts(model)
data = tuple(torch.from_numpy(__import__('numpy').asarray(g[k])) for k in ('img', 'gt', 'val_img', 'val_gt'))
model.feed_data(data)
model.update_learning_rate(0, warmup_iter=-1)
model.optimize_alphas()
