"""GPU: randomized geometry for the four few-channel kernels of round 6 through the C ABI - risp_conv2d_tapout (3-cout layers, filter rows in
the rows of the matrix instruction), the tap-index first layer behind risp_conv2d_toep_first, risp_conv2d_thin5 (5x5 with at most 3 input
channels) and risp_conv2d_narrow3 (3x3 with at most 4 output channels): random batch, height, width, channel counts and epilogues inside
each kernel's domain, against the float64 convolution.  RISP_TEST_SEEDS=64 for soak runs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from test_gpu_toep import first_launch, with_table

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get('RISP_TEST_SEEDS', '8')) * 2


def rnd(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()


def launch(entry, d_kw, extra=()):
    from reconfigisp_amd import lib as L
    d = L.ConvDesc(**d_kw)
    L.call(entry, C.byref(d), *extra, None)
    torch.cuda.synchronize()


def close(y, ref, what):
    m = ref.abs().max().item() or 1.0
    e = (y.double() - ref).abs().max().item() / m
    assert not torch.isnan(y).any() and e < 3e-6, (what, e)


@pytest.mark.parametrize('seed', range(SEEDS))
def test_random_tap_row_layers(seed):
    from reconfigisp_amd import convnets as CN
    rng = np.random.default_rng(31000 + seed)
    k, cin, cout = int(rng.choice([5, 9])), int(rng.choice([16, 32, 48, 64])), int(rng.integers(1, 4))
    n, h, w = int(rng.integers(1, 5)), int(rng.integers(1, 90)), 4 * int(rng.integers(1, 80))
    wt = rnd(rng, cout, cin, k, k) * 0.05
    x = rnd(rng, n, cin, h, w) * float(rng.choice([1e-6, 1.0, 1e3]))
    has_add, has_bias, relu = rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.5
    add, b = rnd(rng, n, cout, h, w), rnd(rng, cout) * 0.1
    y = torch.full((n, cout, h, w), float('nan'), device='cuda')
    epi = (CN.EPI_ADD if has_add else 0) | (0 if has_bias else CN.EPI_NOBIAS) | (CN.EPI_RELU if relu else 0)
    launch('risp_conv2d_tapout', dict(N=n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=epi, add_c=cout if has_add else 0,
                                      x=x.data_ptr(), wpack=CN.tapout_weights(wt).data_ptr(), bias=b.data_ptr() if has_bias else None, cvals=None,
                                      add=add.data_ptr() if has_add else None, mask=None, y=y.data_ptr()), extra=(int(rng.choice([0, 4, 8, 32, 64])),))
    ref = TF.conv2d(x.double(), wt.double(), b.double() if has_bias else None, padding=k // 2) + (add.double() if has_add else 0)
    close(y, torch.relu(ref) if relu else ref, ('tap-row', k, cin, cout, n, h, w, epi))


@pytest.mark.parametrize('seed', range(SEEDS))
def test_random_tap_index_first_layers(seed):
    from reconfigisp_amd import convnets as CN
    rng = np.random.default_rng(32000 + seed)
    cout = int(rng.integers(1, 65))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(8, 80)), 4 * int(rng.integers(2, 70))
    wt, b = rnd(rng, cout, 3, 9, 9) * 0.1, rnd(rng, cout) * 0.05
    x = rnd(rng, n, 3, h, w) * float(rng.choice([1e-3, 1.0, 50.0]))
    table = rnd(rng, n, cout, 9, 9) * 0.1 if rng.random() < 0.6 else None
    relu = rng.random() < 0.7
    epi = (CN.EPI_RELU if relu else 0) | (CN.EPI_CASEBIAS if table is not None else 0)
    y = first_launch(x, CN.toep_first_weights(wt), b, n, h, w, 3, cout, epi, table)
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=4)
    if table is not None:
        ref = with_table(ref, table, h, w)
    close(y, torch.relu(ref) if relu else ref, ('tap-index', cout, n, h, w, epi))


@pytest.mark.parametrize('seed', range(SEEDS))
def test_random_thin_input_layers(seed):
    from reconfigisp_amd import convnets as CN
    rng = np.random.default_rng(33000 + seed)
    cin, cout = int(rng.integers(1, 4)), int(rng.choice([32, 64]))
    n, h, w = int(rng.integers(1, 5)), int(rng.integers(1, 90)), int(rng.integers(1, 300))
    wt, b = rnd(rng, cout, cin, 5, 5) * 0.05, rnd(rng, cout) * 0.1
    x = rnd(rng, n, cin, h, w) * float(rng.choice([1e-6, 1.0, 1e3]))
    has_mask, has_bias, relu = rng.random() < 0.6, rng.random() < 0.5, rng.random() < 0.4
    mask = rnd(rng, n, cout, h, w)
    y = torch.full((n, cout, h, w), float('nan'), device='cuda')
    epi = (CN.EPI_MASK if has_mask else 0) | (0 if has_bias else CN.EPI_NOBIAS) | (CN.EPI_RELU if relu else 0)
    launch('risp_conv2d_thin5', dict(N=n, H=h, W=w, cin=cin, cout=cout, ksize=5, load_mode=0, cin_img=0, epilogue=epi, add_c=0, x=x.data_ptr(),
                                     wpack=CN.thin5_weights(wt).data_ptr(), bias=b.data_ptr() if has_bias else None, cvals=None, add=None,
                                     mask=mask.data_ptr() if has_mask else None, y=y.data_ptr()))
    ref = TF.conv2d(x.double(), wt.double(), b.double() if has_bias else None, padding=2)
    if has_mask:
        ref = ref * (mask > 0).double()
    close(y, torch.relu(ref) if relu else ref, ('thin input', cin, cout, n, h, w, epi))


@pytest.mark.parametrize('seed', range(SEEDS))
def test_random_narrow_3x3_layers(seed):
    from reconfigisp_amd import convnets as CN
    rng = np.random.default_rng(34000 + seed)
    cin, cout = int(rng.choice([16, 32, 48, 64])), int(rng.integers(1, 5))
    n, h, w = int(rng.integers(1, 6)), int(rng.integers(1, 90)), int(rng.integers(1, 300))
    wt, b = rnd(rng, cout, cin, 3, 3) * 0.05, rnd(rng, cout) * 0.1
    x = rnd(rng, n, cin, h, w) * float(rng.choice([1e-6, 1.0, 1e3]))
    shuf, has_bias, relu = cout == 4 and rng.random() < 0.6, rng.random() < 0.5, rng.random() < 0.4
    y = torch.full((n, 1, 2 * h, 2 * w) if shuf else (n, cout, h, w), float('nan'), device='cuda')
    epi = (CN.EPI_SHUFFLE2 if shuf else 0) | (0 if has_bias else CN.EPI_NOBIAS) | (CN.EPI_RELU if relu else 0)
    launch('risp_conv2d_narrow3', dict(N=n, H=h, W=w, cin=cin, cout=cout, ksize=3, load_mode=0, cin_img=0, epilogue=epi, add_c=0, x=x.data_ptr(),
                                       wpack=CN.narrow3_weights(wt).data_ptr(), bias=b.data_ptr() if has_bias else None, cvals=None, add=None, mask=None,
                                       y=y.data_ptr()))
    ref = TF.conv2d(x.double(), wt.double(), b.double() if has_bias else None, padding=1)
    ref = torch.relu(ref) if relu else ref
    close(y, TF.pixel_shuffle(ref, 2) if shuf else ref, ('narrow 3x3', cin, cout, n, h, w, epi))
