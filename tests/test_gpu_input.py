"""GPU: the device-resident input path (integer gathers: bit-exact) vs the numpy restatement of the
reference's dataset code."""
import numpy as np
import pytest
import torch

import isp_oracle as O

pytestmark = pytest.mark.gpu


def test_raw_and_gt_crops_bit_exact():
    from reconfigisp_amd.codes.data import gpu_input as G
    rng = np.random.Generator(np.random.PCG64(3))
    raw = rng.integers(0, 16384, size=(3, 96, 130), dtype=np.uint16)
    gt = rng.integers(0, 256, size=(3, 96, 130, 3), dtype=np.uint8)
    import random
    sel = G.even_crop_positions(7, 3, (96, 130), (48, 64), random.Random(5))
    assert sel.shape == (7, 3) and (sel[:, 1:] % 2 == 0).all()
    for white in (1023.0, 16383.0):
        got = G.raw_crops(torch.from_numpy(raw).cuda(), sel, (48, 64), white).cpu().numpy()
        assert np.array_equal(got, O.crop_raw(raw, sel.numpy(), (48, 64), white))
    got = G.gt_crops(torch.from_numpy(gt).cuda(), sel, (48, 64)).cpu().numpy()
    assert np.array_equal(got, O.crop_gt(gt, sel.numpy(), (48, 64)))


@pytest.mark.parametrize('shape,desired', [((300, 400), 128), ((3000, 4000), 1024), ((480, 640), 256)])
def test_resize_rggb_letterbox(shape, desired):
    from reconfigisp_amd.codes.data import gpu_input as G
    rng = np.random.Generator(np.random.PCG64(4))
    img = rng.integers(0, 1024, size=shape, dtype=np.uint16)
    got, top = G.resize_rggb_letterbox(torch.from_numpy(img).cuda(), desired)
    ref, rtop = O.resize_rggb_letterbox(img, desired)
    assert top == rtop
    assert np.array_equal(got.cpu().numpy(), ref)
    assert (got[:2 * (top // 2)].cpu().numpy() == 0).all()


import os
_FUZZ = int(os.environ.get('RISP_TEST_SEEDS', '8'))               # soak runs: RISP_TEST_SEEDS=64


@pytest.mark.parametrize('seed', range(_FUZZ))
def test_random_input_geometry_bit_exact(seed):
    """Random frame / crop / letterbox geometry: crops and the resize-by-quad letterbox stay bit-exact."""
    import random
    from reconfigisp_amd.codes.data import gpu_input as G
    rng = np.random.Generator(np.random.PCG64(300 + seed))
    fh, fw = 2 * int(rng.integers(20, 90)), 2 * int(rng.integers(20, 120))
    ch, cw = 2 * int(rng.integers(4, fh // 2 + 1)), 2 * int(rng.integers(4, fw // 2 + 1))
    nf, nc = int(rng.integers(1, 4)), int(rng.integers(1, 9))
    raw = rng.integers(0, 16384, size=(nf, fh, fw), dtype=np.uint16)
    gt = rng.integers(0, 256, size=(nf, fh, fw, 3), dtype=np.uint8)
    sel = G.even_crop_positions(nc, nf, (fh, fw), (ch, cw), random.Random(seed))
    white = float(rng.choice([1023.0, 16383.0]))
    assert np.array_equal(G.raw_crops(torch.from_numpy(raw).cuda(), sel, (ch, cw), white).cpu().numpy(),
                          O.crop_raw(raw, sel.numpy(), (ch, cw), white))
    assert np.array_equal(G.gt_crops(torch.from_numpy(gt).cuda(), sel, (ch, cw)).cpu().numpy(),
                          O.crop_gt(gt, sel.numpy(), (ch, cw)))
    desired = 32 * int(rng.integers(2, 10))
    lh, lw = (fh * 2, fw * 2) if fh <= fw else (fw * 2, fh * 2)            # landscape, like every OnePlus frame
    img = rng.integers(0, 1024, size=(lh, lw), dtype=np.uint16)
    got, top = G.resize_rggb_letterbox(torch.from_numpy(img).cuda(), desired)
    ref, rtop = O.resize_rggb_letterbox(img, desired)
    assert top == rtop and np.array_equal(got.cpu().numpy(), ref)
    if lw > lh + 8:        # a portrait frame would need negative padding (oneplus_rggb2obj_dataset.py:127-133 fails too)
        with pytest.raises(RuntimeError):
            G.resize_rggb_letterbox(torch.from_numpy(np.ascontiguousarray(img.T)).cuda(), desired)
