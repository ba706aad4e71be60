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


def test_device_resident_loader_matches_the_host_dataset(tmp_path):
    """`device_resident: true`: the frames of an RGGB2BGR dataset live in HBM and a batch is two gather kernels - the
    crops must be what the host path (data/rggb2bgr_datasets.py, the reference's formula) computes for the same
    (frame, row, col) selection, bit for bit."""
    import pickle
    from reconfigisp_amd.codes.data import create_dataloader, create_dataset
    from reconfigisp_amd.codes.data.device_loader import DeviceCropLoader
    from reconfigisp_amd.codes.data.image_io import write_png
    rng = np.random.default_rng(2)
    keys_n, keys_g, frames = [], [], []
    for i in range(5):
        raw = rng.integers(0, 16384, size=(40, 40), dtype=np.uint16)
        gt = rng.integers(0, 256, size=(40, 40, 3), dtype=np.uint8)
        write_png(str(tmp_path / ('n%d.png' % i)), raw)
        write_png(str(tmp_path / ('g%d.png' % i)), gt)
        keys_n.append('n%d.png' % i); keys_g.append('g%d.png' % i); frames.append((raw, gt))
    with open(tmp_path / 'meta_info.pkl', 'wb') as f:
        pickle.dump({'keys_ratio': keys_n, 'keys_gt': keys_g, 'resolution': 40}, f)
    dopt = {'mode': 'SID_Sony_Ratio_RGGB2BGR', 'phase': 'train', 'dataroot': str(tmp_path), 'data_type': 'img', 'data_size': 16,
            'sid_expo_in': None, 'sid_expo_gt': None, 'batch_size': 6, 'n_workers': 0, 'device_resident': True}
    ds = create_dataset(dopt)
    loader = create_dataloader(ds, dopt, {'dist': False, 'gpu_ids': [0]}, sampler=[0, 2, 3])
    assert isinstance(loader, DeviceCropLoader) and loader.raw.shape == (3, 40, 40) and loader.gt.shape == (3, 40, 40, 3)
    for batch in loader:
        sel = loader.last_selection.tolist()
        assert batch['noisy'].shape == (6, 1, 16, 16) and batch['gt'].shape == (6, 3, 16, 16) and batch['noisy'].is_cuda
        for b, (f, r, c) in enumerate(sel):
            raw, gt = frames[[0, 2, 3][f]]
            assert r % 2 == 0 and c % 2 == 0
            want_n = raw[r:r + 16, c:c + 16].astype(np.float32) / np.float32(16383.)
            want_g = np.transpose(gt[r:r + 16, c:c + 16], (2, 0, 1)).astype(np.float32) / np.float32(255.)
            assert np.array_equal(batch['noisy'][b, 0].cpu().numpy(), want_n)
            assert np.array_equal(batch['gt'][b].cpu().numpy(), want_g)
