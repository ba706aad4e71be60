"""Option files, synthetic data and the test driver's host logic (operator seam bound to the oracle)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

CODES = os.path.join(ROOT, 'reconfigisp_amd', 'codes')


def test_option_files_parse(tmp_path):
    from reconfigisp_amd.codes.options import options as option
    for f in sorted(glob.glob(os.path.join(CODES, 'options', '*', '*.yml'))):
        opt = option.parse(f, is_train='train' in f)
        assert opt['network_G']['which_model_G'] and opt['model'] in ('darts', 'darts_ft', 'isp')
        assert opt['datasets'][list(opt['datasets'])[0]]['phase'] in ('train', 'test')
    text = option.dict2str({'a': 1, 'b': {'c': 2}})
    assert 'b:[' in text and 'c: 2' in text
    assert option.dict_to_nonedict({'a': {'b': 1}})['a']['zzz'] is None


def test_synthetic_raw_contract():
    from reconfigisp_amd.codes.data import create_dataloader, create_dataset
    ds = create_dataset({'mode': 'Synthetic_RGGB2BGR', 'data_size': 32, 'n_images': 6, 'seed': 3, 'phase': 'test'})
    item = ds[0]
    assert item['noisy'].shape == (1, 32, 32) and item['gt'].shape == (3, 32, 32)
    codes = item['noisy'] * 1023
    assert torch.allclose(codes, codes.round(), atol=1e-3)           # 10-bit codes k/1023
    assert 0 <= item['noisy'].min() and item['noisy'].max() <= 1
    batch = next(iter(create_dataloader(ds, {'phase': 'test'})))
    assert batch['noisy'].shape == (1, 1, 32, 32)
    with pytest.raises(NotImplementedError, match='not recognized'):
        create_dataset({'mode': 'nope'})
    with pytest.raises(NotImplementedError, match='outside the scope'):
        create_dataset({'mode': 'OnePlus_Rggb2Obj'})                 # detection labels for the YOLOv3 loss


def test_test_driver_end_to_end(monkeypatch, tmp_path, capsys):
    import reconfigisp_amd.functional as F
    from oracle_backend import OracleImpl
    monkeypatch.setattr(F, '_IMPL', OracleImpl)
    from reconfigisp_amd.codes import test as driver
    from reconfigisp_amd.codes.options import options as option
    real_parse = option.parse

    def parse(path, is_train=True):
        opt = real_parse(path, is_train)
        opt['gpu_ids'] = None                                       # CPU device for the seam
        opt['datasets']['test'].update(data_size=32, n_images=2)
        opt['path']['results_root'] = str(tmp_path / 'results')
        opt['path']['log'] = str(tmp_path / 'results')
        return opt
    monkeypatch.setattr(option, 'parse', parse)
    driver.main(['--opt', os.path.join(CODES, 'options', 'test', 'synthetic_test.yml')])
    out = capsys.readouterr().out
    assert 'PSNR in:' in out and 'PSNR out:' in out
    files = glob.glob(str(tmp_path / 'results' / '*' / '*.ppm'))
    assert len(files) == 2
    w = int(open(files[0], 'rb').read(20).split()[1])
    assert w == 32 * 5                                               # input | 3 stages | gt


def test_alias_install():
    import sys
    from reconfigisp_amd import codes
    codes.install_aliases()
    import models.modules.tools_origin as T                           # the reference's import path
    import whitebalance
    assert T.WbQuadratic.__module__.startswith('reconfigisp_amd.codes')
    assert whitebalance.WhiteBalance().run.__func__ is not None
    for name in ('models', 'options', 'utils', 'data', 'whitebalance', 'gamma', 'demosaic',
                 'globaltonemapping', 'spatialnoisereduction'):
        sys.modules.pop(name, None)
    for name in [k for k in sys.modules if k.split('.')[0] in ('models', 'options', 'utils', 'data')]:
        sys.modules.pop(name, None)


def test_png_reader_round_trip_and_filters(tmp_path):
    """data/image_io.py against hand-built files: every scan-line filter type, 8/16 bit, grey / RGB / RGBA."""
    import struct, zlib
    from reconfigisp_amd.codes.data.image_io import read_image, read_png, write_png
    rng = np.random.default_rng(0)
    g16 = rng.integers(0, 16384, size=(10, 14), dtype=np.uint16)
    bgr = rng.integers(0, 256, size=(9, 7, 3), dtype=np.uint8)
    write_png(str(tmp_path / 'g.png'), g16)
    write_png(str(tmp_path / 'c.png'), bgr)
    assert np.array_equal(read_image(str(tmp_path / 'g.png')), g16) and read_png(str(tmp_path / 'g.png')).dtype == np.uint16
    assert np.array_equal(read_image(str(tmp_path / 'c.png')), bgr)              # BGR in, BGR out (file holds RGB)
    np.save(str(tmp_path / 'f.npy'), g16)
    assert np.array_equal(read_image(str(tmp_path / 'f.npy')), g16)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)

    def encode(rows, bpp, ftypes):                   # the PNG specification's filters, applied forwards
        out, prev = b'', [0] * len(rows[0])
        for y, row in enumerate(rows):
            ft, line = ftypes[y % len(ftypes)], []
            for i, v in enumerate(row):
                a = row[i - bpp] if i >= bpp else 0
                b, c = prev[i], (prev[i - bpp] if i >= bpp else 0)
                pred = {0: 0, 1: a, 2: b, 3: (a + b) >> 1, 4: paeth(a, b, c)}[ft]
                line.append((v - pred) & 0xFF)
            out += bytes([ft]) + bytes(line)
            prev = row
        return out

    def chunk(kind, payload):
        return struct.pack('>I', len(payload)) + kind + payload + struct.pack('>I', zlib.crc32(kind + payload) & 0xFFFFFFFF)

    for ch, ctype, depth in ((1, 0, 16), (3, 2, 8), (4, 6, 8), (1, 0, 8)):
        h, w = 11, 6
        img = rng.integers(0, 2 ** depth, size=(h, w, ch)).astype(np.uint16 if depth == 16 else np.uint8)
        raw = img.astype('>u2').view(np.uint8).reshape(h, -1) if depth == 16 else img.reshape(h, -1)
        body = encode([r.tolist() for r in raw], ch * depth // 8, [0, 1, 2, 3, 4])
        path = str(tmp_path / ('t%d_%d.png' % (ctype, depth)))
        with open(path, 'wb') as f:
            f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, ctype, 0, 0, 0)) +
                    chunk(b'IDAT', zlib.compress(body)) + chunk(b'IEND', b''))
        got = read_png(path)
        want = img[..., 0] if ch == 1 else img[..., [2, 1, 0] + ([3] if ch == 4 else [])]
        assert got.dtype == img.dtype and np.array_equal(got, want), (ctype, depth)


def test_file_backed_rggb2bgr_datasets(tmp_path):
    """The reference's four RGGB2BGR dataset modes (data/*_rggb2bgr*_dataset.py) on a tiny on-disk dataset."""
    import pickle, random
    from reconfigisp_amd.codes.data import create_dataloader, create_dataset
    from reconfigisp_amd.codes.data.image_io import write_png
    rng = np.random.default_rng(1)
    root = tmp_path / 'sid'
    root.mkdir()
    keys_n, keys_g, frames = [], [], {}
    for i, (ein, egt) in enumerate((('0.1s', '10s'), ('0.04s', '10s'), ('0.1s', '30s'))):
        kn, kg = 'n%d_%s.png' % (i, ein), 'g%d_%s.png' % (i, egt)
        raw = rng.integers(0, 16384, size=(24, 24), dtype=np.uint16)
        gt = rng.integers(0, 256, size=(24, 24, 3), dtype=np.uint8)
        write_png(str(root / kn), raw)
        write_png(str(root / kg), gt)
        keys_n.append(kn); keys_g.append(kg); frames[kn] = (raw, gt)
    with open(root / 'meta_info.pkl', 'wb') as f:
        pickle.dump({'keys_ratio': keys_n, 'keys_noisy': keys_n, 'keys_gt': keys_g, 'resolution': 24}, f)
    base = {'dataroot': str(root), 'data_type': 'img', 'data_size': 8, 'sid_expo_in': '0.1s', 'sid_expo_gt': '10s'}
    ds = create_dataset(dict(base, mode='SID_Sony_Ratio_RGGB2BGR', phase='train'))
    assert len(ds) == 1                                             # exposure filter keeps the (0.1s, 10s) pair only
    random.seed(4)
    item = ds[0]
    random.seed(4)
    r = (random.randint(0, 16) // 2) * 2
    c = (random.randint(0, 16) // 2) * 2
    raw, gt = frames[keys_n[0]]
    assert item['noisy'].shape == (1, 8, 8) and item['noisy'].dtype == np.float32 and 'name' not in item
    assert np.array_equal(item['noisy'][0], raw[r:r + 8, c:c + 8].astype(np.float32) / np.float32(16383.))
    assert np.array_equal(item['gt'], np.transpose(gt[r:r + 8, c:c + 8], (2, 0, 1)).astype(np.float32) / np.float32(255.))
    test = create_dataset(dict(base, mode='SID_Sony_Ratio_Test_RGGB2BGR', phase='test', sid_expo_in=None, sid_expo_gt=None))
    assert len(test) == 3 and test[1]['name'] == 'n1_0.04s' and test[1]['noisy'].shape == (1, 8, 8)     # top-left corner
    s7 = create_dataset(dict(base, mode='S7ISP_RGGB2BGR', phase='train'))
    assert len(s7) == 3 and float(s7[2]['noisy'].max()) > 1.0       # /1023: the caller's data decides the range
    s7t = create_dataset(dict(base, mode='S7ISP_RGGB2BGR_Test', phase='test'))
    batch = next(iter(create_dataloader(s7t, {'phase': 'test'})))
    assert batch['noisy'].shape == (1, 1, 24, 24) and batch['gt'].shape == (1, 3, 24, 24) and batch['name'] == ['n0_0.1s']
    with pytest.raises(NotImplementedError, match='memcached'):
        create_dataset(dict(base, mode='S7ISP_RGGB2BGR', data_type='mc'))


REF_OPTIONS = '/root/reference/codes/options'


@pytest.mark.skipif(not os.path.isdir(REF_OPTIONS), reason='dev container only: the nine shipped YAMLs live under /root/reference')
def test_reference_option_files_parse_like_the_reference(monkeypatch, capsys):
    """options/options.py:8-62 on the reference's own nine option files: the mirror's parse must derive the same fields (phase,
    data_type, mode after '_mc' stripping, meta_device, every path.*, the debug overrides) and leave the same tree as the
    reference's parse did (tests/golden/options.npz, written by make_golden.py::gold_options from the IMPORTED reference)."""
    import json
    import sys
    from conftest import GOLDEN, load_golden
    sys.path.insert(0, GOLDEN)
    from option_canon import canonical
    from reconfigisp_amd.codes.options import options as option
    monkeypatch.setenv('CUDA_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES', ''))      # parse() exports it: restore afterwards
    want = json.loads(str(load_golden('options')['table']))
    files = sorted(glob.glob(os.path.join(REF_OPTIONS, '*', '*.yml')))
    assert len(files) == 9 and sorted(want) == [os.path.relpath(f, REF_OPTIONS) for f in files]
    for f in files:
        rel = os.path.relpath(f, REF_OPTIONS)
        got = canonical(option.parse(f, is_train=rel.startswith('train')))
        assert got['derived'] == want[rel]['derived'], rel
        assert got['tree_sha256'] == want[rel]['tree_sha256'], rel
    assert 'export CUDA_VISIBLE_DEVICES=' in capsys.readouterr().out          # every machine but the authors' cluster (:13-17)


@pytest.mark.skipif(not os.path.isdir(REF_OPTIONS), reason='dev container only: the nine shipped YAMLs live under /root/reference')
@pytest.mark.filterwarnings('ignore:Detected call of')
def test_reference_option_files_build_their_models(monkeypatch, tmp_path):
    """create_model(opt) from each shipped YAML on the operator seam (CPU): the six image-loss files construct their networks,
    optimizers and schedulers - for the two searches that is the 5-slot super-net with n_step 3 and prune threshold 0.2 of
    options/train/SID_search.yml:31-34 - and the three YOLOv3-loss files are refused by name (out of scope)."""
    import reconfigisp_amd.functional as F
    from oracle_backend import OracleImpl
    from reconfigisp_amd.codes.models import create_model
    from reconfigisp_amd.codes.options import options as option
    monkeypatch.setattr(F, '_IMPL', OracleImpl)
    monkeypatch.setenv('CUDA_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES', ''))
    built = {}
    for f in sorted(glob.glob(os.path.join(REF_OPTIONS, '*', '*.yml'))):
        rel = os.path.relpath(f, REF_OPTIONS)
        opt = option.parse(f, is_train=rel.startswith('train'))
        opt['gpu_ids'], opt['dist'] = None, False                     # the seam's device
        opt['network_G']['module_path'] = None                        # the proxies' .pth files are not distributed: seeded weights
        if 'individual_module_paths' in opt['network_G']:
            opt['network_G']['individual_module_paths'] = [None] * len(opt['network_G']['individual_module_paths'])
        opt['path']['pretrain_model_G'] = None
        for k in ('experiments_root', 'models', 'training_state', 'log', 'val_images', 'results_root'):
            if k in opt['path']:
                opt['path'][k] = str(tmp_path / k)
        opt = option.dict_to_nonedict(opt)
        if 'yolo' in opt['model']:
            with pytest.raises(NotImplementedError):
                create_model(opt)
            continue
        model = create_model(opt)
        built[rel] = model
        if opt['model'].startswith('darts'):
            assert len(model.netG.alphas) == opt['network_G']['n_step'] + 2 == 5
            assert model.netG.threshold == opt['network_G']['prune_threshold'] == 0.2
        if rel.startswith('train'):
            assert len(model.optimizers) == len(model.schedulers) >= 1
    assert sorted(built) == ['test/S7ISP_test.yml', 'test/SID_test.yml', 'train/S7ISP_isp.yml', 'train/S7ISP_search.yml',
                             'train/SID_isp.yml', 'train/SID_search.yml']
