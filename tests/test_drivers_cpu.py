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
        create_dataset({'mode': 'OnePlus_Rggb2Obj'})


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
