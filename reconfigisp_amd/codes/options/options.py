"""Option files -> nested OrderedDict, with the derived fields the reference adds in
options/options.py:8-62 (phase / data_type per dataset, meta_device, experiment or result folders),
so its nine shipped YAMLs load unchanged.  Host-side only."""
import os
import os.path as osp

import yaml

from ..utils.util import OrderedYaml

Loader, Dumper = OrderedYaml()
_PACKAGE_ROOT = osp.abspath(osp.join(osp.dirname(__file__), osp.pardir, osp.pardir))


def _select_devices(opt):
    # every machine except the authors' cluster pins the visible GPUs (PyTorch-ROCm honours
    # CUDA_VISIBLE_DEVICES as well)
    if opt['machine'] == 'st_sh34':
        return
    ids = ','.join(map(str, opt['gpu_ids']))
    os.environ['CUDA_VISIBLE_DEVICES'] = ids
    print('export CUDA_VISIBLE_DEVICES=' + ids)


def _annotate_datasets(datasets):
    for key, ds in datasets.items():
        ds['phase'] = key.split('_')[0]
        root = ds.get('dataroot') or ''
        ds['data_type'] = 'lmdb' if str(root).endswith('lmdb') else 'img'
        if ds['mode'].endswith('mc'):                 # memcached-backed variant
            ds['data_type'], ds['mode'] = 'mc', ds['mode'].replace('_mc', '')


def _expand_paths(paths, name, is_train):
    for key in list(paths):
        if paths[key] and key != 'strict_load':
            paths[key] = osp.expanduser(paths[key])
    paths['root'] = _PACKAGE_ROOT
    if not is_train:
        out = osp.join(_PACKAGE_ROOT, 'results', name)
        paths['results_root'] = paths['log'] = out
        return
    out = osp.join(_PACKAGE_ROOT, 'experiments', name)
    paths['experiments_root'] = paths['log'] = out
    for sub in ('models', 'training_state', 'val_images'):
        paths[sub] = osp.join(out, sub)


def parse(opt_path, is_train=True):
    with open(opt_path) as stream:
        opt = yaml.load(stream, Loader=Loader)
    opt['is_train'] = is_train
    _select_devices(opt)
    _annotate_datasets(opt['datasets'])
    opt['meta_device'] = 'Meta' in opt['network_G']['which_model_G']
    _expand_paths(opt['path'], opt['name'], is_train)
    if is_train and 'debug' in opt['name']:            # short cycles for debug runs
        opt['train']['val_freq'] = 8
        opt['logger']['save_checkpoint_freq'] = 8
    return opt


def dict2str(opt, indent_l=1):
    """pretty-print a nested option dict for the log"""
    pad, out = ' ' * (2 * indent_l), []
    for key, val in opt.items():
        if isinstance(val, dict):
            out += [pad, key, ':[\n', dict2str(val, indent_l + 1), pad, ']\n']
        else:
            out += [pad, key, ': ', str(val), '\n']
    return ''.join(out)


class NoneDict(dict):
    """dict whose missing keys read as None"""

    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    if isinstance(opt, dict):
        return NoneDict((k, dict_to_nonedict(v)) for k, v in opt.items())
    return opt
