"""Option-YAML parsing - mirror of options/options.py:8-93 (same keys and derived fields), so the
reference's nine shipped option files parse unchanged.  Host-side only."""
import os
import os.path as osp
from collections import OrderedDict

import yaml

from ..utils.util import OrderedYaml

Loader, Dumper = OrderedYaml()


def parse(opt_path, is_train=True):
    with open(opt_path, mode='r') as f:
        opt = yaml.load(f, Loader=Loader)

    # device selection: every machine except the reference authors' cluster exports gpu_ids
    # (HIP honours CUDA_VISIBLE_DEVICES through PyTorch-ROCm as well)
    if opt['machine'] != 'st_sh34':
        gpu_list = ','.join(str(x) for x in opt['gpu_ids'])
        os.environ['CUDA_VISIBLE_DEVICES'] = gpu_list
        print('export CUDA_VISIBLE_DEVICES=' + gpu_list)
    opt['is_train'] = is_train

    for phase, dataset in opt['datasets'].items():
        dataset['phase'] = phase.split('_')[0]
        dataset['data_type'] = 'lmdb' if str(dataset.get('dataroot') or '').endswith('lmdb') else 'img'
        if dataset['mode'].endswith('mc'):                # memcached-backed variant of a dataset
            dataset['data_type'] = 'mc'
            dataset['mode'] = dataset['mode'].replace('_mc', '')

    opt['meta_device'] = 'Meta' in opt['network_G']['which_model_G']

    for key, path in opt['path'].items():
        if path and key != 'strict_load':
            opt['path'][key] = osp.expanduser(path)
    root = osp.abspath(osp.join(__file__, osp.pardir, osp.pardir, osp.pardir))
    opt['path']['root'] = root
    if is_train:
        exp_root = osp.join(root, 'experiments', opt['name'])
        opt['path'].update(experiments_root=exp_root, models=osp.join(exp_root, 'models'),
                           training_state=osp.join(exp_root, 'training_state'), log=exp_root,
                           val_images=osp.join(exp_root, 'val_images'))
        if 'debug' in opt['name']:
            opt['train']['val_freq'] = 8
            opt['logger']['save_checkpoint_freq'] = 8
    else:
        results_root = osp.join(root, 'results', opt['name'])
        opt['path'].update(results_root=results_root, log=results_root)
    return opt


def dict2str(opt, indent_l=1):
    """nested dict -> indented text for the log"""
    pad = ' ' * (indent_l * 2)
    lines = []
    for k, v in opt.items():
        if isinstance(v, dict):
            lines.append(pad + k + ':[\n' + dict2str(v, indent_l + 1) + pad + ']\n')
        else:
            lines.append(pad + k + ': ' + str(v) + '\n')
    return ''.join(lines)


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt
