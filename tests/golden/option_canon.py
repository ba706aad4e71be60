"""Canonical form of a parsed option tree, shared by make_golden.py::gold_options (the IMPORTED reference's parse) and
tests/test_drivers_cpu.py (the mirror's parse): the fields options/options.py:8-62 DERIVES, spelled out, plus a hash of the whole
tree with the install root folded to '<root>' - data about the nine shipped YAMLs, not their text."""
import hashlib
import json


def _fold(v, root):
    if isinstance(v, dict):
        return {str(k): _fold(x, root) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_fold(x, root) for x in v]
    if isinstance(v, str) and v.startswith(root):
        return '<root>' + v[len(root):]
    return v


def canonical(opt):
    root = opt['path']['root']
    tree = _fold(opt, root)
    derived = {
        'is_train': tree['is_train'],
        'meta_device': tree['meta_device'],
        'datasets': {k: {f: d.get(f) for f in ('phase', 'data_type', 'mode')} for k, d in tree['datasets'].items()},
        'path': tree['path'],
        'val_freq': (tree.get('train') or {}).get('val_freq'),
        'save_checkpoint_freq': (tree.get('logger') or {}).get('save_checkpoint_freq'),
        'model': tree['model'],
        'which_model_G': tree['network_G']['which_model_G'],
        'top_level_keys': list(tree.keys()),
    }
    text = json.dumps(tree, sort_keys=True, default=str)
    return {'derived': derived, 'tree_sha256': hashlib.sha256(text.encode()).hexdigest()}
