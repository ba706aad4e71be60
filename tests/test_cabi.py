"""The C-ABI library loads on a CPU-only machine and exports every symbol include/risp.h declares
(no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'risp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(risp_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    from reconfigisp_amd import lib
    names = declared_symbols()
    assert len(names) >= 30
    handle = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), 'libreconfigisp_hip.so does not export %s' % n
    assert sorted(lib.SIGNATURES) == names, 'lib.SIGNATURES and include/risp.h disagree: %s' % (
        sorted(set(lib.SIGNATURES) ^ set(names)),)
    loaded = lib.load()
    assert loaded.risp_version() == 100
    assert loaded.risp_last_error() == b''


def test_conv_desc_matches_header():
    from reconfigisp_amd import lib
    text = open(os.path.join(ROOT, 'include', 'risp.h')).read()
    body = text[text.index('typedef struct {'): text.index('} risp_conv_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.replace('typedef struct {', '').strip()
        if not decl:
            continue
        names = re.sub(r'^(const\s+)?(float|int|long long)\s*', '', decl)
        fields += [n.strip().lstrip('*') for n in names.split(',')]
    assert [f for f, _ in lib.ConvDesc._fields_] == fields


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    from reconfigisp_amd import lib
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', '/nonexistent/libreconfigisp_hip.so')
    with pytest.raises(RuntimeError, match='no\\s+CPU fallback'):
        lib.load()


def test_cpu_tensor_is_rejected_not_emulated():
    import pytest
    import torch
    import reconfigisp_amd.functional as F
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.gamma(torch.rand(1, 3, 4, 4), torch.rand(1, 1))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.demosaic_nearest(torch.rand(1, 1, 4, 4))


def _struct_fields(name):
    """field names of `typedef struct <name> { ... } <name>;` in include/risp.h, in declaration order"""
    text = open(os.path.join(ROOT, 'include', 'risp.h')).read()
    body = text[text.index('typedef struct %s {' % name) + len('typedef struct %s {' % name): text.index('} %s;' % name)]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r'^(const\s+)?(float|int|long long|unsigned char)\s*', '', decl)
        for part in decl.split(','):
            fields.append(re.sub(r'\[.*?\]', '', part).replace('*', '').strip())
    return fields


def test_training_and_mixture_descriptors_match_header():
    from reconfigisp_amd import lib
    assert [f.rstrip('_') for f, _ in lib.TrainDesc._fields_] == _struct_fields('risp_train_desc')
    assert [f for f, _ in lib.ParamBlocksDesc._fields_] == _struct_fields('risp_param_blocks_desc')
    assert [f for f, _ in lib.SrcnnGroupDesc._fields_] == _struct_fields('risp_srcnn_group_desc')
    assert [f for f, _ in lib.SlotMixDesc._fields_] == _struct_fields('risp_slot_mix_desc')
    assert [f for f, _ in lib.ListDesc._fields_] == _struct_fields('risp_list_desc')
    assert lib.LIST_MAX == 64 and '#define RISP_MAX_LIST 64' in open(os.path.join(ROOT, 'include', 'risp.h')).read()
    assert lib.MIX_MAX == 16 and '#define RISP_MAX_MIX 16' in open(os.path.join(ROOT, 'include', 'risp.h')).read()
    assert lib.TRAIN_MAX == 6 and lib.PARAM_OPS_MAX == 16           # RISP_MAX_TRAIN_CHAIN / RISP_MAX_PARAM_OPS
    assert lib.GROUP_MAX == 16 and '#define RISP_MAX_GROUP 16' in open(os.path.join(ROOT, 'include', 'risp.h')).read()
    assert (lib.GROUP_SHARED_X, lib.GROUP_SHARED_ADD) == (1, 2)
    text = open(os.path.join(ROOT, 'include', 'risp.h')).read()
    assert '#define RISP_MAX_TRAIN_CHAIN 6' in text and '#define RISP_MAX_PARAM_OPS 16' in text
