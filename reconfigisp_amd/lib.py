"""ctypes binding of ``libreconfigisp_hip.so`` (C ABI declared in ``include/risp.h``).

There is no CPU or PyTorch fallback behind this module: if the shared library is
missing or fails to load, importing an operator raises ``RuntimeError`` telling the
user to build it (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C reconfigisp_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RISP_HIP_LIBRARY: another BUILD of the same library (the diagnostic / ablation builds of tools/*.sh live in /tmp, the in-tree file is
# never rebuilt under a measurement); absent, the in-tree library
LIB_PATH = os.environ.get('RISP_HIP_LIBRARY') or os.path.join(_HERE, 'lib', 'libreconfigisp_hip.so')

_f = C.c_void_p        # device float*
_i = C.c_int
_z = C.c_size_t
_fl = C.c_float
_s = C.c_void_p        # hipStream_t
_pp = C.POINTER(C.c_void_p)


class ConvDesc(C.Structure):
    """mirror of risp_conv_desc"""
    _fields_ = [('N', _i), ('H', _i), ('W', _i), ('cin', _i), ('cout', _i), ('ksize', _i),
                ('load_mode', _i), ('cin_img', _i), ('epilogue', _i), ('add_c', _i),
                ('x', _f), ('wpack', _f), ('bias', _f), ('cvals', _f), ('add', _f), ('mask', _f),
                ('y', _f),
                ('group_n', _i), ('group_flags', _i), ('wpack_gs', C.c_longlong), ('bias_gs', C.c_longlong)]


GROUP_SHARED_X, GROUP_SHARED_ADD = 1, 2
GROUP_MAX = 16


class SrcnnGroupDesc(C.Structure):
    """mirror of risp_srcnn_group_desc"""
    _fields_ = [('G', _i), ('N', _i), ('HW', _i), ('M', _i), ('P', _i * GROUP_MAX), ('pv', _f * GROUP_MAX),
                ('rcase', _f * GROUP_MAX), ('wconst', _f * GROUP_MAX)]


TRAIN_MAX = 6


class TrainDesc(C.Structure):
    """mirror of risp_train_desc"""
    _fields_ = [('in_', _f), ('gt', _f), ('y', _f), ('from_bayer', _i), ('n_ops', _i), ('ops', _i * TRAIN_MAX),
                ('blocks', _f * TRAIN_MAX), ('raw', _f * TRAIN_MAX), ('grad', _f * TRAIN_MAX),
                ('exp_avg', _f * TRAIN_MAX), ('exp_avg_sq', _f * TRAIN_MAX), ('loss_kind', _i),
                ('N', _i), ('H', _i), ('W', _i), ('lr_step', _fl), ('beta1', _fl), ('beta2', _fl),
                ('one_minus_beta1', _fl), ('one_minus_beta2', _fl), ('bias2_sqrt', _fl), ('eps', _fl), ('loss', _f), ('scratch', _f)]


PARAM_OPS_MAX = 16


class ParamBlocksDesc(C.Structure):
    """mirror of risp_param_blocks_desc"""
    _fields_ = [('n_ops', _i), ('N', _i), ('width', _i * PARAM_OPS_MAX), ('raw', _f * PARAM_OPS_MAX),
                ('block', _f * PARAM_OPS_MAX), ('gblock', _f * PARAM_OPS_MAX), ('graw', _f * PARAM_OPS_MAX),
                ('gstride', _i * PARAM_OPS_MAX)]


MIX_MAX = 16
SLOT_TENSOR = -1


class SlotMixDesc(C.Structure):
    """mirror of risp_slot_mix_desc"""
    _fields_ = [('K', _i), ('N', _i), ('HW', _i), ('kind', _i * MIX_MAX), ('w', _fl * MIX_MAX), ('pmul', _fl * MIX_MAX),
                ('ptr', _f * MIX_MAX), ('go', _f * MIX_MAX), ('gp', _f * MIX_MAX), ('x', _f), ('y', _f)]


LIST_MAX = 64


class ListDesc(C.Structure):
    """mirror of risp_list_desc"""
    _fields_ = [('n', _i), ('numel', _i * LIST_MAX), ('a', _f * LIST_MAX), ('b', _f * LIST_MAX), ('c', _f * LIST_MAX), ('e', _f * LIST_MAX)]


def _pw(n_extra=0):
    # forward: (x, p, y, N, HW, stream); backward (n_extra = 3): (x, p, gy, gx, gp, scratch, N, HW, stream)
    return [_f] * (3 + n_extra) + [_i, _i, _s]


# name -> (restype, argtypes); every symbol include/risp.h declares
SIGNATURES = {
    'risp_version': (_i, []),
    'risp_last_error': (C.c_char_p, []),
    'risp_demosaic_nearest_fwd': (_i, [_f, _f, _i, _i, _i, _s]),
    'risp_demosaic_nearest_bwd': (_i, [_f, _f, _i, _i, _i, _s]),
    'risp_wb_manual_fwd': (_i, _pw()), 'risp_wb_manual_bwd': (_i, _pw(3)),
    'risp_gamma_fwd': (_i, _pw()), 'risp_gamma_bwd': (_i, _pw(3)),
    'risp_gtm_manual_fwd': (_i, _pw()), 'risp_gtm_manual_bwd': (_i, _pw(3)),
    'risp_wb_quadratic_fwd': (_i, _pw()), 'risp_wb_quadratic_bwd': (_i, _pw(3)),
    'risp_gain3_fwd': (_i, _pw()), 'risp_gain3_bwd': (_i, _pw(3)),
    'risp_channel_stats_scratch_floats': (_z, [_i, _i]),
    'risp_channel_stats': (_i, [_f, _f, _f, _f, _i, _i, _s]),
    'risp_stats_bwd': (_i, [_f, _f, _f, _f, _f, _i, _i, _s]),
    'risp_stats_bwd_rows': (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _s]),
    'risp_histc': (_i, [_f, _f, _i, _i, _i, _s]),
    'risp_srcnn_cvals': (_i, [_f, _f, _f, _i, _i, _i, _s]),
    'risp_srcnn_case_table': (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _s]),
    'risp_grayworld_gains_fwd': (_i, [_f, _f, _i, _i, _s]),
    'risp_grayworld_gains_bwd': (_i, [_f, _f, _f, _i, _i, _s]),
    'risp_chain_fwd': (_i, [_f, _i, C.POINTER(_i), _pp, _pp, _i, _i, _i, _s]),
    'risp_mix_fwd': (_i, [_pp, C.POINTER(C.c_float), _i, _f, _z, _s]),
    'risp_mix_scratch_floats': (_z, []),
    'risp_mix_bwd': (_i, [_pp, C.POINTER(C.c_float), _i, _f, _pp, _f, _f, _z, _s]),
    'risp_slot_mix_fwd': (_i, [C.POINTER(SlotMixDesc), _s]),
    'risp_slot_mix_scratch_floats': (_z, [_i, _i]),
    'risp_slot_mix_bwd': (_i, [C.POINTER(SlotMixDesc), _f, _f, _f, _f, _s]),
    'risp_param_grad_scratch_floats': (_z, [_i]),
    'risp_conv_wpack_floats': (_z, [_i, _i, _i]),
    'risp_conv_pack_weights': (_i, [_f, _i, _i, _i, _i, _f, _s]),
    'risp_conv2d': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_small_cout_pad': (_i, [_i]),
    'risp_conv_small_wpack_floats': (_z, [_i, _i, _i]),
    'risp_conv2d_small': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_small_groups': (_i, [C.POINTER(ConvDesc)]),
    'risp_conv2d_small_split': (_i, [C.POINTER(ConvDesc), _f, _i, _s]),
    'risp_conv_k3_cout_block': (_i, [_i, _i]),
    'risp_conv_k3_wpack_floats': (_z, [_i, _i, _i]),
    'risp_conv2d_k3': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_rect_sums': (_i, [_f, _f, _i, _i, _i, _i, _s]),
    'risp_rect_sums_tiles': (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _s]),
    'risp_srcnn_const_grad': (_i, [_f, _f, _f, _i, _i, _i, _s]),
    'risp_srcnn_case_table_group': (_i, [_f, C.POINTER(SrcnnGroupDesc), _f, _s]),
    'risp_srcnn_const_grad_group': (_i, [_f, C.POINTER(SrcnnGroupDesc), _f, _i, _s]),
    'risp_group_sum': (_i, [_f, _f, _i, _i, _i, _i, _f, _i, _f, _s]),
    'risp_conv_wino43_chunk': (_i, []),
    'risp_conv_wino43_wpack_floats': (_z, [_i, _i]),
    'risp_conv2d_wino43': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_wino45_chunk': (_i, []),
    'risp_conv_wino45_layout': (_i, []),
    'risp_conv_wino45_wpack_floats': (_z, [_i, _i]),
    'risp_conv2d_wino45': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_f16x2_wpack_bytes': (_z, [_i, _i, _i]),
    'risp_conv2d_f16x2': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv2d_f16x2_uniform': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_toep_wpack_bytes': (_z, [_i, _i, _i]),
    'risp_conv2d_toep': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_toep_tiles': (_i, [_i, _i]),
    'risp_conv2d_toep_sums': (_i, [C.POINTER(ConvDesc), _f, _s]),
    'risp_conv_tapout_wpack_bytes': (_z, [_i, _i]),
    'risp_conv_tapout_seg_rows': (_i, [_i, _i, _i]),
    'risp_conv2d_tapout': (_i, [C.POINTER(ConvDesc), _i, _s]),
    'risp_conv_tapout_items': (_i, [_i, _i, _i, _i]),
    'risp_conv2d_tapout_sums': (_i, [C.POINTER(ConvDesc), _i, _f, _s]),
    'risp_conv_toep_first_wpack_bytes': (_z, [_i, _i]),
    'risp_conv2d_toep_first': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_narrow3_wpack_bytes': (_z, [_i]),
    'risp_conv2d_narrow3': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv_thin5_wpack_bytes': (_z, [_i]),
    'risp_conv2d_thin5': (_i, [C.POINTER(ConvDesc), _s]),
    'risp_conv2d_toep_first_exact': (_i, [C.POINTER(ConvDesc), _f, C.c_longlong, C.c_void_p, C.c_uint, _s]),
    'risp_conv_wgrad_scratch_floats': (_z, [_i, _i, _i]),
    'risp_conv2d_wgrad': (_i, [C.POINTER(ConvDesc), _f, _f, _f, _z, _s]),
    'risp_plane_sums': (_i, [_f, _f, _i, _i, _i, _i, _i, _s]),
    'risp_cond_fc_row_floats': (_i, [C.POINTER(C.c_int), _i]),
    'risp_cond_fc_fwd': (_i, [_f, _f, C.POINTER(C.c_int), _i, _f, _f, _i, _s]),
    'risp_cond_fc_bwd': (_i, [_f, C.POINTER(C.c_int), _i, _f, _f, _f, _f, _f, _i, _i, _s]),
    'risp_tile_gather': (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _s]),
    'risp_tile_blend': (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    'risp_origin_demosaic': (_i, [_f, _f, _i, _i, _i, _i, _fl, _fl, _s]),
    'risp_origin_bilateral': (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _fl, _fl, _s]),
    'risp_origin_median': (_i, [_f, _f, _i, _i, _i, _i, _fl, _fl, _s]),
    'risp_origin_fastnlm': (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _fl, _s]),
    'risp_origin_tonemap_scratch_floats': (C.c_size_t, [_i]),
    'risp_origin_tonemap': (_i, [_f, _f, _i, _f, _f, _f, _f, _i, _i, _fl, _fl, _s]),
    'risp_bilateral_chain_fwd': (_i, [_f, _i, _f, _f, _f, _f, _f, _i, _i, C.POINTER(_i), _pp, _pp, _i, _i, _i, _s]),
    'risp_raw_crop': (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _fl, _s]),
    'risp_gt_crop': (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _s]),
    'risp_resize_rggb': (_i, [_f, _f, _i, _i, _i, _i, _i, _i, _s]),
    'risp_bilateral_chain_kernel': (C.c_char_p, [_i, _i, _i]),
    'risp_sse_uint8_doubles': (_z, []),
    'risp_sse_uint8': (_i, [_f, _f, _f, _z, _z, _s]),
    'risp_prune_softmax_fwd': (_i, [_f, _f, _fl, _i, _f, _f, _f, _s]),
    'risp_prune_softmax_bwd': (_i, [_f, _f, _f, _i, _f, _s]),
    'risp_param_blocks_fwd': (_i, [C.POINTER(ParamBlocksDesc), _s]),
    'risp_param_blocks_bwd': (_i, [C.POINTER(ParamBlocksDesc), _s]),
    'risp_loss_scratch_floats': (_z, []),
    'risp_local_global_scratch_floats': (_z, [_i, _i]),
    'risp_local_global_l2': (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _s]),
    'risp_pixel_loss': (_i, [_f, _f, _f, _f, _f, _z, _i, _s]),
    'risp_darts_virtual_step': (_i, [C.POINTER(ListDesc), _fl, _fl, _s]),
    'risp_list_norm_eps': (_i, [C.POINTER(ListDesc), _f, _s]),
    'risp_list_norm_eps_part': (_i, [C.POINTER(ListDesc), _f, _i, _i, _s]),
    'risp_list_axpy_scalar': (_i, [C.POINTER(ListDesc), _f, _fl, _s]),
    'risp_darts_alpha_grad': (_i, [C.POINTER(ListDesc), _f, _fl, _f, _s]),
    'risp_sgd_momentum_step': (_i, [C.POINTER(ListDesc), _fl, _fl, _i, _s]),
    'risp_adam_step': (_i, [C.POINTER(ListDesc), _fl, _fl, _fl, _fl, _fl, _fl, _s]),
    'risp_train_scratch_floats': (_z, [_i]),
    'risp_chain_train_step': (_i, [C.POINTER(TrainDesc), _s]),
}

_lib = None


def load():
    """Load the library once; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'reconfigisp_amd: %s is missing - the HIP extension has not been built. There is no '
            'CPU fallback. Build it with `make -C reconfigisp_amd/csrc` (needs hipcc, gfx950).' % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. no HIP runtime on this machine
        raise RuntimeError('reconfigisp_amd: cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI and this table diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_FN = {}                # entry points by name (one attribute lookup on the CDLL per name, not per call)
CALLS = None            # diagnostics (bench.py, tools/): set to {} to count the C-ABI calls by entry point


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError with risp_last_error() on failure."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    if CALLS is not None:
        CALLS[name] = CALLS.get(name, 0) + 1
    status = fn(*args)
    if status != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, status, load().risp_last_error().decode()))


def ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*[C.c_void_p(p) if p else None for p in ptrs])
