"""THE dispatch table, pinned: which entry point of libreconfigisp_hip.so serves every convolution launch of the four learned-proxy
families (srcnn_res_arch.py:14-24, srcnn_demosaic_arch.py:12-24, path_14l_bayer_arch.py:59-88, path_14l_bgr_arch.py:58-86) in
{training forward, backward-data, inference} x {single, grouped}, under the default arithmetic (RISP_CONV_ARITH=f16x2,
RISP_CONV_TOEP_FIRST=train) and under RISP_CONV_ARITH=f32.  ``convnets.route`` / ``route_small`` are pure functions of the launch's
geometry; this literal table is the documentation DESIGN.md section 4.3 points at.  (That ``conv`` / ``conv_small`` really launch what
``route`` says is checked on the GPU: tests/test_gpu_conv_modes.py::test_conv_launches_what_route_says.)"""
import pytest

from reconfigisp_amd import convnets as CN

R, A, M, NB, CB, SH = CN.EPI_RELU, CN.EPI_ADD, CN.EPI_MASK, CN.EPI_NOBIAS, CN.EPI_CASEBIAS, CN.EPI_SHUFFLE2
PLAIN, UNSH = CN.LOAD_PLAIN, CN.LOAD_UNSHUFFLE2
H = W = 256            # the search's patch; a proxy at 128 x 128 after space-to-depth where noted
BIG, SMALL = 32, 1     # images of a launch: the batch of config 3 / one image (a training grid below TOEP_MIN_TILES)

# (family, layer, how it is launched) -> entry.  'conv' rows: (k, cin_layer, cout_layer, transpose, load, epilogue, add_c); 'small' rows:
# (k, cin, cout) of the SmallConv as launched.
CONV = {
    # ---- SRCNNRes (first layer with its 9 + P constant planes folded out: SrcnnResFold)
    ('srcnn_res', '9x9 3->64 first', 'fwd'): ((9, 3, 64, False, PLAIN, R | CB, 0), {'train': 'risp_conv2d_toep_first_exact', 'infer': 'risp_conv2d_toep_first', 'f32': 'risp_conv2d_k3'}),
    ('srcnn_res', '5x5 64->32', 'fwd'): ((5, 64, 32, False, PLAIN, R, 0), {'train': 'risp_conv2d_f16x2', 'infer': 'risp_conv2d_f16x2', 'f32': 'risp_conv2d_wino45'}),
    ('srcnn_res', '5x5 64->32', 'bwd'): ((5, 64, 32, True, PLAIN, M | NB, 0), {'train': 'risp_conv2d_f16x2', 'f32': 'risp_conv2d_wino45'}),
    # (round 6: 3 input channels as (filter row, channel) pairs in ONE matrix instruction's reduction index, split precision - it was the fp32 F(4,5) kernel on 4 padded channels)
    ('srcnn_res', '5x5 32->3 last', 'bwd'): ((5, 32, 3, True, PLAIN, M | NB, 0), {'train': 'risp_conv2d_thin5', 'f32': 'risp_conv2d_wino45'}),
    # the unfolded form (_SrcnnRes / _SrcnnResTrain: 12 + P input channels with constant planes)
    ('srcnn_res', '9x9 17->64 unfolded', 'fwd'): ((9, 17, 64, False, CN.LOAD_CONSTCH, R, 0), {'train': 'risp_conv2d', 'infer': 'risp_conv2d', 'f32': 'risp_conv2d'}),
    ('srcnn_res', '9x9 17->64 unfolded', 'bwd'): ((9, 17, 64, True, PLAIN, A | NB, 3), {'train': 'risp_conv2d', 'f32': 'risp_conv2d'}),
    # ---- SRCNNDemosaic
    ('srcnn_demosaic', '9x9 4->64 first (space-to-depth load)', 'fwd'): ((9, 4, 64, False, UNSH, R, 0), {'train': 'risp_conv2d_toep_first_exact', 'infer': 'risp_conv2d_toep_first', 'f32': 'risp_conv2d_k3'}),
    ('srcnn_demosaic', '1x1 64->32', 'fwd'): ((1, 64, 32, False, PLAIN, R, 0), {'train': 'risp_conv2d', 'infer': 'risp_conv2d', 'f32': 'risp_conv2d'}),
    ('srcnn_demosaic', '1x1 64->32', 'bwd'): ((1, 64, 32, True, PLAIN, M | NB, 0), {'train': 'risp_conv2d', 'f32': 'risp_conv2d'}),
    ('srcnn_demosaic', '5x5 32->12 last (through PixelShuffle)', 'bwd'): ((5, 32, 12, True, UNSH, M | NB, 0), {'train': 'risp_conv2d', 'f32': 'risp_conv2d'}),
    # ---- Path-Restore (Bayer: 4 space-to-depth planes; BGR: 3 channels)
    ('path14l_bayer', '3x3 4->64 first', 'fwd'): ((3, 4, 64, False, UNSH, R, 0), {'train': 'risp_conv2d_k3', 'infer': 'risp_conv2d_k3', 'f32': 'risp_conv2d_k3'}),
    ('path14l_bgr', '3x3 3->64 first', 'fwd'): ((3, 3, 64, False, PLAIN, R, 0), {'train': 'risp_conv2d_k3', 'infer': 'risp_conv2d_k3', 'f32': 'risp_conv2d_k3'}),
    ('path14l', '3x3 64->64 block conv 1', 'fwd'): ((3, 64, 64, False, PLAIN, R, 0), {'train': 'risp_conv2d_f16x2', 'infer': 'risp_conv2d_f16x2', 'f32': 'risp_conv2d_wino43'}),
    ('path14l', '3x3 64->64 block conv 2 (+ skip)', 'fwd'): ((3, 64, 64, False, PLAIN, R | A, 64), {'train': 'risp_conv2d_f16x2', 'infer': 'risp_conv2d_f16x2', 'f32': 'risp_conv2d_wino43'}),
    ('path14l', '3x3 64->64 block conv 2', 'bwd'): ((3, 64, 64, True, PLAIN, M | NB, 0), {'train': 'risp_conv2d_f16x2', 'f32': 'risp_conv2d_wino43'}),
    ('path14l', '3x3 64->64 block conv 1 (+ skip gradient)', 'bwd'): ((3, 64, 64, True, PLAIN, A | M | NB, 64), {'train': 'risp_conv2d_f16x2', 'f32': 'risp_conv2d_wino43'}),
    ('path14l_bayer', '3x3 64->4 last (through PixelShuffle)', 'bwd'): ((3, 64, 4, True, UNSH, M | NB, 0), {'train': 'risp_conv2d', 'f32': 'risp_conv2d'}),
    ('path14l_bgr', '3x3 64->3 last', 'bwd'): ((3, 64, 3, True, PLAIN, M | NB, 0), {'train': 'risp_conv2d_wino43', 'f32': 'risp_conv2d_wino43'}),
}
SMALLS = {
    # (k, cin, cout, has_mask): {(mode, images): entry}
    # (round 6: the 3-cout layers with whole chunks of 16 input channels run with the filter rows in the rows of the matrix instruction -
    # useful / issued products 0.84 for 9x9 64 -> 3 (27 of 32 rows, every reduction slot a channel) against 0.42 in the band form, 0.47
    # against 0.23 for 5x5 32 -> 3; a batch of ONE 256 x 256 image is 16 work items: below TAPOUT_MIN_ITEMS, the vector kernel)
    ('srcnn_res', '5x5 32->3 last', 'fwd'): ((5, 32, 3, False), {('train', BIG): 'risp_conv2d_tapout', ('train', SMALL): 'risp_conv2d_small', ('infer', SMALL): 'risp_conv2d_tapout', ('f32', BIG): 'risp_conv2d_small'}),
    ('srcnn_res', '9x9 64->3 first', 'bwd'): ((9, 64, 3, False), {('train', BIG): 'risp_conv2d_tapout', ('train', SMALL): 'risp_conv2d_small', ('train', 8): 'risp_conv2d_tapout', ('f32', BIG): 'risp_conv2d_small'}),
    ('srcnn_demosaic', '5x5 32->12 last + PixelShuffle', 'fwd'): ((5, 32, 12, False), {('train', BIG): 'risp_conv2d_toep', ('train', SMALL): 'risp_conv2d_small', ('infer', SMALL): 'risp_conv2d_toep', ('f32', BIG): 'risp_conv2d_small'}),
    ('srcnn_demosaic', '9x9 64->4 first (through PixelShuffle)', 'bwd'): ((9, 64, 4, False), {('train', BIG): 'risp_conv2d_toep', ('train', SMALL): 'risp_conv2d_small', ('f32', BIG): 'risp_conv2d_small'}),
    # (round 6: Path-Restore's 3x3 tails of INFERENCE launches with (filter row, cout) pairs in the rows of the matrix instruction - one scale per
    # wave and row, so any grid and any batch give the same bits; training launches keep the vector kernel)
    ('path14l_bayer', '3x3 64->4 last + PixelShuffle', 'fwd'): ((3, 64, 4, False), {('train', BIG): 'risp_conv2d_small', ('infer', BIG): 'risp_conv2d_narrow3', ('infer', SMALL): 'risp_conv2d_narrow3', ('f32', BIG): 'risp_conv2d_small'}),
    ('path14l_bgr', '3x3 64->3 last', 'fwd'): ((3, 64, 3, False), {('train', BIG): 'risp_conv2d_small', ('infer', SMALL): 'risp_conv2d_narrow3', ('f32', BIG): 'risp_conv2d_small'}),
    ('path14l', '3x3 64->4 / 3 first', 'bwd'): ((3, 64, 4, False), {('train', BIG): 'risp_conv2d_small', ('f32', BIG): 'risp_conv2d_small'}),
}


def _mode(monkeypatch, mode):
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32' if mode == 'f32' else 'f16x2')
    monkeypatch.setattr(CN, 'TOEP_FIRST', 'train')
    return mode == 'infer'


@pytest.mark.parametrize('key', sorted(CONV), ids=lambda k: ' / '.join(k))
def test_route_table(key, monkeypatch):
    (k, cin_l, cout_l, transpose, load, epi, add_c), want = CONV[key]
    cin, cout = (cout_l, cin_l) if transpose else (cin_l, cout_l)
    have = CN.pack_kinds(k, cin_l, cout_l, transpose)
    for mode, entry in want.items():
        infer = _mode(monkeypatch, mode)
        got = CN.route(k, cin, cout, H, W, transpose, load, epi, add_c, infer, True, have)
        assert got == entry, (key, mode, got)
        # a grouped launch (the 8 SRCNNRes / 2 SRCNNDemosaic members of a slot stacked along N) takes the same route: the group sits
        # in the grid, not in the dispatch (3x3 layers are never grouped: conv() refuses)
        # ... and so does an unaligned view or a width that is no multiple of 4, except that everything falls back to risp_conv2d
        assert CN.route(k, cin, cout, H, W + 2, transpose, load, epi, add_c, infer, True, have) == 'risp_conv2d'
        assert CN.route(k, cin, cout, H, W, transpose, load, epi, add_c, infer, False, have) == 'risp_conv2d'


@pytest.mark.parametrize('key', sorted(SMALLS), ids=lambda k: ' / '.join(k))
def test_route_small_table(key, monkeypatch):
    (k, cin, cout, has_mask), want = SMALLS[key]
    for (mode, images), entry in want.items():
        infer = _mode(monkeypatch, mode)
        got = CN.route_small(k, cin, cout, H, W, images, infer, has_mask, CN.small_has_toep(k, cout), None, CN.small_has_tapout(k, cin, cout),
                             CN.small_has_narrow3(k, cin, cout))
        assert got == entry, (key, mode, images, got)


def test_useful_over_issued_products_of_the_few_channel_kernels():
    """what a matrix instruction of each form carries: rows used x reduction slots used"""
    band = lambda k, cout: (k / 16.0) * (cout / 4.0)                 # risp_conv2d_toep: 16-slot window, rows = 4 couts x 8 positions
    taprow = lambda k, cout: (k * cout) / 32.0                      # risp_conv2d_tapout: rows = (cout, ky), every slot a channel
    assert abs(band(9, 3) - 0.42) < 0.005 and abs(taprow(9, 3) - 0.84) < 0.005
    assert abs(band(5, 3) - 0.23) < 0.005 and abs(taprow(5, 3) - 0.47) < 0.005
    # the 9x9 3 -> 64 first layer: (channel, tap) reduction index (27 of 32 slots) against 16 window slots per (channel, filter row)
    useful = 3 * 2.0 * 81 * 3 * 64
    assert CN.first_layer_form(3, 64, 256, 256) == 'xwin' and abs(useful / CN.first_layer_issued(3, 64, 256, 256) - 0.84) < 0.005
    assert CN.first_layer_form(3, 64, 256, 256, True) == 'band' and abs(useful / CN.first_layer_issued(3, 64, 256, 256, True) - 0.5625) < 1e-9
    assert CN.first_layer_form(3, 64, 3000, 4000) == 'band' and CN.first_layer_form(4, 64, 256, 256) == 'band'
    assert abs(3 * 2.0 * 81 * 3 * 32 / CN.first_layer_issued(3, 32, 256, 256) - 0.42) < 0.005      # 32 couts: half the rows carry zeros
    # 5x5 3 -> 32 backward-data: (filter row, channel) pairs in one instruction's 16 reduction slots, 5 instructions x 3 products per pixel
    assert abs(3 * 2.0 * 25 * 3 * 32 / (3 * 2.0 * 5 * 16 * 32) - 0.9375) < 1e-9
    assert 'thin5' in CN.pack_kinds(5, 32, 3, True) and 'thin5' not in CN.pack_kinds(5, 32, 3, False) and 'thin5' in CN.pack_kinds(5, 3, 64, False)
    assert 'thin5' not in CN.pack_kinds(5, 32, 4, True) and 'thin5' not in CN.pack_kinds(3, 32, 3, True)
    # Path-Restore's 3x3 64 -> 3 / 4 tails: rows (filter row, cout) 9 / 12 of 32 - an HBM-bound layer, the matrix pipe is idle either way
    assert CN.small_has_narrow3(3, 64, 3) and CN.small_has_narrow3(3, 16, 4) and not CN.small_has_narrow3(3, 64, 5) and not CN.small_has_narrow3(3, 80, 3)
    assert not CN.small_has_narrow3(3, 24, 3) and not CN.small_has_narrow3(5, 64, 3)
    # the layers that stay on the band form: 4 couts (36 rows do not fit 32), 12 couts, channel counts that are no multiple of 16
    assert not CN.small_has_tapout(9, 64, 4) and not CN.small_has_tapout(5, 32, 12) and not CN.small_has_tapout(5, 7, 1)
    assert CN.small_has_tapout(9, 64, 3) and CN.small_has_tapout(5, 32, 3) and CN.small_has_tapout(5, 16, 1)


def test_first_layer_switch_and_addressing(monkeypatch):
    first = (9, 3, 64, H, W, False, PLAIN, R | CB, 0)
    have = CN.pack_kinds(9, 3, 64)
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f16x2')
    for setting, (train, infer) in {'train': ('risp_conv2d_toep_first_exact', 'risp_conv2d_toep_first'), 'plain': ('risp_conv2d_toep_first',) * 2,
                                    'infer': ('risp_conv2d_k3', 'risp_conv2d_toep_first'), '0': ('risp_conv2d_k3',) * 2}.items():
        monkeypatch.setattr(CN, 'TOEP_FIRST', setting)
        assert CN.route(*first, False, True, have) == train and CN.route(*first, True, True, have) == infer, setting
    # a 64-channel 3x3 layer on an untiled 3000 x 4000 frame is past the 2^31-byte buffers of the split-precision kernel
    assert CN.route(3, 64, 64, 3000, 4000, have=CN.pack_kinds(3, 64, 64)) == 'risp_conv2d_wino43'
    assert CN.route(3, 64, 64, 512, 512, have=CN.pack_kinds(3, 64, 64)) == 'risp_conv2d_f16x2'
    # a residual narrower than the layer (SRCNNRes' 3-channel skip) is not the split-precision kernel's epilogue
    assert CN.route(5, 64, 32, H, W, epi=A, add_c=3, have=CN.pack_kinds(5, 64, 32)) == 'risp_conv2d_wino45'
