"""spatialnoisereduction.SpatialNoiseReduction - options 'bilateral', 'median', 'fastnlm'
(0..255 domain, non-differentiable; call sites tools_origin.py:686-710, :734-751, :775-797)."""
from .. import functional as F
from ._layout import to_nchw, to_nhwc


class SpatialNoiseReduction:
    def run(self, img, option, params):
        if option not in ('bilateral', 'median', 'fastnlm'):
            raise ValueError('SpatialNoiseReduction: unknown option %r' % (option,))
        return to_nhwc(F.origin_denoise(to_nchw(img), option, params))
