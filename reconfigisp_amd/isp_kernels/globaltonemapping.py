"""globaltonemapping.GlobalToneMapping - options 'reinhard', 'crysisengine', 'filmic'
(0..255 domain, non-differentiable; call sites tools_origin.py:526-543, :566-581, :604-623)."""
from .. import functional as F
from ._layout import to_nchw, to_nhwc


class GlobalToneMapping:
    def run(self, img, option, params):
        if option not in ('reinhard', 'crysisengine', 'filmic'):
            raise ValueError('GlobalToneMapping: unknown option %r' % (option,))
        return to_nhwc(F.origin_tonemap(to_nchw(img), option, params))
