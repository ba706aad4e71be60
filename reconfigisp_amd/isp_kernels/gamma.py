"""gamma.Gamma - option 'manual' (call sites tools_origin.py:59-69, :189-194)."""
from .. import functional as F
from ._layout import to_nchw, to_nhwc


class Gamma:
    def run(self, img, option, params):
        if option != 'manual':
            raise ValueError('Gamma: unknown option %r' % (option,))
        return to_nhwc(F.gamma(to_nchw(img), params['gamma']))
