"""whitebalance.WhiteBalance - options 'grayworld', 'manual', 'whiteworld'.

Call sites in the reference: tools_origin.py:33-41 (grayworld), :211-221 and :245-250 (manual),
:647-662 (whiteworld).  Arithmetic is the build-defined OPSPEC (parity unpinned)."""
from .. import functional as F
from ._layout import to_nchw, to_nhwc


class WhiteBalance:
    def run(self, img, option, params):
        x = to_nchw(img)
        if option == 'manual':
            y = F.wb_manual(x, params['gain'])     # gain (N,3) in [0,5], BGR order
        elif option == 'grayworld':
            y = F.grayworld(x)
        elif option == 'whiteworld':
            y = F.origin_whiteworld(x, params['white_point_ratio'])
        else:
            raise ValueError('WhiteBalance: unknown option %r' % (option,))
        return to_nhwc(y)
