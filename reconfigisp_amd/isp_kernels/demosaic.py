"""demosaic.Demosaic - options 'nearestneighbor' (NCHW in/out, tools_origin.py:278-284),
'bilinear' / 'laplacian' (NHWC, 0..255 domain, :457-468, :491-502) and 'demosaicnet' (:302-308)."""
from .. import functional as F
from ._layout import to_nchw, to_nhwc


_DEMOSAICNET = None


def register_demosaicnet(fn):
    """Plug a user-supplied differentiable DemosaicNet: fn((N,1,H,W) RGGB in [0,1]) -> (N,3,H,W) BGR.
    The original lives only in the private ISP_Kernels package and is not reproducible."""
    global _DEMOSAICNET
    _DEMOSAICNET = fn


def demosaicnet_available():
    return _DEMOSAICNET is not None


class Demosaic:
    def run(self, img, option, params):
        fmt_in = params.get('input', {}).get('format', 'RGGB')
        if fmt_in != 'RGGB':
            raise ValueError('Demosaic: only the RGGB mosaic is supported, got %r' % (fmt_in,))
        if option == 'nearestneighbor':
            return F.demosaic_nearest(img)
        if option in ('bilinear', 'laplacian'):
            return to_nhwc(F.origin_demosaic(to_nchw(img), option))
        if option == 'demosaicnet':
            if _DEMOSAICNET is not None:
                return _DEMOSAICNET(img)
            raise NotImplementedError(
                "Demosaic 'demosaicnet': the Gharbi-style network and its weights live only in the private "
                'ISP_Kernels package and cannot be reproduced (SURVEY.md section 2); the registry slot is kept')
        raise ValueError('Demosaic: unknown option %r' % (option,))
