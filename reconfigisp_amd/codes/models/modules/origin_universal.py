"""OriginUniversal - fixed ISP that uses the classical kernels where they exist; only BM3D and the
two Path-Restore nets are learned (mirror of models/modules/origin_universal.py:9-165).

The shipped reference puts an *instance* ``GtmManual(4)`` into its class pool (:61) and crashes
with TypeError when sRGB index 14 is selected; the documented behaviour (a 4-segment tone curve)
is what is built here."""
from . import registry as R
from .isp_universal import _FixedPipeline


class OriginUniversal(_FixedPipeline):
    srgb_names = R.NAMES_SRGB
    use_origin_kernels = True

    def __init__(self, module_path, architecture):
        super().__init__()
        self._build(module_path, architecture)
