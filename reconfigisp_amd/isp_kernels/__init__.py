"""Plugin-shaped modules replacing the absent private ``/DATA/ISP_Kernels`` package
(imported at the reference's models/modules/tools_origin.py:8-17).

Five modules, each with one no-argument class exposing
``run(img, option: str, params: dict) -> Tensor`` (B1 boundary).
``install()`` registers them under the reference's top-level module names
(``whitebalance``, ``gamma``, ``demosaic``, ``globaltonemapping``, ``spatialnoisereduction``)
so an unmodified ``tools_origin.py`` binds to the HIP kernels.
"""
import sys

from . import demosaic, gamma, globaltonemapping, spatialnoisereduction, whitebalance

_NAMES = ('whitebalance', 'gamma', 'demosaic', 'globaltonemapping', 'spatialnoisereduction')


def install():
    for name in _NAMES:
        sys.modules.setdefault(name, globals()[name])
