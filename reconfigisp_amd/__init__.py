"""reconfigisp_amd - MI355X-native implementation of ReconfigISP's per-image ISP forward path.

``reconfigisp_amd.functional``  differentiable operators over the C ABI (include/risp.h)
``reconfigisp_amd.isp_kernels`` the plugin-shaped modules tools_origin.py imports (B1 boundary)
``reconfigisp_amd.codes``       host-side mirror of the reference's registry / model surface (B2)
"""
__version__ = '0.1.0'
