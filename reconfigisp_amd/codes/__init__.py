"""Host-side mirror of the reference's ``codes/`` tree (registry, model wrappers, options, utils,
drivers).  ``install_aliases()`` additionally exposes it - and the plugin-shaped kernels - under the
reference's own top-level module names, so code written against the reference
(``from models import create_model``, ``import options.options as option``, ``import whitebalance``)
imports this implementation unchanged."""
import importlib
import sys

_TOP = ('models', 'options', 'utils', 'data')


def install_aliases():
    from .. import isp_kernels
    isp_kernels.install()
    for name in _TOP:
        module = importlib.import_module(__name__ + '.' + name)
        sys.modules.setdefault(name, module)
    # submodules that the reference imports by dotted path
    for dotted in ('models.networks', 'models.base_model', 'models.isp_model', 'models.darts_model',
                   'models.darts_ft_model', 'models.lr_scheduler', 'models.modules',
                   'models.modules.super_prune_fifteen_demos_four_bayer_two_ft', 'models.modules.tools_origin',
                   'models.modules.tools_proxy', 'models.modules.srcnn_res_arch',
                   'models.modules.srcnn_demosaic_arch', 'models.modules.path_14l_bayer_arch',
                   'models.modules.path_14l_bgr_arch', 'models.modules.isp_universal',
                   'models.modules.origin_universal', 'models.modules.super_prune_fifteen_demos_four_bayer_two',
                   'options.options', 'utils.util', 'utils.util_loss', 'utils.util_path_restore',
                   'data.data_sampler'):
        sys.modules.setdefault(dotted, importlib.import_module(__name__ + '.' + dotted))
