"""define_G(opt) - which_model_G -> pipeline graph (mirror of models/networks.py:10-48).

The reference hard-codes ``module_path = '/DATA/module/'``; that stays the default, and the
optional key ``network_G.module_path`` overrides it (``~``/None = no weight files: proxies keep
their random initialisation - used by the synthetic benchmark and the tests)."""
import logging

logger = logging.getLogger('base')
DEFAULT_MODULE_PATH = '/DATA/module/'


def define_G(opt):
    opt_net = opt['network_G']
    module_path = opt_net['module_path'] if 'module_path' in opt_net else DEFAULT_MODULE_PATH
    which = opt_net['which_model_G']

    if which == 'SuperPruneFifteenDemosFourBayerTwo':
        from .modules.super_prune_fifteen_demos_four_bayer_two import SuperPruneFifteenDemosFourBayerTwo
        opt_net['n_modules']  # read (and required) by the reference, unused there too (networks.py:23)
        return SuperPruneFifteenDemosFourBayerTwo(n_step=opt_net['n_step'], threshold=opt_net['prune_threshold'],
                                                  module_path=module_path)
    if which == 'SuperPruneFifteenDemosFourBayerTwoFt':
        from .modules.super_prune_fifteen_demos_four_bayer_two_ft import SuperPruneFifteenDemosFourBayerTwoFt
        opt_net['n_modules']
        return SuperPruneFifteenDemosFourBayerTwoFt(n_step=opt_net['n_step'], threshold=opt_net['prune_threshold'],
                                                    module_path=module_path)
    if which == 'IspUniversal':
        from .modules.isp_universal import IspUniversal
        cond = opt_net['conditional_modules'] if 'conditional_modules' in opt_net else {}
        return IspUniversal(module_path=module_path, indiv_module_paths=opt_net['individual_module_paths'],
                            architecture=opt_net['architecture'], **(cond or {}))
    if which == 'OriginUniversal':
        from .modules.origin_universal import OriginUniversal
        return OriginUniversal(module_path=module_path, architecture=opt_net['architecture'])
    raise NotImplementedError('Generator model [{:s}] not recognized'.format(which))
