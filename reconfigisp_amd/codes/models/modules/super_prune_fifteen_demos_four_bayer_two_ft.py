"""SuperPruneFifteenDemosFourBayerTwoFt - the search super-net whose classical-operator proxies are fine-tuned
online against their real kernels (mirror of models/modules/super_prune_fifteen_demos_four_bayer_two_ft.py).
Identical forward to the plain super-net; adds ``proxy_ft_flag`` (which of the 15 sRGB entries are fine-tuned,
:103-118 - reinhard / filmic are flagged off there for a NaN problem, bm3d has no teacher) and
``load_proxy_nets`` (:194-209), which copies a fine-tuned proxy into the same entry of every sRGB step."""
from .super_prune_fifteen_demos_four_bayer_two import SuperPruneFifteenDemosFourBayerTwo

_FT_NAMES = ('gamma', 'reinhard', 'crysisengine', 'filmic', 'grayworld', 'whiteworld', 'bilateral', 'median', 'fastnlm',
             'skip', 'wbmanual', 'path_restore_14l_bgr', 'wbquadratic', 'gtmmanual', 'bm3d')
_FT_ENABLED = ('crysisengine', 'whiteworld', 'bilateral', 'median', 'fastnlm')


class SuperPruneFifteenDemosFourBayerTwoFt(SuperPruneFifteenDemosFourBayerTwo):
    def __init__(self, n_step, threshold, module_path):
        super().__init__(n_step, threshold, module_path)
        self.n_step = n_step
        self.proxy_ft_flag = [(name, int(name in _FT_ENABLED)) for name in _FT_NAMES]

    def load_proxy_nets(self, name_net_dict):
        """name -> fine-tuned proxy; its weights replace that entry in all n_step sRGB slots."""
        for idx, (name, enabled) in enumerate(self.proxy_ft_flag):
            if not enabled or name not in name_net_dict:
                continue
            state = name_net_dict[name].state_dict()
            for k in range(self.n_step):
                self.all_modules[-1 - k][idx].load_state_dict(state)
