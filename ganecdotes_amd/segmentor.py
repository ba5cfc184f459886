"""One-shot fine-tune head (`segmentor` of the plugin surface).

Same constructor `(in_ch, n_class, size)` and layer plan as the reference's OneShotSegmentor
(hfc_with_swav/swav_clustering.py:697-758): a stack of dilated 3x3 convs + LeakyReLU(0.2)
('XXS'..'L') or a per-pixel linear layer ('Lin').  This is the torch module the plugin surface exports
and that `projection.pt`-style pickles / `OneShotHead.to_module()` hand back; the fine-tune and the test
loop themselves run on HIP (ganecdotes_amd/oneshot.py `OneShotHead`, csrc/gx_head.hip — same plan, same
`layers.{2i}.weight/bias` state_dict keys; 'Lin' is a single 1-tap layer there).
"""
from torch import nn

_PLANS = {      # size -> (dilations, hidden widths)
    'XXS': ([1], [12]),
    'XS': ([1, 2, 1], [16, 8]),
    'S': ([1, 2, 1, 2, 1], [128, 64, 64, 32]),
    'M': ([1, 2, 4, 1, 2, 4, 1], [128, 64, 64, 64, 64, 32]),
    'L': ([1, 2, 4, 8, 1, 2, 4, 8, 1], [128, 64, 64, 64, 64, 64, 64, 32]),
}


class OneShotSegmentor(nn.Module):
    def __init__(self, in_ch, n_class, size='S'):
        super().__init__()
        if size != 'Lin' and size not in _PLANS:
            raise AssertionError(f"size must be one of {sorted(_PLANS) + ['Lin']}")
        self.channels, self.size = n_class, size
        if size == 'Lin':
            self.layers = nn.Sequential(nn.Linear(in_ch, n_class), nn.LeakyReLU(0.2, inplace=True))
            return
        dil, hidden = _PLANS[size]
        widths = [in_ch] + hidden + [n_class]
        mods = []
        for d, a, b in zip(dil, widths[:-1], widths[1:]):     # zip stops at len(dil) like the reference
            mods += [nn.Conv2d(a, b, kernel_size=3, padding=d, dilation=d), nn.LeakyReLU(0.2, inplace=True)]
        self.layers = nn.Sequential(*mods[:-1])

    def forward(self, x):
        if self.size != 'Lin':
            return self.layers(x)
        b, c, h, w = x.shape
        y = self.layers(x.view(b, c, -1).permute(0, 2, 1))
        return y.permute(0, 2, 1).reshape(b, self.channels, h, w)
