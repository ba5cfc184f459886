"""transposed conv alone and blur alone for the four up-sampling layers at the bench batch"""
import math, sys, torch
sys.path.insert(0, ".")
from ganecdotes_amd import ops, _lib
from ganecdotes_amd.ops import _p, _stream, lib, workspace
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
dev = torch.device("cuda"); B = 32
kf = torch.tensor([[1., 3, 3, 1]], device=dev); kf = (kf.t() @ kf) / 16.0
L = lib()
for h, cin, cout in ((16, 512, 512), (32, 512, 512), (64, 512, 256), (128, 256, 128)):
    x = torch.randn(B, h, h, cin, device=dev); wt = torch.randn(9, cout, cin, device=dev) / math.sqrt(9 * cin)
    s = torch.rand(B, cin, device=dev) + 0.5; dm = torch.rand(B, cout, device=dev) + 0.5
    wtp = ops.split_conv_weights(wt); xp = ops.modsplit(x, s)
    scratch = workspace("upconv", L.gx_modconv_up_scratch_bytes(B, h, h, cout), dev)
    out = torch.empty(B, 2 * h, 2 * h, cout, device=dev)
    noise = torch.randn(1, 1, 2 * h, 2 * h, device=dev); nw = torch.tensor([0.1], device=dev); bias = torch.randn(cout, device=dev)
    sn = torch.rand(B, cout, device=dev) + 0.5
    npl = ops.xplanes_buffer(B, 2 * h, 2 * h, cout, dev, "lab")
    def convt():
        _lib.check(L.gx_modconv_up_convt_xp(_p(xp), B, h, h, cin, _p(wt), _p(wtp), cout, _p(scratch), _stream()), "convt")
    def blur():
        _lib.check(L.gx_blur_act_fwd_xp(_p(scratch), B, 2 * h + 1, 2 * h + 1, cout, _p(kf), _p(dm), _p(noise), 0, _p(nw), _p(bias), 1,
                                        _p(out), _p(sn), _p(npl), _stream()), "blur")
    tc = timeit(convt)
    tbs = []
    for dma in (1, 0, 1, 0):                       # the blur pass with its T rows by LDS-DMA (round 5) / through registers, alternating
        L.gx_set_blur_dma(dma)
        tbs.append(timeit(blur, n=20, warm=3))
    L.gx_set_blur_dma(1)
    gf = 2.0 * B * h * h * cin * cout * 9 / 1e9
    el = B * 4 * h * h * cout
    print(f"H={h:3d} {cin}->{cout}: convt {tc:.3f} ms {gf / tc:.0f} TF/s | blur dma/reg/dma/reg {tbs[0]:.3f} / {tbs[1]:.3f} / {tbs[2]:.3f} / {tbs[3]:.3f} ms "
          f"= {el * 14 / tbs[2] / 1e9:.2f} / {el * 14 / tbs[3] / 1e9:.2f} TB/s (14 B/elem)", flush=True)
