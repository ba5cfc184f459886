"""The finest pyramid level of the K <= 16 path without its projected table (csrc/gx_fineq.hip, pyramid.py):
the quadratic-form route against the route that forms Z (itself pinned by the oracle / golden tests), kernel by kernel
and through Pyramid.project_tables / backward_lowrank, and the shifted tables of a pyramid whose top levels have no taps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops as o
    o.lib()
    prev = o.set_gemm_mode(1)
    yield o
    o.set_gemm_mode(prev)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _taps(b, res_top, chans, seed=5, scale=1.0):
    """NHWC taps of a pyramid: chans = {res: [c, ...]}; values grow towards the fine levels like generator activations"""
    g = torch.Generator().manual_seed(seed)
    taps = []
    for r in sorted(chans):
        for c in chans[r]:
            taps.append((torch.randn(b, r, r, c, generator=g) * scale * (1.0 + 0.5 * np.log2(r))).cuda())
    return taps


@pytest.mark.parametrize("h2", [True, False], ids=["f16x3", "bf16x6"])
@pytest.mark.parametrize("k,two_taps", [(8, True), (16, True), (5, False)])
def test_fineq_tables_equal_the_projected_table_route(ops, k, two_taps, h2, monkeypatch):
    """rn / score tables of every pixel: gx_fineq_fwd (CF-wide forms of the finest taps + three tables of the level below)
    against the tables computed from the projected table Z (gx_upsample2_add_scores), and against fp64 — with the weight-plane
    contractions (the level GEMMs and the quadratic-form kernel) on their f16x3 forms and on bf16x6."""
    from ganecdotes_amd.pyramid import Pyramid
    monkeypatch.setattr(ops, "GEMM_H2", h2)
    b, d = 3, 512
    chans = {16: [64], 32: [128, 128], 64: [128, 128] if two_taps else [256]}
    taps = _taps(b, 64, chans)
    hlen = sum(sum(v) for v in chans.values())
    torch.manual_seed(11)
    wp = (torch.randn(d, hlen) / hlen ** 0.5).cuda()
    wc = torch.nn.functional.normalize(torch.randn(k, d), dim=1).cuda()
    bias = (0.1 * torch.randn(k)).cuda()
    pyr = Pyramid(ops.TapSet(taps), hlen)
    rn_q, sc_q, sh_q = pyr.project_tables(wp, wc, bias, quad=True)
    assert pyr._lr["mode"] == "quad" and sh_q == 0
    fine = [t for t in taps if t.shape[1] == 64]
    if h2:      # the taps' maximum as the forward pass met them: the bound of the backward pass's f16x3 form
        assert float(pyr._lr["t_amax"]) == max(float(t.abs().max()) for t in fine)
    else:
        assert pyr._lr["t_amax"] is None and pyr._lr["z_amax"] is None
    rn_t, sc_t, sh_t = Pyramid(ops.TapSet(taps), hlen).project_tables(wp, wc, bias, quad=False)
    assert sh_t == 0
    # fp64 truth
    z = None
    off = 0
    for r in sorted(chans):
        lv = [t for t in taps if t.shape[1] == r]
        x = torch.cat([t.reshape(-1, t.shape[3]) for t in lv], 1).double()
        c = x.shape[1]
        p = (x @ wp[:, off:off + c].double().t()).reshape(b, r, r, d)
        off += c
        z = p if z is None else p + z.repeat_interleave(2, 1).repeat_interleave(2, 2)
    z = z.reshape(-1, d)
    nrm = z.norm(dim=1)
    sc64 = (z @ wc.double().t()) / nrm[:, None] + bias.double()
    assert rel(rn_q, 1.0 / nrm) < 2e-5 and rel(rn_t, 1.0 / nrm) < 2e-5
    assert rel(sc_q, sc64) < 2e-5 and rel(sc_t, sc64) < 2e-5
    assert rel(rn_q, rn_t) < 2e-5 and rel(sc_q, sc_t) < 2e-5


@pytest.mark.parametrize("k,two_taps", [(8, True), (16, True), (5, False)])
def test_fineq_backward_equals_the_projected_table_route(ops, k, two_taps):
    """d loss / d Wp and d loss / d Wc from a random coefficient grid: the Gram-form backward (gx_fineq_bwd,
    GEMMs on the level below) against the dense finish on the projected table + the pyramid adjoint."""
    from ganecdotes_amd.pyramid import Pyramid
    b, d = 3, 512
    chans = {16: [64], 32: [128, 128], 64: [128, 128] if two_taps else [256]}
    taps = _taps(b, 64, chans, seed=9)
    hlen = sum(sum(v) for v in chans.values())
    torch.manual_seed(13)
    wp = (torch.randn(d, hlen) / hlen ** 0.5).cuda()
    wc = torch.nn.functional.normalize(torch.randn(k, d), dim=1).cuda()
    bias = torch.zeros(k).cuda()
    ls = ops.lib().gx_lowrank_stride(k)
    coeffs = torch.zeros(b * 64 * 64, ls)
    hit = torch.rand(b * 64 * 64) < 0.4                       # never-sampled pixels have all-zero coefficients
    coeffs[hit, :k + 1] = torch.randn(int(hit.sum()), k + 1) * 1e-3
    coeffs = coeffs.cuda()
    outs = []
    for quad in (True, False):
        pyr = Pyramid(ops.TapSet(taps), hlen)
        pyr.project_tables(wp, wc, bias, quad=quad)
        assert pyr._lr["mode"] == ("quad" if quad else "table")
        gwp = torch.full_like(wp, 7.0)                         # overwritten, not accumulated
        gwc = torch.full_like(wc, 7.0)
        seen = []
        pyr.backward_lowrank(coeffs, wp, wc, gwp, gwc, on_level=lambda cols: seen.append(tuple(cols)),
                             on_heads=lambda: seen.append("heads"))
        assert seen[0] == "heads" and len(seen) == 4           # heads first, then one call per level, finest first
        outs.append((gwp, gwc, seen))
    assert outs[0][2] == outs[1][2]
    assert rel(outs[0][0], outs[1][0]) < 2e-5
    assert rel(outs[0][1], outs[1][1]) < 2e-5


@pytest.mark.parametrize("k,two_taps,res", [(8, True, 64), (16, True, 32), (5, False, 16)])
def test_fineq_bwd_one_pass_vs_fp64(ops, k, two_taps, res):
    """gx_fineq_bwd: the weighted Gram matrix S (upper blocks contracted, lower mirrored), P, the pooled e t and the pooled
    coefficients from one pass over the taps — against fp64"""
    torch.manual_seed(29)
    b = 3
    rows = b * res * res
    t1 = torch.randn(rows, 128 if two_taps else 256).cuda()
    t2 = torch.randn(rows, 128).cuda() if two_taps else None
    ls = ops.lib().gx_lowrank_stride(k)
    c = torch.zeros(rows, ls)
    c[:, :k + 1] = torch.randn(rows, k + 1) * 1e-2
    c = c.cuda()
    s, pm, tt, cpool, tt_amax = ops.fineq_bwd(c, t1, t2, k, res, want_amax=True)
    assert float(tt_amax) == float(tt.abs().max())      # the bound the f16x3 GEMMs on Tt read from the device
    t = (t1 if t2 is None else torch.cat([t1, t2], 1)).double()
    e = c[:, k].double()
    assert rel(s, (t * e[:, None]).t() @ t) < 3e-6
    assert rel(s, s.t()) < 1e-6                     # mirrored across the diagonal blocks; inside one, (e t)^T t both ways
    assert rel(pm, c[:, :k].double().t() @ t) < 3e-6
    pool = lambda x: x.reshape(b, res // 2, 2, res // 2, 2, -1).sum((2, 4)).reshape(rows // 4, -1)
    assert rel(tt, pool(t * e[:, None])) < 2e-6 and rel(cpool, pool(c.double())) < 2e-6
    # the f16x3 form (images scaled from max |t| and the coefficient grid's two maxima, all on the device): fp32-class as well —
    # also with gradient-sized coefficients whose prototype part and e part differ by three orders of magnitude
    cm = ops.lowrank_coeff_amax(c, k)
    assert float(cm[0]) == float(c[:, :k].abs().max()) and float(cm[1]) == float(c[:, k].abs().max())
    t_amax = ops.absmax(torch.cat([t1.reshape(-1), t2.reshape(-1)]) if t2 is not None else t1.reshape(-1))
    c2 = c * 1e-4
    c2[:, k] *= 1e-3
    for cc in (c, c2):
        sh, pmh, tth, cph, tta = ops.fineq_bwd(cc, t1, t2, k, res, want_amax=True, t_amax=t_amax)
        eh = cc[:, k].double()
        err6 = [rel(x, y) for x, y in zip(ops.fineq_bwd(cc, t1, t2, k, res)[:2], ((t * eh[:, None]).t() @ t, cc[:, :k].double().t() @ t))]
        assert rel(sh, (t * eh[:, None]).t() @ t) < max(3e-6, 1.5 * err6[0]) and rel(sh, sh.t()) < 1e-6
        assert rel(pmh, cc[:, :k].double().t() @ t) < max(3e-6, 1.5 * err6[1])
        assert rel(tth, pool(t * eh[:, None])) < 2e-6 and rel(cph, pool(cc.double())) < 2e-6
        assert float(tta) == float(tth.abs().max())
    assert not torch.equal(sh, ops.fineq_bwd(c2, t1, t2, k, res)[0])             # (the other form did run)


def test_tables_shifted_below_levels_without_taps(ops):
    """hlen cuts the finest level off (a 512^2 generator at hlen = 5376): the tables and the coefficient grid live one
    level down and the pick lists index them through the shift — same scores, same gradients as with the upsampled table."""
    from ganecdotes_amd.pyramid import Pyramid
    b, d, k = 2, 256, 8
    chans = {8: [32], 16: [64, 64], 32: [16, 16]}
    taps = _taps(b, 32, chans, seed=21)
    hlen = 32 + 128                                             # the 32^2 taps are cut off
    torch.manual_seed(17)
    wp = (torch.randn(d, hlen) / hlen ** 0.5).cuda()
    wc = torch.nn.functional.normalize(torch.randn(k, d), dim=1).cuda()
    bias = (0.1 * torch.randn(k)).cuda()
    pyr = Pyramid(ops.TapSet(taps), hlen)
    rn, sc, shift = pyr.project_tables(wp, wc, bias)
    assert shift == 1 and pyr.table_pixels() == 16 * 16 and rn.numel() == b * 256
    z = pyr.project(wp)                                         # the full-resolution table (what predict uses)
    rn_f, sc_f = ops.proto_fwd_lowrank(z, None, None, wc, bias)
    up = lambda t, c: t.reshape(b, 16, 16, c).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(-1, c)
    assert rel(up(rn, 1).flatten(), rn_f) < 1e-6 and rel(up(sc, k), sc_f) < 1e-6
    # gather through the shift == gather from the upsampled tables (B = 1 image per view, 2 patches)
    n, hw = 300, 32 * 32
    picks = torch.stack([torch.randperm(hw)[:n] for _ in range(2)]).reshape(2, 1, n).int().cuda()
    maps = torch.stack([torch.randperm(hw), torch.arange(hw)]).int().cuda()
    maps[0, :50] = -1                                           # rotated out of the frame
    out_s = torch.empty(2, 2, n, k).cuda()
    out_f = torch.empty(2, 2, n, k).cuda()
    rn_s = ops.gather_scores_batch(sc, rn, bias, picks, maps, 1, hw, out_s, shift=1)
    rn_g = ops.gather_scores_batch(sc_f, rn_f, bias, picks, maps, 1, hw, out_f)
    assert rel(out_s, out_f) < 1e-6 and rel(rn_s, rn_g) < 1e-6
    # scatter through the shift == 2x2 pooling of the full-resolution coefficient grid
    ds = (torch.randn(2, 2, n, k) * 1e-3).cuda()
    c_s = ops.lowrank_coeffs(2, 256, k, "cuda")
    c_f = ops.lowrank_coeffs(2, hw, k, "cuda")
    gb_s, gb_f = torch.zeros(k).cuda(), torch.zeros(k).cuda()
    ops.proto_bwd_lowrank_batch_(ds, out_s, rn_s, bias, picks, maps, c_s, 1, hw, gb_s, 0.0, shift=1)
    ops.proto_bwd_lowrank_batch_(ds, out_f, rn_g, bias, picks, maps, c_f, 1, hw, gb_f, 0.0)
    ls = c_s.shape[1]
    pooled = c_f.reshape(2, 16, 2, 16, 2, ls).sum((2, 4)).reshape(-1, ls)
    assert rel(c_s, pooled) < 1e-5 and rel(gb_s, gb_f) < 1e-6


@pytest.mark.parametrize("k,res,pool", [(8, 64, True), (16, 128, True), (5, 64, False), (1, 64, True), (12, 256, True)])
def test_fused_finish_gemm_vs_fp64_and_the_two_kernel_route(ops, k, res, pool):
    """gx_gemm_f32_nt_planes_finish (the dense finish in the GEMM's epilogue, rows in 2 x 64-pixel tiles): G, its 2x2 sums
    and gWc against fp64, and against gx_gemm_f32_nt_planes + gx_proto_lowrank_finish(add_to) which it replaces"""
    b, d, cf = (2, 512, 256) if res < 256 else (1, 256, 128)       # res 256: another D, CF and a single image
    rows = b * res * res
    torch.manual_seed(31 + k)
    tt = torch.randn(rows, cf).cuda()
    wf = (torch.randn(d, cf) / 16).cuda()
    z = torch.randn(b, res, res, d).cuda()
    wc = torch.nn.functional.normalize(torch.randn(k, d), dim=1).cuda()
    ls = ops.lib().gx_lowrank_stride(k)
    coeffs = torch.zeros(rows, ls)
    hit = torch.rand(rows) < 0.5
    coeffs[hit, :k + 1] = torch.randn(int(hit.sum()), k + 1)
    coeffs = coeffs.cuda()
    assert ops.gemm_finish_ok(d, cf, res, k)
    planes = ops.split_gemm_weights(wf)
    gwc = torch.full((k, d), 3.0).cuda()
    g, gp = ops.gemm_nt_planes_finish(tt, planes, coeffs, z, wc, gwc, 0.0, res, pool)
    # fp64
    c64, z64 = coeffs.double(), z.double().reshape(rows, d)
    want = tt.double() @ wf.double().t() + c64[:, :k] @ wc.double() + c64[:, k:k + 1] * z64
    assert rel(g.reshape(rows, d), want) < 2e-6
    assert rel(gwc, c64[:, :k].t() @ z64) < 2e-6
    if pool:
        wp_ = want.reshape(b, res // 2, 2, res // 2, 2, d).sum((2, 4))
        assert rel(gp, wp_) < 2e-6
    else:
        assert gp is None
    # the route it replaces
    g2 = torch.empty(rows, d).cuda()
    ops.gemm_nt_planes(tt, None, planes, g2)
    gwc2 = torch.full((k, d), 3.0).cuda()
    out = ops.proto_lowrank_finish(coeffs, z, wc, gwc2, 0.0, want_pool=pool, add_to=g2.reshape(z.shape))
    g2f, gp2 = out if pool else (out, None)
    assert rel(g, g2f) < 2e-6 and rel(gwc, gwc2) < 2e-6
    if pool:
        assert rel(gp, gp2) < 2e-6
    # the f16x3 form (two fp16 planes of Wf, Tt scaled from its maximum on the device): fp32-class too; also on gradient-sized Tt
    if ops.GEMM_H2:
        for scale in (1.0, 3e-7):
            tts = tt * scale
            ph = ops.split_gemm_weights(wf, h2=True)
            assert ph.h2
            with pytest.raises(RuntimeError):
                ops.gemm_nt_planes_finish(tts, ph, coeffs, z, wc, gwc2.clone(), 0.0, res, pool)        # no bound given
            gwh = torch.full((k, d), 3.0).cuda()
            gh, gph, gmax = ops.gemm_nt_planes_finish(tts, ph, coeffs * scale, z, wc, gwh, 0.0, res, pool, want_amax=True,
                                                      a_amax=ops.absmax(tts))
            assert rel(gh.reshape(rows, d), want * scale) < 2e-6 and rel(gwh, c64[:, :k].t() @ z64 * scale) < 2e-6
            assert float(gmax) == float(gh.abs().max())
            if pool:
                assert rel(gph, wp_ * scale) < 2e-6
        assert not torch.equal(gh, g * scale)
    # accumulate into gWc (beta = 1), and idempotence (no atomics)
    g3, gp3 = ops.gemm_nt_planes_finish(tt, planes, coeffs, z, wc, gwc2, 1.0, res, pool)
    assert rel(gwc2, 2.0 * gwc) < 2e-6
    assert torch.equal(g3, g) and (not pool or torch.equal(gp3, gp))


def test_backward_lowrank_fused_finish_equals_two_kernels(ops):
    """Pyramid.backward_lowrank with the level below the finest at 64^2 (the fused finish applies) against the same call
    with the finish as a pass of its own"""
    from ganecdotes_amd.pyramid import Pyramid
    b, d, k = 2, 512, 8
    chans = {32: [64], 64: [128, 128], 128: [128, 128]}
    taps = _taps(b, 128, chans, seed=11)
    hlen = sum(sum(v) for v in chans.values())
    torch.manual_seed(17)
    wp = (torch.randn(d, hlen) / hlen ** 0.5).cuda()
    wc = torch.nn.functional.normalize(torch.randn(k, d), dim=1).cuda()
    bias = torch.zeros(k).cuda()
    ls = ops.lib().gx_lowrank_stride(k)
    coeffs = torch.zeros(b * 128 * 128, ls)
    hit = torch.rand(b * 128 * 128) < 0.4
    coeffs[hit, :k + 1] = torch.randn(int(hit.sum()), k + 1) * 1e-3
    coeffs = coeffs.cuda()
    outs = []
    for fused in (True, False):
        pyr = Pyramid(ops.TapSet(taps), hlen)
        pyr.project_tables(wp, wc, bias)
        assert pyr._lr["mode"] == "quad"
        gwp, gwc = torch.full_like(wp, 7.0), torch.full_like(wc, 7.0)
        pyr.backward_lowrank(coeffs, wp, wc, gwp, gwc, fused_finish=fused)
        outs.append((gwp, gwc))
    assert rel(outs[0][0], outs[1][0]) < 5e-6 and rel(outs[0][1], outs[1][1]) < 5e-6
