"""projn_nw = '1-layer' / '2-layer' (hfc_with_swav/swav_clustering.py:250-269) on HIP: the row kernels of
csrc/gx_projhead.hip against torch autograd, the reference's own pretrain() / predict trajectory (GV17), one optimizer
step against the oracle in every scope / projection mode, and 2 faked ranks = 1."""
import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden
from test_hip_swav import make_swav, mini_gen, rel

pytestmark = pytest.mark.gpu
T = torch.from_numpy

KINDS = ["1-layer", "2-layer"]


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops
    ops.lib()
    return torch.device("cuda")


# ------------------------------------------------------------------------------------- kernels
def test_gather_scatter_rows_are_adjoint(gpu):
    from ganecdotes_amd import ops
    torch.manual_seed(0)
    b, hw, d, n = 3, 12, 40, 50
    z = torch.randn(b, hw, hw, d, device="cuda")
    picks = torch.randint(0, hw * hw, (b, n), dtype=torch.int32, device="cuda")
    maps = torch.randint(-1, hw * hw, (b, hw * hw), dtype=torch.int32, device="cuda")
    for m in (None, maps):
        rows = ops.gather_rows(z, picks, m)
        src = picks.long() if m is None else torch.gather(m.long(), 1, picks.long())
        want = torch.where((src >= 0)[..., None], torch.gather(z.reshape(b, -1, d), 1, src.clamp(min=0)[..., None].expand(-1, -1, d)),
                           torch.zeros((), device="cuda"))
        assert torch.equal(rows, want.reshape(b * n, d))
        g = torch.randn(b * n, d, device="cuda")
        acc = ops.scatter_rows_add_(g, picks, m, torch.zeros_like(z))
        # <gather(z), g> == <z, scatter(g)>
        lhs, rhs = (rows.double() * g.double()).sum(), (z.double() * acc.double()).sum()
        assert abs(float(lhs - rhs)) < 1e-6 * max(1.0, abs(float(lhs)))      # fp32 atomics where picks repeat


def test_leaky_relu_rows(gpu):
    from ganecdotes_amd import ops
    torch.manual_seed(1)
    x = torch.randn(70, 24, device="cuda")
    x[3, 5] = 0.0
    y = ops.leaky_relu_(x.clone())
    assert torch.equal(y, torch.nn.functional.leaky_relu(x, 0.01))
    dy = torch.randn_like(x)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.leaky_relu(xr, 0.01).backward(dy)
    # from the output, or from any positive multiple of it (the step hands over the normalised rows)
    assert torch.equal(ops.leaky_relu_bwd_(dy.clone(), y * 3.0), xr.grad)


@pytest.mark.parametrize("act", [0, 1], ids=["lrelu", "tanh"])
@pytest.mark.parametrize("rows", [37, 5000])
def test_batchnorm_act_forward_backward_vs_autograd(gpu, act, rows):
    """BatchNorm1d(train) + LeakyReLU(0.01) / Tanh: output, batch statistics, d input, d gamma, d beta"""
    from ganecdotes_amd import ops
    torch.manual_seed(2)
    d = 48
    x = (torch.randn(rows, d, device="cuda") * 2.0 + 0.7)
    gamma, beta = torch.rand(d, device="cuda") + 0.5, torch.randn(d, device="cuda") * 0.3
    da = torch.randn(rows, d, device="cuda")
    xr, gr, br = (t.double().clone().requires_grad_(True) for t in (x, gamma, beta))
    pre = torch.nn.functional.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5)
    out = torch.tanh(pre) if act == 1 else torch.nn.functional.leaky_relu(pre, 0.01)
    out.backward(da.double())
    mean, var = ops.batch_stats(x)
    assert rel(mean, x.double().mean(0)) < 1e-6 and rel(var, x.double().var(0, unbiased=False)) < 1e-5
    a, xhat = ops.bn_act_fwd(x, mean, var, gamma, beta, 1e-5, act)
    assert rel(a, out.detach()) < 1e-5
    dx, dg, db = ops.bn_act_bwd(da, xhat, var, gamma, beta, 1e-5, act)
    assert rel(dx, xr.grad) < 2e-5 and rel(dg, gr.grad) < 2e-5 and rel(db, br.grad) < 2e-5


# ------------------------------------------------------------------------------------- GV17: the reference's own run
@pytest.mark.parametrize("kind", KINDS)
def test_pretrain_predict_golden_gv17(gpu, tmp_path, kind):
    d = load_golden("gv17_projection_heads")
    tag = kind[0] + "l"
    g = mini_gen(golden_sd(d), gpu)
    sw = make_swav(g, gpu, tmp_path, use_larc=False, transforms=False, cpu_reference_rng=True, projn_nw=kind)
    torch.manual_seed(42)
    np.random.seed(42)
    z = torch.randn(64, 16)                              # the reference constructor's mean_latent draw
    sw.mean_latent = g.style(z.cuda()).mean(0, keepdim=True)
    sw.pretrain(None, num_test_samples=2)
    np.testing.assert_allclose([float(l.item()) for l in sw.loss_history], d[f"{tag}_losses"], rtol=1e-3)
    sdp = sw.projection.state_dict()
    want = {k[len(tag) + 6:]: v for k, v in d.items() if k.startswith(f"{tag}_proj_") and "after" not in k}
    assert set(sdp) == set(want)                                          # the reference's pickled module, key for key
    for k, v in want.items():
        if "num_batches" in k:
            assert int(sdp[k]) == int(v)
        else:
            assert rel(sdp[k], T(v)) < 1e-3, k
    assert rel(sw.prototype.weight.data, T(d[f"{tag}_proto_weight"])) < 1e-3
    assert rel(sw.prototype.bias.data, T(d[f"{tag}_proto_bias"])) < 1e-3
    proj = torch.load(tmp_path / "projection.pt", weights_only=False)
    assert [type(m).__name__ for m in proj] == [type(m).__name__ for m in sw.projection]
    # predict from the CAPTURED modules (BatchNorm in train mode: batch statistics of the image's pixels)
    for k, v in want.items():
        sdp[k].copy_(T(v).to(sdp[k].dtype).cuda())
    for k in [k for k in d if k.startswith(f"{tag}_proj_after_pred_call_")]:
        sdp[k[len(tag) + 22:]].copy_(T(d[k]).cuda())
    sw.mean_latent = T(d[f"{tag}_mean_latent"]).cuda()
    preds, labels = sw.predict_swav_codes(T(d[f"{tag}_pred_w"]).cuda())
    assert rel(preds, T(d[f"{tag}_preds"])) < 5e-5
    assert (labels.cpu().numpy() == d[f"{tag}_labels"]).mean() > 0.995
    for k in [k for k in d if k.startswith(f"{tag}_proj_after_predict_")]:
        assert rel(sdp[k[len(tag) + 20:]], T(d[k])) < 1e-4, k


# ------------------------------------------------------------------------------------- one step vs the oracle
def _oracle_head(S, sw, kind):
    if kind == "1-layer":
        return S.HeadParams(kind)
    ex = [p.detach().cpu().clone() for p in sw._proj_head().extra_params()]
    return S.HeadParams(kind, *ex)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mode", ["pyramid", "panel"])
@pytest.mark.parametrize("scope,batch,nproto", [("image", 1, 12), ("image", 2, 12), ("global", 3, 12), ("global", 2, 200)])
def test_train_step_vs_oracle(gpu, tmp_path, kind, mode, scope, batch, nproto):
    """two optimizer steps (rotated / flipped views, LARC): loss, every gradient — the BatchNorm affines and the second
    Linear included — and the updated parameters; K = 12 would take the few-prototype routes with a linear projection and
    must not here, K = 200 runs the weight-plane score GEMM"""
    from oracle import swav_ref as S
    from ganecdotes_amd import ops
    from ganecdotes_amd.draws import draw_image
    sd = golden_sd(load_golden("gv10_pretrain"))
    g = mini_gen(sd, gpu)
    sw = make_swav(g, gpu, tmp_path, sinkhorn_scope=scope, batch_size=batch, projection_mode=mode, projn_nw=kind,
                   nprototypes=nproto, nclasses=16 if nproto > 16 else 8)
    torch.manual_seed(123)
    np.random.seed(123)
    ml = S.G.mean_latent(sd, 64, 16)
    sw.mean_latent = ml.cuda()
    sw.build_heads()
    if kind == "2-layer":            # off the (1, 0) start, so that d gamma / d beta are exercised away from it
        for p in sw._proj_head().extra_params():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.2)
    base = [p.detach().cpu().clone() for p in sw._params()[:3]]
    params = S.SwavParams(*base, _oracle_head(S, sw, kind))
    nleaf = len(sw._params())
    bufs = [None] * nleaf
    for step in range(2):
        images = [draw_image(16, 7, 3, 32 * 32, 2, transforms=True) for _ in range(batch)]
        loss = sw.train_step(images)
        oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip,
                                                 ops.rotflip_map(v.angle, v.flip, 32, 32, "cuda").cpu().long())
                                     for v in im.views], im.picks) for im in images]
        ref_loss, grads, bufs = S.swav_train_step(
            sd, ml, 0.7, oimgs, params, bufs, lr=0.01, momentum=0.9, trust=0.01, use_larc=True, hlen=56,
            patch_size=300, num_patches=2, temperature=0.1, eps=0.05, niters=10, source_pdf="uniform",
            sinkhorn_scope=scope)
        assert len(grads) == nleaf == (8 if kind == "2-layer" else 3)
        np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-3)
        for got, want in zip(sw._grads, grads):
            assert rel(got, want) < 2e-3
        for got, want in zip(sw._params(), [params.proj_w, params.proto_w, params.proto_b] + params.head.leaves()):
            assert rel(got, want) < 1e-4


def test_unknown_projection_network_is_refused(gpu, tmp_path):
    sd = golden_sd(load_golden("gv10_pretrain"))
    sw = make_swav(mini_gen(sd, gpu), gpu, tmp_path, projn_nw="3-layer")
    with pytest.raises(ValueError):
        sw.build_heads()


def test_add_local_loss_is_refused_like_the_reference(gpu, tmp_path):
    """swav_clustering.py:420-425 raises TypeError on `sum(int)` with every shipped layer_hf_dim: refused with the reason"""
    sd = golden_sd(load_golden("gv10_pretrain"))
    sw = make_swav(mini_gen(sd, gpu), gpu, tmp_path, add_local_loss=True)
    with pytest.raises(NotImplementedError, match="raises in the reference itself"):
        sw.pretrain(None, num_test_samples=0)
