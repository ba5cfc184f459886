"""N>1 on real HIP tensors: two ranks (sharing the one GPU of the test box, gloo collectives) run
the product's sharded train step; the result must equal the 1-rank step on the concatenated batch
(SURVEY.md par. 8e: W-rank result == 1-rank result)."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden

pytestmark = pytest.mark.gpu
MINI = {4: 16, 8: 16, 16: 8, 32: 8, 64: 8}


class Cfg:
    num_latents_for_mean = 64
    truncation = 0.7
    latent_dim = 16


def _args(scope, pdf, nproto=12, projn="linear"):
    swav = dict(num_epochs=1, num_samples=1, num_patches=2, sampling_method="random", patch_size=300,
                hf_interp="nearest", warmup_epochs=1, start_warmup=0.01, use_scheduler=False, base_lr=0.01,
                final_lr=0.0001, trust_coeff=0.01, train_args=dict(lr=0.01, momentum=0.9), projn_nw=projn,
                temperature=0.1, nprototypes=nproto, nclasses=8 if nproto <= 16 else 16,   # D = 16: weight-plane GEMMs
                hlen=56, add_local_loss=False, plot_test_images=False,
                epoch_print_freq=1, max_masks=4, sinkhorn_scope=scope)
    if projn == "linear+table":         # many prototypes with the per-pixel d(scores) tables under the sharded protocol
        swav.update(projn_nw="linear", ds_table=True)
    return dict(perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None, perturb_std=[1.0] * 3),
                swav_args=swav, sinkhorn_args=dict(source_pdf=pdf, niters=10, eps=0.05))


def _build(sd, group, scope, pdf, nproto=12, projn="linear"):
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.swav_clustering import SwAVClustering
    g = Generator(32, 16, 2, channels=MINI)
    g.load_state_dict(sd, strict=False)
    g = g.eval().cuda()
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", dist_group=group, **_args(scope, pdf, nproto, projn))
    torch.manual_seed(77)
    sw.mean_latent = g.style(torch.randn(64, 16).cuda()).mean(0, keepdim=True)
    torch.manual_seed(78)
    sw.build_heads()
    return sw


def _images(n):
    from ganecdotes_amd.draws import draw_image
    torch.manual_seed(79)
    np.random.seed(79)
    return [draw_image(16, 7, 3, 32 * 32, 2) for _ in range(n)]


def _worker(rank, world, port, sd, scope, pdf, out, nproto=12, projn="linear"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sw = _build(sd, dist.group.WORLD, scope, pdf, nproto, projn)
    imgs = _images(4)
    loss = None
    for _ in range(2):
        # the ranks run the pipelined form (views on their own stream, ViewMixin.prefetch_views), the 1-rank reference the plain one
        loss = sw.train_step(sw.prefetch_views(imgs[rank * 2:(rank + 1) * 2]))
    out[rank] = (float(loss.item()), [p.cpu().numpy() for p in sw._params()])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scope,pdf,nproto,projn", [
    ("global", "uniform", 12, "linear"), ("image", "uniform", 12, "linear"), ("global", "image", 12, "linear"),
    ("global", "uniform", 200, "linear"), ("image", "uniform", 200, "linear"),
    ("global", "uniform", 12, "1-layer"), ("global", "uniform", 200, "2-layer"), ("image", "uniform", 12, "2-layer"),
    ("global", "uniform", 200, "linear+table"), ("image", "uniform", 200, "linear+table"), ("global", "image", 200, "linear+table"),
    # a prototype count that is not a multiple of 4: padded columns and explicit marginals under the sharded protocol (round 5)
    ("global", "uniform", 203, "linear+table"), ("global", "image", 203, "linear+table")])
def test_two_ranks_equal_one_rank(scope, pdf, nproto, projn):
    """nproto = 200: the many-prototype path (score GEMM on weight planes with the Sinkhorn column maxima from its
    epilogue, one-pass CE with the bias gradient) under the sharded protocol; projn '2-layer' with the global scope:
    BatchNorm statistics and backward sums all-reduced (a BatchNorm batch = the rows of one Sinkhorn problem)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    sd = golden_sd(load_golden("gv10_pretrain"))
    sw = _build(sd, None, scope, pdf, nproto, projn)
    imgs = _images(4)
    for _ in range(2):
        loss1 = sw.train_step(imgs)
    ref = (float(loss1.item()), [p.cpu().numpy() for p in sw._params()])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, sd, scope, pdf, out, nproto, projn), nprocs=2, join=True)
    for r in range(2):
        np.testing.assert_allclose(out[r][0], ref[0], rtol=2e-4)
        for a, b in zip(out[r][1], ref[1]):
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-6 * np.abs(b).max())
    for a, b in zip(out[0][1], out[1][1]):          # replicas stay identical
        np.testing.assert_array_equal(a, b)


# ------------------------------------------------------------------------------------- RCCL itself
def _nccl_worker(rank, port, sd, scope, pdf, out):
    """ONE rank on the `nccl` backend (= RCCL): every exchange step of the sharded protocol runs through
    RCCL as an identity, asynchronous gradient messages included."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    calls = {"n": 0}
    real = dist.all_reduce

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    dist.all_reduce = counted
    sw = _build(sd, dist.group.WORLD, scope, pdf)
    imgs = _images(4)
    loss = None
    for _ in range(2):
        loss = sw.train_step(imgs)
    torch.cuda.synchronize()
    out[0] = (float(loss.item()), [p.cpu().numpy() for p in sw._params()], calls["n"], dist.get_backend())
    dist.all_reduce = real
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scope,pdf", [("global", "uniform"), ("global", "image")])
def test_nccl_backend_single_rank_equals_plain_step(scope, pdf):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    sd = golden_sd(load_golden("gv10_pretrain"))
    sw = _build(sd, None, scope, pdf)
    imgs = _images(4)
    for _ in range(2):
        loss1 = sw.train_step(imgs)
    ref = (float(loss1.item()), [p.cpu().numpy() for p in sw._params()])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_nccl_worker, args=(port, sd, scope, pdf, out), nprocs=1, join=True)
    loss, params, ncalls, backend = out[0]
    assert backend == "nccl"
    # per step: Sinkhorn 1 max + 10 sums (all patches and both views travel together), the {dWc, db, loss}
    # message and one message per pyramid level that owns columns of dWp (hlen = 56 ends inside the 16^2 taps:
    # 3 levels) — plus, with the 'image' pdf, the marginal histograms
    assert ncalls >= 2 * (11 + 1 + 3), ncalls
    np.testing.assert_allclose(loss, ref[0], rtol=2e-4)
    for a, b in zip(params, ref[1]):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-6 * np.abs(b).max())


# ------------------------------------------------------------------------------------- pretrain() under N ranks
def _pretrain_worker(rank, world, port, sd, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(42)                      # what lib/util/util.py does at import on EVERY rank
    np.random.seed(42)
    sw = _build(sd, dist.group.WORLD, "global", "uniform")
    sw.swav_args.update(num_epochs=3, batch_size=2)
    drawn = []
    real = sw.draw_batch

    def spy(batch):
        imgs = real(batch)
        drawn.append(np.concatenate([im.z.numpy().ravel() for im in imgs]))
        return imgs
    sw.draw_batch = spy
    heads0 = None

    def spy_heads():
        nonlocal heads0
        type(sw).build_heads(sw)
        # give the ranks DIFFERENT initial heads, as un-broadcast per-rank inits would be
        for p in sw._params():
            p.add_(0.01 * rank)
    sw.build_heads = spy_heads
    sw.pretrain(None, num_test_samples=1)
    out[rank] = (np.stack(drawn), [p.cpu().numpy() for p in sw._params()],
                 [float(l.item()) for l in sw.loss_history])
    dist.barrier()
    dist.destroy_process_group()


def test_pretrain_two_ranks_draw_different_images_and_stay_replicated():
    """pretrain() itself under 2 ranks (gloo, sharing the GPU): rank 0's heads are broadcast, each rank
    draws its own latents (a per-rank reseed — the import-time seed 42 is the same everywhere), and the
    replicas hold bit-identical parameters and losses after every step."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    sd = golden_sd(load_golden("gv10_pretrain"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_pretrain_worker, args=(2, port, sd, out), nprocs=2, join=True)
    z0, p0, l0 = out[0]
    z1, p1, l1 = out[1]
    assert z0.shape == z1.shape == (3, 2 * 16)
    assert not np.allclose(z0, z1)                       # different images per rank
    for a, b in zip(p0, p1):
        np.testing.assert_array_equal(a, b)               # replicas identical
    assert l0 == l1 and all(np.isfinite(l0))
