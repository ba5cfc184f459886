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


def _args(scope, pdf):
    swav = dict(num_epochs=1, num_samples=1, num_patches=2, sampling_method="random", patch_size=300,
                hf_interp="nearest", warmup_epochs=1, start_warmup=0.01, use_scheduler=False, base_lr=0.01,
                final_lr=0.0001, trust_coeff=0.01, train_args=dict(lr=0.01, momentum=0.9), projn_nw="linear",
                temperature=0.1, nprototypes=12, nclasses=8, hlen=56, add_local_loss=False, plot_test_images=False,
                epoch_print_freq=1, max_masks=4, sinkhorn_scope=scope)
    return dict(perturb_args=dict(truncation=0.7, n_layers=3, n_samples=1, layer_no=None, perturb_std=[1.0] * 3),
                swav_args=swav, sinkhorn_args=dict(source_pdf=pdf, niters=10, eps=0.05))


def _build(sd, group, scope, pdf):
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.swav_clustering import SwAVClustering
    g = Generator(32, 16, 2, channels=MINI)
    g.load_state_dict(sd, strict=False)
    g = g.eval().cuda()
    sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda", dist_group=group, **_args(scope, pdf))
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


def _worker(rank, world, port, sd, scope, pdf, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sw = _build(sd, dist.group.WORLD, scope, pdf)
    imgs = _images(4)
    loss = None
    for _ in range(2):
        loss = sw.train_step(imgs[rank * 2:(rank + 1) * 2])
    out[rank] = (float(loss.item()), [p.cpu().numpy() for p in sw._params()])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scope,pdf", [("global", "uniform"), ("image", "uniform"), ("global", "image")])
def test_two_ranks_equal_one_rank(scope, pdf):
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
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, sd, scope, pdf, out), nprocs=2, join=True)
    for r in range(2):
        np.testing.assert_allclose(out[r][0], ref[0], rtol=2e-4)
        for a, b in zip(out[r][1], ref[1]):
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-6 * np.abs(b).max())
    for a, b in zip(out[0][1], out[1][1]):          # replicas stay identical
        np.testing.assert_array_equal(a, b)
