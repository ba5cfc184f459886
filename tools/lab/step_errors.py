"""Measured error of one full-size optimizer step against the oracle (loss, the three gradients, the updated
parameters) — the numbers behind the tolerances of tests/test_hip_fullsize.py.
    python tools/lab/step_errors.py [batch] [patches] [prototypes] [engine 1|0] [own]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from test_hip_fullsize import Cfg, _oracle_images, ffhq_args, full_generator, rel   # noqa: E402
from oracle import swav_ref as S                                                   # noqa: E402
from ganecdotes_amd import ops                                                     # noqa: E402
from ganecdotes_amd.draws import draw_image                                        # noqa: E402
from ganecdotes_amd.swav_clustering import SwAVClustering                          # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
patches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nproto = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ops.set_gemm_mode(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
torch.set_num_threads(min(16, torch.get_num_threads()))
g, sd = full_generator(256)
g = g.cuda()
sw = SwAVClustering(model=g, model_config=Cfg, out_dir=None, device="cuda",
                    **ffhq_args(batch, nproto, num_patches=patches, pick_source="host"))
torch.manual_seed(5)
np.random.seed(5)
with torch.no_grad():
    ml = S.G.mean_latent(sd, 128, 512)
sw.mean_latent = ml.cuda()
sw.build_heads()
params = S.SwavParams(*[p.detach().cpu().clone() for p in sw._params()])
images = [draw_image(512, 13, 6, 256 * 256, patches, transforms=True) for _ in range(batch)]
loss = sw.train_step(images)
own = len(sys.argv) > 5 and sys.argv[5] == "own"       # the oracle's OWN rotate / flip source maps instead of the HIP ones
oimgs = [S.ImageDraws(im.z, [S.ViewDraws(v.layer_no, v.map_noise, v.angle, v.flip) for v in im.views], im.picks)
         for im in images] if own else _oracle_images(images, 256)
if own:
    agree = [float((ops.rotflip_map(v.angle, v.flip, 256, 256, "cuda").cpu().long() ==
                    S.rotflip_index_map(v.angle, v.flip, 256, 256)).float().mean()) for im in images for v in im.views]
    print("source-map agreement per view:", " ".join(f"{a:.5f}" for a in agree))
ref_loss, grads, _ = S.swav_train_step(
    sd, ml, 0.7, oimgs, params, [None] * 3, lr=0.01, momentum=0.9, trust=0.01, use_larc=True,
    hlen=5376, patch_size=20000, num_patches=patches, temperature=0.01, eps=0.005, niters=10, source_pdf="uniform",
    sinkhorn_scope="global")
print(f"B={batch} P={patches} K={nproto} engine={ops.get_gemm_mode()}: loss {loss.item():.6f} oracle {ref_loss:.6f} "
      f"rel {abs(loss.item() - ref_loss) / abs(ref_loss):.2e}")
for name, got, want in zip(("dWp", "dWc", "db"), sw._grads, grads):
    print(f"  {name}: max-norm rel {rel(got.cpu(), want):.2e}")
for name, got, want in zip(("Wp", "Wc", "b"), sw._params(), (params.proj_w, params.proto_w, params.proto_b)):
    print(f"  {name} after the update: {rel(got.cpu(), want):.2e}")
