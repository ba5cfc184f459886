"""Host-side logic that needs no GPU: RNG draw order, the C-ABI library's exported symbols,
config/plugin surface."""
import ctypes
import os
import re

import numpy as np
import torch

from conftest import ROOT, load_golden

T = torch.from_numpy


def test_draws_follow_reference_rng_order():
    from ganecdotes_amd.draws import draw_image
    d = load_golden("gv9_view_rng")
    torch.manual_seed(42)
    np.random.seed(42)
    torch.randn(64, 16)                                  # mean_latent draw of the constructor
    im = draw_image(16, 7, 3, 32 * 32, 0, transforms=False)
    assert torch.equal(im.z, T(d["z"]))
    assert torch.equal(torch.randn(4), T(d["after"]))
    for v, tag in zip(im.views, "st"):
        lno = int(d[f"layer_no_{tag}"])
        assert v.layer_no == lno
        ref = T(d[f"replay_map_noise_{tag}"])[[2 * lno, 2 * lno + 1], 0]
        assert torch.equal(v.map_noise, ref)


def test_draws_with_transforms_and_picks():
    from ganecdotes_amd.draws import draw_image
    torch.manual_seed(0)
    np.random.seed(0)
    im = draw_image(16, 7, 3, 64, 5)
    assert len(im.picks) == 5 and all(sorted(p.tolist()) == list(range(64)) for p in im.picks)
    assert all(-10 <= v.angle <= 10 for v in im.views)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/ganecdotes_hip.h declares
    (no compute calls: there is no GPU here)."""
    from ganecdotes_amd import _lib, build
    build.build(verbose=False)
    header = open(os.path.join(ROOT, "include", "ganecdotes_hip.h")).read()
    declared = set(re.findall(r"\b(gx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) > 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.load()
    assert b"gfx950" in L.gx_version()
    assert L.gx_sinkhorn_workspace_bytes(8) == (16 + 512 * 8) * 4
    assert L.gx_gemm_f32_workspace_bytes(512, 5376, 8) == 8 * 512 * 5376 * 4


def test_product_path_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ganecdotes_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                    bad.append(f)
    assert not bad, bad
