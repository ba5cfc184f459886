#!/usr/bin/env python3
"""Headline benchmark: pretrain images/sec of the per-pixel SwAV path (ffhq-256, hfc_with_swav).

    python bench.py --gpus N --steps K --warmup W
        N>1: either under `python -m torch.distributed.run --nproc-per-node N ...` (RANK / WORLD_SIZE from the
        environment) or plain — then this script starts its N ranks itself (one process per GPU, RCCL).

One "step" = one optimizer step of the hot path over one batch of synthetic latents on every
rank: draws -> 2 views (generator taps) -> 5 patches x {gather, projection, prototypes,
Sinkhorn, swapped CE, backward} -> LARC-SGD.  Workload = BASELINE.json configs[1]
(ffhq-256 hfc_with_swav, 8 prototypes, batch 16 per GPU); random-init generator weights
under seed 42 and random latents (no checkpoints exist offline).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA = 157.3e12          # /opt/skills/guides/MI355X_MICROARCH.md, dense v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA = 2500e12          # same guide: dense v_mfma_f32_32x32x16_bf16 (no sparsity)
PEAK_HBM = 8.0e12
TRAFFIC_FILE = "r05_traffic.json"   # PMC passes of this round (tools/profile_bench.sh): HBM bytes per launch of the dominant kernel


class ModelCfg:
    num_latents_for_mean = 4096
    truncation = 0.7
    latent_dim = 512


def workload(batch, nproto, size=256, patch=20000, scope="global"):
    layer_hf_dim = [512, 1024, 1024, 1024, 1024, 512, 256] if size == 256 else None
    hlen = 5376
    return dict(
        perturb_args=dict(truncation=0.7, n_layers=6, n_samples=1, layer_no=None, perturb_std=[1.0] * 6),
        swav_args=dict(num_epochs=1, num_samples=1, num_patches=5, sampling_method="random", patch_size=patch,
                       hf_interp="nearest", warmup_epochs=1, start_warmup=0.01, use_scheduler=False, base_lr=0.01,
                       final_lr=0.0001, trust_coeff=0.01, train_args=dict(lr=0.01, momentum=0.9),
                       projn_nw="linear", temperature=0.01, nprototypes=nproto, nclasses=512, hlen=hlen,
                       add_local_loss=False, plot_test_images=False, epoch_print_freq=5, max_masks=4,
                       batch_size=batch, sinkhorn_scope=scope, pick_source="device"),
        sinkhorn_args=dict(source_pdf="uniform", niters=10, eps=0.005),
        layer_hf_dim=layer_hf_dim)


def build_generator(size, device):
    from ganecdotes_amd.generator import Generator
    torch.manual_seed(42)
    g = Generator(size, 512, 8)
    for n, p in g.named_parameters():
        if n.endswith("noise.weight"):
            p.data.fill_(0.1)        # init is 0 => the noise path would be untested (SURVEY par. 8d)
    return g.eval().to(device)


def host_cpu_info():
    """What the host offers this process: CPU model, logical CPUs (nproc), physical cores, the
    scheduler affinity and the cgroup CPU quota — the thread count used is the smallest of the
    last three (oversubscribing a cgroup share only slows the oracle down)."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid)); pid = cid = None
    except OSError:
        pass
    nproc = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = nproc
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    physical = len(phys) or nproc
    threads = max(1, min(physical, affinity, int(quota) if quota and quota >= 1 else physical))
    return dict(cpu_model=model, nproc=nproc, physical_cores=physical, affinity=affinity,
                cgroup_cpu_quota=quota, threads=threads)


def cpu_baseline(nproto, threads=None):
    """The oracle (CPU restatement of the reference op sequence: materialised upsample+concat, rotation
    / flip of the whole stack, norm maps, full-permute pixel sampling, N x 5376 projection GEMM, unfused
    Sinkhorn, autograd backward) timed on the host cores on ONE whole image of the same workload: both
    views at 256 x 256 and all 5 patches forward + backward — nothing extrapolated.  K = the GPU line's."""
    from oracle import stylegan2_ref as G
    from oracle import swav_ref as S
    from ganecdotes_amd.generator import Generator
    info = host_cpu_info()
    threads = threads or info["threads"]
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    g = Generator(256, 512, 8)
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    for k in sd:
        if k.endswith("noise.weight"):
            sd[k].fill_(0.1)
    del g
    with torch.no_grad():
        ml = G.mean_latent(sd, 256, 512)
    proj_w, _ = S.init_linear_like_torch(512, 5376, False)
    proto_w, proto_b = S.init_linear_like_torch(nproto, 512, True)
    im = S.draw_image_reference_order(512, 13, 6, 256 * 256, 5, transforms=True)
    t0 = time.time()
    with torch.no_grad():
        w = G.mapping(im.z, sd)
        stacks = [S.make_view(sd, w, ml, 0.7, vd, 5376)[0] for vd in im.views]
        _ = [S.norm_map(s) for s in stacks]
    t_views = time.time() - t0
    t0 = time.time()
    leaves = [proj_w.requires_grad_(True), proto_w.requires_grad_(True), proto_b.requires_grad_(True)]
    loss = 0.0
    for pno in range(5):
        sc = [S.codes(S.sample_pixels(s, im.picks[pno], 20000), *leaves) for s in stacks]
        with torch.no_grad():
            q = [S.sinkhorn_knopp(s, 0.005, 10) for s in sc]
        loss = loss + S.swapped_prediction_loss(sc[0] / 0.01, sc[1] / 0.01, q[0], q[1])
    (loss / 5).backward()
    with torch.no_grad():
        S.larc_sgd_step(leaves, [p.grad for p in leaves], [None] * 3, 0.01, 0.9, 0.01, True)
    t_patches = time.time() - t0
    return dict(value=1.0 / (t_views + t_patches), unit="images/s", cores=threads, kind="port",
                cpu_model=info["cpu_model"], nproc=info["nproc"], physical_cores=info["physical_cores"],
                affinity=info["affinity"], cgroup_cpu_quota=info["cgroup_cpu_quota"],
                sample=f"1 whole image, K={nproto}: 2 views at 256x256 incl. rotate/flip + norm maps "
                       f"({t_views:.1f} s) + all 5 patches fwd+bwd at N=20000 and the LARC-SGD step "
                       f"({t_patches:.1f} s); oracle (torch CPU fp32, {threads} threads)")


def self_launch(n, argv=None, poll_s=0.2, grace_s=5.0):
    """Run this script as n ranks on one node (what `torch.distributed.run --nproc-per-node n` would do):
    one child process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.  The parent touches
    no GPU; children are ordinary subprocesses (never an exec of a process that has initialised HIP).
    All children are polled: the first one that exits non-zero takes the others down (terminate, then kill
    after `grace_s`) and its code is returned within seconds — a rank that dies during start-up must not leave
    its peers waiting in the rendezvous or in a collective until the backend's timeout.  Every rank's stderr
    (and the stdout of ranks > 0) is relayed line by line with a `[rank r]` prefix; rank 0's JSON line is
    printed last."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    argv = sys.argv[1:] if argv is None else argv
    procs, readers, out0 = [], [], []

    def relay(stream, prefix, keep):
        for raw in iter(stream.readline, b""):
            line = raw.decode(errors="replace").rstrip("\n")
            if keep is not None and line.startswith("{"):
                keep.append(line)                                   # rank 0's JSON line(s)
            else:
                print(f"{prefix}{line}", file=sys.stderr, flush=True)
        stream.close()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # dmabuf IPC: the pool's host driver has no legacy IPC, and without this RCCL's peer-memory exchange fails
        # with `hipIpcGetMemHandle: invalid argument` (the image exports it already; kept for a bare environment)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        procs.append(p)
        for stream, keep in ((p.stdout, out0 if r == 0 else None), (p.stderr, None)):
            t = threading.Thread(target=relay, args=(stream, f"[rank {r}] ", keep), daemon=True)
            t.start()
            readers.append(t)
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[bench] rank {r} exited with code {code}: stopping the other ranks", file=sys.stderr, flush=True)
        if live and rc == 0:
            time.sleep(poll_s)
    if live:                                                        # a rank failed: take the rest down
        for r in live:
            procs[r].terminate()
        deadline = time.time() + grace_s
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    for t in readers:
        t.join(timeout=5.0)
    if rc == 0:
        for line in out0:           # ONE JSON line (backends may chat on stdout: "[Gloo] Rank 0 is ..." goes to stderr)
            print(line, flush=True)
    return rc


def measure(sw, args, steps, warmup, dev, world):
    """`warmup` untimed steps, then exactly `steps` steps bracketed by barrier + synchronize on both sides;
    returns (seconds = max over ranks, per-tag HIP-event statistics, last loss)."""
    import queue
    import threading
    from ganecdotes_amd import ops

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def feed(n):
        """Background producer: host draws (reference RNG order) + pinned upload of batch i+1
        overlap the kernels of batch i.  Started INSIDE the timed region for the timed steps."""
        q = queue.Queue(maxsize=2)

        def run():
            torch.cuda.set_device(dev)             # a new thread starts on device 0: stage on this rank's GPU
            for _ in range(n):
                q.put(sw.prepare_batch(sw.draw_batch(args.batch)))
        threading.Thread(target=run, daemon=True).start()
        return q

    overlap = getattr(args, "overlap", "on") == "on"

    def run(q, n):
        """n steps.  overlap: the views of batch i + 1 are queued on the view stream BEFORE step i is queued (the frozen
        generator's work runs beside the step, SwAVClustering.prefetch_views — what pretrain() does); the views of the first
        batch are queued here too, so n steps always contain n view passes and n training passes."""
        loss = None
        if n <= 0:
            return loss
        nxt = sw.prefetch_views(q.get()) if overlap else q.get()
        for i in range(n):
            cur = nxt
            if i + 1 < n:
                nxt = sw.prefetch_views(q.get()) if overlap else q.get()
            loss = sw.train_step(cur)
        return loss

    run(feed(warmup), warmup)
    ops.PROFILE = {}
    sync()
    t0 = time.perf_counter()
    loss = run(feed(steps), steps)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # HIP events on the launch stream over the timed region (ops.record brackets the launches of the MFMA kernels with
    # their algorithmic FLOPs and of the large HBM-bound passes with their algorithmic bytes: SURVEY par. 8d figures x the
    # units the launch processes)
    prof, ops.PROFILE = ops.PROFILE or {}, None
    stats = {}
    for tag, evs in prof.items():
        ms = sum(a.elapsed_time(b) for a, b, *_ in evs)
        stats[tag] = dict(ms_per_step=ms / steps, launches_per_step=sum(e[4] for e in evs) / steps,
                          flops_per_step=sum(e[2] for e in evs) / steps, bytes_per_step=sum(e[3] for e in evs) / steps)
    return dt, stats, float(loss.item())


def measure_nested(sw, args, steps, warmup, dev, world):
    """`measure` for the short nested records: two timed blocks of `steps` (the second without warm-up), the faster one kept.  A
    block of 5-8 steps right after another configuration's tens of GB were returned to the driver has been seen to contain
    one multi-second stall inside the view phase (1 run in 4 on one box: cat_per_rank 407 instead of 21 ms/step with every
    kernel at its usual time); the headline measurement — 60 steps, its own warm-up — is a single block as the contract says."""
    best = None
    for rep_, wu in enumerate((warmup, 0)):
        r = measure(sw, args, steps, wu, dev, world)
        if best is None or r[0] < best[0]:
            best = r
    return best


def roofline_of(stats, ms_per_step, traffic_file):
    """The `roofline` object of the JSON line: the dominant MFMA kernel (algorithmic FLOPs per launch / its average
    launch time, HIP events), the whole step against the same peak, and the largest HBM-bound pass."""
    from ganecdotes_amd import ops
    kern = {k: v for k, v in stats.items() if not k.startswith("phase:") and v["flops_per_step"] > 0}
    if not kern:
        return None
    name = max(kern, key=lambda k: kern[k]["ms_per_step"])
    v = kern[name]
    ach = v["flops_per_step"] / (v["ms_per_step"] * 1e-3)
    x6 = ops.get_gemm_mode() == 1
    # products per fp32 product on the matrix cores: 3 (f16x3: two scaled fp16 planes per operand) where the named kernel class
    # runs that form — the x-planes convs under ops.get_conv_planes() == 1, the weight-plane GEMMs under ops.GEMM_H2 (the
    # TN gradient GEMMs still run bf16x6: pricing the whole class at the f16x3 peak is the conservative choice) — else 6
    h3 = x6 and ((name == "modconv_kernel" and ops.get_conv_planes() == 1) or (name == "gemm_f32_kernel" and ops.GEMM_H2))
    nprod = 3.0 if h3 else 6.0
    eng = ("fp16 MFMA, f16x3: each fp32 operand scaled by a power of two and split into two fp16 planes, three MFMA products "
           "per fp32 product, fp32 accumulate (fp32-class error against fp64, tools/lab/mfma_f16x3.hip)" if h3 else
           "bf16 MFMA, bf16x6: each fp32 operand split exactly into three bf16, six MFMA products per fp32 "
           "product, fp32 accumulate" if x6 else "fp32 MFMA")
    what = {"modconv_kernel": f"gx::modconv_xp_kernel / gx::modconv_kernel (3x3 modulated conv, implicit GEMM on {eng}; "
                              "the modulated activations are split once by their producer and both operands reach LDS by "
                              "LDS-DMA; algorithmic FLOPs = 2*B*Hin*Win*Cin*Cout*9 per layer)",
            "gemm_f32_kernel": f"gx::gemm kernels (pyramid projection / prototype / gradient GEMMs on {eng}; "
                               "algorithmic FLOPs = 2*M*N*K per launch)"}.get(name, name)
    # peak for ALGORITHMIC fp32 FLOPs: the fp32 MFMA peak, or the dense 16-bit MFMA peak / the products per fp32 product
    # (bf16x6: / 6; f16x3: / 3 — fp16 and bf16 share the 2.5 PFLOP/s dense peak)
    peak = PEAK_BF16_MFMA / nprod if x6 else PEAK_F32_MFMA
    # HBM bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    # tools/pmc_traffic.py; counters cannot be read live without the profiler attached)
    traffic = traffic_src = None
    if traffic_file and os.path.exists(traffic_file):
        stem = {"modconv_kernel": "modconv", "gemm_f32_kernel": "gemm"}.get(name, name)
        cands = [t for k, t in json.load(open(traffic_file)).items() if stem in k]
        if cands:
            traffic = max(cands, key=lambda t: t["launches"])["hbm_bytes_per_launch"]
            traffic_src = (f"from profiles/{os.path.basename(traffic_file)}: rocprofv3 --pmc passes over this same command "
                           f"(tools/profile_bench.sh), committed; NOT measured in this run")
    step_flops = sum(x["flops_per_step"] for x in kern.values())
    roof = dict(bound="mfma", achieved=ach / 1e12, peak=peak / 1e12, unit="TFLOP/s",
                frac=ach / peak, traffic=traffic, traffic_source=traffic_src, kernel=what,
                engine=("f16x3" if h3 else "bf16x6") if x6 else "f32mfma",
                mfma_tflops_issued=(nprod if x6 else 1.0) * ach / 1e12,
                mfma_peak=(PEAK_BF16_MFMA if x6 else PEAK_F32_MFMA) / 1e12,
                fp32_mfma_peak=PEAK_F32_MFMA / 1e12,
                launches_per_step=v["launches_per_step"],
                avg_launch_ms=v["ms_per_step"] / v["launches_per_step"],
                gflop_per_launch=v["flops_per_step"] / v["launches_per_step"] / 1e9,
                # the whole step against the same peak: the algorithmic FLOPs of every contraction the step EXECUTES
                # (conv + GEMM launches; restructurings that remove products lower this count) / wall time per step
                step_tflop=step_flops / 1e12, step_frac=step_flops / (ms_per_step * 1e-3) / peak,
                ms_per_step_by_kernel={k: round(x["ms_per_step"], 3) for k, x in stats.items()})
    hb = {k: x for k, x in stats.items() if not k.startswith("phase:") and x["flops_per_step"] == 0 and x["bytes_per_step"] > 0}
    if hb:
        hname = max(hb, key=lambda k: hb[k]["ms_per_step"])
        h = hb[hname]
        gbs = h["bytes_per_step"] / (h["ms_per_step"] * 1e-3)
        roof["hbm"] = dict(kernel=hname, bound="hbm", achieved=gbs / 1e9, peak=PEAK_HBM / 1e9, unit="GB/s", frac=gbs / PEAK_HBM,
                           gbytes_per_step=h["bytes_per_step"] / 1e9, ms_per_step=h["ms_per_step"],
                           launches_per_step=h["launches_per_step"])
    return roof


def nested_record(workload_str, images_per_step, dt, steps, warmup, stats, loss=None, extra=None):
    """One of the nested per-configuration records of the JSON line: ms per step, images/s, and the dominant kernel of THAT
    configuration against its roofline (HIP events of this run), the whole step against the MFMA peak, the largest HBM-bound pass."""
    ms = 1e3 * dt / steps
    r = roofline_of(stats, ms, None) or {}
    hb = r.get("hbm")
    rec = {"workload": workload_str, "steps": steps, "warmup": warmup, "ms_per_step": ms,
           "value": images_per_step * steps / dt, "unit": "images/s",
           "roofline": {"kernel": r.get("kernel"), "bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"),
                        "unit": r.get("unit"), "frac": r.get("frac"), "step_tflop": r.get("step_tflop"),
                        "step_frac": r.get("step_frac"),
                        "hbm": None if not hb else {k: hb[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "ms_per_step")},
                        "ms_per_step_by_kernel": r.get("ms_per_step_by_kernel")}}
    if loss is not None:
        rec["loss"] = loss
    if extra:
        rec.update(extra)
    return rec


def timed_loop(step, steps, warmup):
    """`warmup` untimed calls of step(), then `steps` timed ones between two synchronizes; (seconds, HIP-event statistics)"""
    from ganecdotes_amd import ops
    for _ in range(warmup):
        step()
    ops.PROFILE = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    stats = {}
    for tag, evs in prof.items():
        ms = sum(a.elapsed_time(b) for a, b, *_ in evs)
        stats[tag] = dict(ms_per_step=ms / steps, launches_per_step=sum(e[4] for e in evs) / steps,
                          flops_per_step=sum(e[2] for e in evs) / steps, bytes_per_step=sum(e[3] for e in evs) / steps)
    return dt, stats


def simclr_record(g, dev, iters=200, warmup=20):
    """BASELINE.json configs[4]: ffhq-256 hfc_with_simclr (baseline/hfc_with_simclr/simclr_clustering.py:131-280) — one image and
    20 picked pixel pairs per iteration, T = 1.0, the 2-layer MLP with train-mode BatchNorm, LARC-SGD"""
    from ganecdotes_amd.simclr_clustering import SimCLRClustering
    a = dict(num_iters=1, batch_size=20, patch_size=20000, hf_interp="nearest", trust_coeff=0.01,
             train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=512, hlen=5376,
             epoch_print_freq=1, max_masks=4, use_larc=True, transforms=True)
    sim = SimCLRClustering(model=g, model_config=ModelCfg, out_dir=None, device=dev, simclr_args=a,
                           perturb_args=dict(truncation=0.7, n_layers=6, n_samples=1, layer_no=None, perturb_std=[1.0] * 6))
    torch.manual_seed(42)
    np.random.seed(42)
    sim.build_heads()
    last = []
    dt, stats = timed_loop(lambda: last.append(sim.train_step(sim.draw_iteration())), iters, warmup)
    return nested_record("ffhq-256 hfc_with_simclr (BASELINE.json configs[4]): 1 image per iteration, both views at 256x256, 20 "
                         "shared pixel picks per view -> 40 rows, Linear(5376,512)-BatchNorm1d(train)-LeakyReLU-Linear(512,512), "
                         "NT-Xent at T=1.0 with the reference's wrap-around pairs, LARC-SGD; host draws included",
                         1, dt, iters, warmup, stats, loss=float(last[-1].item()),
                         extra={"iterations_per_s": iters / dt})


def predict_record(g, dev, batch=16, steps=6, warmup=2):
    """The evaluate.py loop per test latent (src/one_shot_pipeline.py:632-665): predict_swav_codes (generator with fixed noise,
    projection of all 65 536 pixels, argmax) + the one-shot head ('XXS': Conv2d(512, 12, 3, pad 1), config :77-78) + argmax"""
    from ganecdotes_amd.oneshot import OneShotHead
    from ganecdotes_amd.swav_clustering import SwAVClustering
    wl = workload(batch, 8)
    sw = SwAVClustering(model=g, model_config=ModelCfg, out_dir=None, device=dev, **wl)
    torch.manual_seed(42)
    sw.build_heads()
    head = OneShotHead(512, 12, size="XXS", device=dev)
    lat = g.style(torch.randn(batch, 512, device=dev))

    def step():
        codes, _ = sw.predict_swav_codes(lat)
        head.predict(codes.permute(0, 2, 3, 1))
    dt, stats = timed_loop(step, steps, warmup)
    return nested_record(f"ffhq-256 predict (evaluate.py / run_tests): {batch} test latents per call -> generator (fixed noise, one "
                         f"truncation; the RGB image, which predict_swav_codes discards, is not rendered) -> pyramid projection of every pixel (31 GFLOP per image instead of the "
                         f"reference's 361: DESIGN 4.1) -> (B,512,256,256) codes + argmax -> 'XXS' one-shot head (3x3 conv 512->12) "
                         f"-> label map",
                         batch, dt, steps, warmup, stats)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--prototypes", type=int, default=8)
    ap.add_argument("--scope", default="global", choices=["global", "image"],
                    help="Sinkhorn scope: 'global' = one problem per view and patch over the whole (sharded) batch, normalised "
                         "across ranks by all-reduce as BASELINE.json's north_star asks (default); 'image' = one problem per "
                         "image, view and patch (the reference's batch-1 semantics, no Sinkhorn collective)")
    ap.add_argument("--pdf", default="uniform", choices=["uniform", "image"],
                    help="Sinkhorn marginals: 'uniform' (BASELINE.json's config, hfc_with_swav_ffhq_config.py) or 'image' = the "
                         "histograms of the norm maps (hfc_with_swav_cat_config.py:63-65; eps defaults to that file's 0.003); "
                         "not the headline configuration: a line with it says so in config.workload")
    ap.add_argument("--eps", type=float, default=0.0, help="Sinkhorn eps (default: 0.005, or 0.003 with --pdf image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ds-table", default="auto", choices=["auto", "on", "off"],
                    help="many prototypes: d(scores) summed per pixel before the backward GEMMs (DESIGN 4.15); auto = when "
                         "the patches draw more samples than the image has pixels")
    ap.add_argument("--deterministic", action="store_true",
                    help="swav_args['deterministic']: fixed-point coefficient scatter on the K <= 16 route — bitwise run-to-run "
                         "reproducible steps (not the headline configuration: a line with it says so in config.workload)")
    ap.add_argument("--sinkhorn-table", default="auto", choices=["auto", "on", "off"],
                    help="many prototypes: the table Sinkhorn (rows grouped by membership, DESIGN 4.16) or — off — the per-problem "
                         "sweeps on pixel-ordered tables (round 3's form); the A/B of the two table orderings")
    ap.add_argument("--reference-faithful", default="auto", choices=["auto", "on", "off"],
                    help="also time the reference's own nprototypes = 5000 (hfc_with_swav_ffhq_config.py:52) for a few steps and "
                         "nest the result in the JSON line; auto = on one GPU with the default 8 prototypes only")
    ap.add_argument("--all-configs", default="auto", choices=["auto", "on", "off"],
                    help="also time, for a few steps each and nested in the same JSON line, the per-rank shapes of the other "
                         "BASELINE.json configurations: cat-256 (16 prototypes, 'image' pdf, eps 0.003), car-512 (true 512^2 "
                         "generator, 4 images, 16 and 4000 prototypes, eps 0.01), hfc_with_simclr, and predict + one-shot head; "
                         "auto = on one GPU with the default 8 prototypes only")
    ap.add_argument("--overlap", default="on", choices=["on", "off"],
                    help="on: the views of batch i+1 (frozen generator) are queued on their own stream before step i, so they "
                         "run beside it (what pretrain() does, swav_args['overlap_views']); off: one stream, views then step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo lets several ranks share ONE GPU to rehearse N>1")
    ap.add_argument("--dry-rendezvous", action="store_true",
                    help="launcher check: every rank joins the process group, runs one barrier and exits (no GPU work)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves — fresh child processes, created
        # before this process has made any GPU call (it never does); rank 0's JSON line is relayed.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("GX_BENCH_FAIL_RANK") == str(rank):          # test hook: a rank that dies during start-up
        print(f"rank {rank}: GX_BENCH_FAIL_RANK set, exiting with code 3", file=sys.stderr, flush=True)
        sys.exit(3)
    from datetime import timedelta
    pg_timeout = timedelta(seconds=120)
    if args.dry_rendezvous:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_rendezvous": True, "world": world}), flush=True)
        return
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 1) // max(1, world))))   # the box's CPU share
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if world > 1 and args.backend == "nccl" and ndev < world:
        raise SystemExit(f"--backend nccl needs one GPU per rank: {world} ranks but {ndev} visible device(s) "
                         f"(use --backend gloo to rehearse several ranks on one GPU)")
    local = local % ndev                     # gloo rehearsal: several ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout)
        group = dist.group.WORLD

    from ganecdotes_amd import ops
    from ganecdotes_amd.swav_clustering import SwAVClustering
    g = build_generator(256, dev)

    def make(nproto, gen=None, batch=None, size=256, pdf=None, eps=None, nlayers=6):
        pdf = args.pdf if pdf is None else pdf
        eps = args.eps if eps is None else eps
        wl = workload(args.batch if batch is None else batch, nproto, size=size, scope=args.scope)
        wl["perturb_args"].update(n_layers=nlayers, perturb_std=[1.0] * nlayers)
        if pdf != "uniform":
            wl["sinkhorn_args"].update(source_pdf=pdf, eps=eps if eps else 0.003)
        elif eps:
            wl["sinkhorn_args"]["eps"] = eps
        if args.ds_table != "auto":
            wl["swav_args"]["ds_table"] = args.ds_table == "on"
        if args.sinkhorn_table != "auto":
            wl["swav_args"]["sinkhorn_table"] = args.sinkhorn_table == "on"
        if args.deterministic:
            wl["swav_args"]["deterministic"] = True
        sw = SwAVClustering(model=g if gen is None else gen, model_config=ModelCfg, out_dir=None, device=dev, dist_group=group, **wl)
        torch.manual_seed(42 + rank)
        np.random.seed(42 + rank)
        sw.build_heads()
        if world > 1:   # replicated heads: rank 0's init everywhere
            import torch.distributed as dist
            for p in sw._params():
                dist.broadcast(p, 0)
        return sw

    sw = make(args.prototypes)
    dt, stats, loss = measure(sw, args, args.steps, args.warmup, dev, world)
    images = args.steps * args.batch * world
    ms_step = 1e3 * dt / args.steps
    if os.environ.get("GX_BENCH_EVENTS"):
        for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms_per_step"]):
            print(f"# {k:24s} {v['ms_per_step']:9.2f} ms/step  {v['launches_per_step']:7.1f} launches  "
                  f"{v['flops_per_step'] / max(v['ms_per_step'], 1e-9) / 1e9:8.1f} TFLOP/s  "
                  f"{v['bytes_per_step'] / max(v['ms_per_step'], 1e-9) / 1e6:8.1f} GB/s", file=sys.stderr)
    roof = roofline_of(stats, ms_step, os.path.join(ROOT, "profiles", TRAFFIC_FILE))
    if args.overlap == "on" and roof is not None:
        # The timed region runs TWO streams (the next batch's views beside the step).  A HIP-event bracket around a launch there
        # also contains the time the launch waited, queued, behind the other stream's kernel — rocprofv3's per-dispatch
        # durations over the same command do NOT (profiles/r05_bench_b16_k8_kernel_stats.csv: 1.34 ms per conv launch in both
        # kinds of step).  The kernel figures (`achieved`, `frac`, `avg_launch_ms`, `hbm`) are therefore taken from six more
        # steps on ONE stream, same process, same state, where an event bracket IS the launch — the number that agrees with
        # rocprofv3; the brackets of the timed region are kept under `timed_region`; `step_tflop` / `step_frac` / `value` are
        # wall-clock figures of the timed (two-stream) region.
        class _X:
            batch, overlap = args.batch, "off"
        dtx, stx, _ = measure(sw, _X, 6, 1, dev, world)
        rx = roofline_of(stx, 1e3 * dtx / 6, os.path.join(ROOT, "profiles", TRAFFIC_FILE))
        if rx is not None:
            timed = {k: roof[k] for k in ("achieved", "frac", "avg_launch_ms", "launches_per_step", "mfma_tflops_issued",
                                          "ms_per_step_by_kernel", "hbm") if k in roof}
            for k in ("achieved", "frac", "avg_launch_ms", "launches_per_step", "gflop_per_launch", "mfma_tflops_issued", "hbm"):
                if k in rx:
                    roof[k] = rx[k]
            roof["one_stream_ms_per_step"] = 1e3 * dtx / 6
            roof["one_stream_step_frac"] = rx["step_frac"]
            roof["one_stream_ms_per_step_by_kernel"] = rx["ms_per_step_by_kernel"]
            roof["timed_region"] = timed
            roof["concurrent_streams"] = 2
            roof["note"] = ("value / ms_per_step / step_frac: the timed region (views of batch i+1 on their own stream beside step "
                            "i).  achieved / frac / avg_launch_ms / hbm: the same kernels over 6 further steps on one stream, "
                            "where a HIP-event bracket is the launch itself (agrees with rocprofv3); `timed_region` holds the "
                            "brackets of the two-stream region, which include time queued behind the other stream")
    x6 = ops.get_gemm_mode() == 1
    out = {
        "metric": "pretrain images/sec (ffhq-256 hfc_with_swav)", "value": images / dt, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "arithmetic": (("fp32 in / fp32 out; the modulated convs" +
                        (", the weight-plane and long-k gradient GEMMs and the finest level's quadratic-form passes" if ops.GEMM_H2 else "") +
                        " as f16x3 on the fp16 matrix cores (operands scaled by powers of two and split into two fp16 planes, "
                        "three products, fp32 accumulate: the error against fp64 is that of an fp32 fma chain — "
                        "tools/lab/mfma_f16x3.hip, tests/test_hip_kernels.py::test_gemm_nt_planes_f16x3_form_is_fp32_class; operands "
                        "out of the fp16 range come out NaN; every bound is stated, or found on the device by the operand's producer), the short "
                        "contractions as bf16x6 (three bf16 planes, six products); "
                        "GX_CONV_PLANES=0 GX_GEMM_PLANES=0 select bf16x6 throughout, GX_GEMM_MODE=0 the exact-fp32 MFMA")
                       if x6 and ops.get_conv_planes() == 1 else
                       "fp32 in / fp32 out; contractions as bf16x6 on the bf16 matrix cores (every operand bit kept: fp32-level "
                       "error, tests/test_hip_kernels.py::test_gemm_x6_accuracy); GX_GEMM_MODE=0 selects the exact-fp32 MFMA"
                       if x6 else "fp32 MFMA (exact fmaf chains)"),
        "config": {"workload": f"ffhq-256 hfc_with_swav, {args.prototypes} prototypes, batch={args.batch}/GPU, "
                               f"N=20000 pixels x 5 patches, D=512, C=5376, 10 Sinkhorn iters, "
                               f"eps={args.eps if args.eps else (0.005 if args.pdf == 'uniform' else 0.003)}, "
                               + ("" if args.pdf == "uniform" else f"source_pdf={args.pdf} (NOT the headline configuration), ")
                               + ("deterministic=True (NOT the headline configuration), " if args.deterministic else "") +
                               f"T=0.01, sinkhorn_scope={args.scope}, pixel picks = the N smallest of H*W uniform keys drawn on "
                               f"the device generator (= the set randperm[:N] selects); views = the 13 feature taps of both "
                               f"perturbed latents (the RGB image, which enters no loss term, is not rendered); "
                               f"{sw.last_route.describe()}",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "generator": "StyleGAN2 256x256 random-init (seed 42)", "loss": loss},
        "roofline": roof,
    }
    # the same headline step on the bf16x6 forms (three bf16 planes, six products: every operand bit kept, fp32's range) — a
    # reader who does not accept f16x3's range contract as fp32 arithmetic finds the number of the engine that needs none here,
    # timed by the same command in the same process (GX_CONV_PLANES=0 GX_GEMM_PLANES=0 make it the headline)
    if x6 and world == 1 and args.all_configs != "off" and (ops.get_conv_planes() == 1 or ops.GEMM_H2):
        sw = None
        torch.cuda.empty_cache()
        prev_fmt, prev_h2 = ops.set_conv_planes(0), ops.GEMM_H2
        ops.GEMM_H2 = False
        try:
            swb = make(args.prototypes)
            dtb, stb, lossb = measure_nested(swb, args, 12, 3, dev, world)
            out["engine_bf16x6"] = nested_record(
                "the headline configuration with every contraction on the bf16x6 forms (ops.set_conv_planes(0), ops.GEMM_H2 = False); "
                + swb.last_route.describe(), args.batch, dtb, 12, 3, stb, loss=lossb)
            del swb
        finally:
            ops.set_conv_planes(prev_fmt)
            ops.GEMM_H2 = prev_h2
        torch.cuda.empty_cache()
    # the reference's own configuration (nprototypes = 5000): a second, short measurement inside the same line, so that
    # the number the reference actually ships is timed by whoever runs this script and not only by the builder
    want_rf = args.reference_faithful == "on" or (args.reference_faithful == "auto" and world == 1 and args.prototypes == 8)
    if want_rf:
        sw = None
        torch.cuda.empty_cache()
        rf_steps, rf_warm = 8, 2
        sw5 = make(5000)
        dt5, st5, loss5 = measure_nested(sw5, args, rf_steps, rf_warm, dev, world)
        ms5 = 1e3 * dt5 / rf_steps
        r5 = roofline_of(st5, ms5, None)
        peak = (PEAK_BF16_MFMA / (3.0 if ops.GEMM_H2 else 6.0) if x6 else PEAK_F32_MFMA)   # (the TN GEMM still runs bf16x6: conservative)
        gemm = st5.get("gemm_f32_kernel")
        hb = (r5 or {}).get("hbm")
        out["reference_faithful"] = {
            "workload": f"ffhq-256 hfc_with_swav, 5000 prototypes (hfc_with_swav_ffhq_config.py:52), batch={args.batch}/GPU, "
                        f"otherwise the configuration of this line; scores and d(scores) live on per-pixel tables, the three GEMMs run "
                        f"over pixels (5 x 20 000 samples are drawn from 65 536 pixels: --ds-table, DESIGN par. 4.15) and the "
                        f"step's Sinkhorn problems are solved on the tables, every row read once per pass (DESIGN par. 4.16)",
            "route": sw5.last_route.describe(),
            "prototypes": 5000, "steps": rf_steps, "warmup": rf_warm,
            "ms_per_step": ms5, "value": rf_steps * args.batch * world / dt5, "unit": "images/s", "loss": loss5,
            "roofline": {
                "kernel": "gx::gemm kernels (score NT, prototype-gradient TN and dz GEMMs over the pixels of both views, the "
                          "pyramid); FLOPs = those the launches execute",
                "mfma_tflops": None if not gemm else gemm["flops_per_step"] / (gemm["ms_per_step"] * 1e-3) / 1e12,
                "mfma_frac": None if not gemm else gemm["flops_per_step"] / (gemm["ms_per_step"] * 1e-3) / peak,
                "gemm_ms_per_step": None if not gemm else gemm["ms_per_step"],
                "step_tflop": None if not r5 else r5["step_tflop"], "step_frac": None if not r5 else r5["step_frac"],
                "hbm_kernel": None if not hb else hb["kernel"], "hbm_gbs": None if not hb else hb["achieved"],
                "hbm_frac": None if not hb else hb["frac"], "hbm_ms_per_step": None if not hb else hb["ms_per_step"],
                "ms_per_step_by_kernel": None if not r5 else r5["ms_per_step_by_kernel"]}}
        del sw5
        torch.cuda.empty_cache()
    # every other BASELINE.json configuration at its per-rank shape, a few steps each, in the same line
    want_all = args.all_configs == "on" or (args.all_configs == "auto" and world == 1 and args.prototypes == 8
                                            and args.pdf == "uniform")
    if want_all and world == 1:
        class _A:                                                        # `measure` reads the batch from here
            pass
        sw = None
        torch.cuda.empty_cache()
        # configs[2] cat-256 per rank: 16 images, 16 prototypes, source_pdf='image', eps 0.003 (hfc_with_swav_cat_config.py:52-65)
        a = _A(); a.batch = 16; a.overlap = args.overlap
        swc = make(16, pdf="image", eps=0.003)
        dtc, stc, lossc = measure_nested(swc, a, 8, 2, dev, 1)
        out["cat_per_rank"] = nested_record(
            "cat-256 hfc_with_swav per-rank shape of BASELINE.json configs[2] (batch 64 over 4 GPUs = 16 images per GPU): 16 "
            "prototypes, source_pdf='image' (marginals = histograms of the norm maps), eps=0.003, N=20000 x 5 patches; "
            "same 256x256 StyleGAN2 widths as ffhq; " + swc.last_route.describe(), 16, dtc, 8, 2, stc, loss=lossc)
        del swc
        torch.cuda.empty_cache()
        # configs[3] car-512 per rank: TRUE 512^2 generator (15 taps, hlen 5376 of 5504 channels), 4 images per GPU, eps 0.01;
        # BASELINE's 16 prototypes and the reference's own 4000 (hfc_with_swav_car_config.py:52, :65)
        g512 = build_generator(512, dev)
        a = _A(); a.batch = 4; a.overlap = args.overlap
        for kk, key in ((16, "car512_per_rank"), (4000, "car512_per_rank_k4000")):
            swk = make(kk, gen=g512, batch=4, size=512, eps=0.01, nlayers=7)
            st_, wu_ = (8, 2) if kk == 16 else (5, 2)
            dtk, stk, lossk = measure_nested(swk, a, st_, wu_, dev, 1)
            out[key] = nested_record(
                f"car-512 hfc_with_swav per-rank shape of BASELINE.json configs[3] (batch 32 over 8 GPUs = 4 images per GPU): true "
                f"512x512 StyleGAN2 generator (15 taps; hlen=5376 drops the two native-512 taps), {kk} prototypes"
                f"{' (hfc_with_swav_car_config.py:52)' if kk == 4000 else ''}, eps=0.01, uniform marginals, N=20000 of 262144 "
                f"pixels x 5 patches; " + swk.last_route.describe(), 4, dtk, st_, wu_, stk, loss=lossk)
            del swk
            torch.cuda.empty_cache()
        del g512
        torch.cuda.empty_cache()
        out["simclr"] = simclr_record(g, dev)                            # configs[4]
        out["predict"] = predict_record(g, dev)                          # evaluate.py / run_tests
        torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.prototypes)
            if want_rf:     # the reference's own configuration gets its CPU figure too (BASELINE.md par. 3: K = 5000)
                out["reference_faithful"]["cpu_baseline"] = cpu_baseline(5000)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
