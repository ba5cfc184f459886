"""N>1 control flow on CPU: world_size-2 gloo processes drive the product's sharding protocol
(ganecdotes_amd/sharded.py) with a torch stand-in for the per-rank HIP kernels; the result must
equal the 1-rank oracle on the concatenated pixels (SURVEY.md par. 8e)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ganecdotes_amd import sharded


class TorchSinkhornKernels:
    """CPU stand-in with the interface sharded.run_sinkhorn expects (same maths as
    csrc/gx_swav.hip: colmax / first pass / sweep / update in the log domain)."""

    def __init__(self, scores, eps, n_global):
        self.S, self.eps, self.K = scores, eps, scores.shape[1]
        self.cmax = None
        self.la = torch.zeros(self.K)
        self.u = torch.zeros(self.K)
        self.logc = float(np.log(1.0 / n_global))

    def colmax(self):
        self.cmax = self.S.max(0)[0]

    def cmax_get(self, out=None):
        return self.cmax.clone()

    def cmax_set(self, cm):
        self.cmax = cm

    def sweep(self, first):
        if first:
            self.u.copy_(torch.exp((self.S - self.cmax) / self.eps).sum(0))
        else:
            w = self.S / self.eps + self.la
            lb = self.logc - torch.logsumexp(w, dim=1, keepdim=True)
            self.u.copy_(torch.exp(w + lb).sum(0))

    def colsums(self):
        return self.u

    def update(self, first, u=None):
        u = self.u if u is None else u
        base = -self.cmax / self.eps if first else self.la
        self.la = base + float(np.log(1.0 / self.K)) - torch.log(u)

    def codes(self):
        return torch.softmax(self.S / self.eps + self.la, dim=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scores, eps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharded.shard_range(scores.shape[0], rank, world)
    k = TorchSinkhornKernels(scores[lo:hi].clone(), eps, scores.shape[0])
    sharded.run_sinkhorn(k, 10, sharded.make_reduce(dist.group.WORLD))
    q = k.codes()
    # gradient message: every rank contributes rank+1 -> sum = 3 everywhere
    g = [torch.full((3, 4), float(rank + 1)), torch.full((5,), float(rank + 1)), torch.tensor([float(rank + 1)])]
    sharded.allreduce_flat(g, dist.group.WORLD)
    assert all(bool((t == 3.0).all()) for t in g)
    out[rank] = (lo, hi, q.numpy())
    dist.barrier()
    dist.destroy_process_group()


class CountingReduce:
    def __init__(self, group):
        self.fn, self.calls = sharded.make_reduce(group), 0

    def __call__(self, t, op):
        self.calls += 1
        self.fn(t, op)


def _group_worker(rank, world, port, problems, eps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharded.shard_range(problems[0].shape[0], rank, world)
    ks = [TorchSinkhornKernels(sc[lo:hi].clone(), eps, sc.shape[0]) for sc in problems]
    red = CountingReduce(dist.group.WORLD)
    sharded.run_sinkhorn_group(ks, 10, red)
    out[rank] = (lo, hi, [k.codes().numpy() for k in ks], red.calls)
    dist.barrier()
    dist.destroy_process_group()


def test_grouped_sharded_sinkhorn_shares_the_exchanges():
    """A step's independent problems (views x patches) exchange together: 1 + niters
    collectives for the whole group, each problem's codes equal to its own 1-rank run."""
    from oracle import swav_ref as S
    torch.manual_seed(6)
    n, k, eps = 257, 8, 0.02
    problems = [0.3 * torch.randn(n, k) for _ in range(3)]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_group_worker, args=(2, _free_port(), problems, eps, out), nprocs=2, join=True)
    assert out[0][3] == 11 and out[1][3] == 11
    for i, sc in enumerate(problems):
        ref = S.sinkhorn_knopp(sc, eps, 10, fp64=True).numpy()
        q = np.zeros_like(ref)
        for r in range(2):
            lo, hi, qs, _ = out[r]
            q[lo:hi] = qs[i]
        assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))


class TorchTableKernels:
    """CPU stand-in with the interface sharded.run_sinkhorn_table expects (the maths of csrc/gx_sktab.hip): `nprob` problems
    over this rank's rows; scores (nprob, rows, K).  The shifted sweeps exponentiate relative to the previous sweep's
    log-sum-exp + the largest step of la, exactly like the kernel, and check what the kernel relies on."""

    def __init__(self, scores, eps, n_global):
        self.S, self.eps = scores, eps
        self.nprob, _, self.K = scores.shape
        self.c = 1.0 / n_global
        self.cmax = self.u = self.la = self.lse = self.dmax = None
        self.shifted_sweeps = 0

    def first(self):
        self.cmax = self.S.max(1)[0]
        self.u = torch.exp((self.S - self.cmax[:, None]) / self.eps).sum(1)

    def rebase(self, cm):
        self.u *= torch.exp((self.cmax - cm) / self.eps)

    def update(self, first, cmax=None):
        d = float(np.log(1.0 / self.K)) - torch.log(self.u)
        self.la = (-(self.cmax if cmax is None else cmax) / self.eps if first else self.la) + d
        self.dmax = None if first else d.max(1)[0]                       # the step statistics (all-reduced sums: same on every rank)
        self.spread = None if first else (d.max(1)[0] - d.min(1)[0])

    def sweep(self, shifted=False):
        w = self.S / self.eps + self.la[:, None]
        if shifted:
            assert self.lse is not None and bool((self.spread < 64 * np.log(2.0)).all())
            sigma = self.lse + self.dmax[:, None]                        # known before the row is touched
            assert bool((w.max(2)[0] <= sigma + 1e-4).all())             # an upper bound of every row's new maximum
            e = torch.exp(w - sigma[..., None])
            tot = e.sum(2)
            self.shifted_sweeps += 1
        else:
            sigma = w.max(2)[0]
            e = torch.exp(w - sigma[..., None])
            tot = e.sum(2)
        self.lse = sigma + torch.log(tot)
        self.u = (e * (self.c / tot)[..., None]).sum(1)

    def codes(self):
        return torch.softmax(self.S / self.eps + self.la[:, None], dim=2)


def _table_worker(rank, world, port, problems, eps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharded.shard_range(problems.shape[1], rank, world)
    k = TorchTableKernels(problems[:, lo:hi].clone(), eps, problems.shape[1])
    red = CountingReduce(dist.group.WORLD)
    sharded.run_sinkhorn_table(k, 10, red)
    out[rank] = (lo, hi, k.codes().numpy(), red.calls, k.shifted_sweeps)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_table_sinkhorn_two_ranks_equals_oracle():
    """The table solver's exchange protocol (sharded.run_sinkhorn_table: first pass -> MAX -> re-base -> SUM -> update, then
    sweep -> SUM -> update, the later sweeps in the shifted form) under two gloo ranks: every problem's codes equal the
    reference's sinkhorn_knopp on all rows, with 1 + niters collectives for all problems together."""
    from oracle import swav_ref as S
    torch.manual_seed(9)
    n, kk, eps, nprob = 301, 12, 0.02, 4
    problems = 0.3 * torch.randn(nprob, n, kk)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_table_worker, args=(2, _free_port(), problems, eps, out), nprocs=2, join=True)
    for r in range(2):
        lo, hi, q, calls, nshift = out[r]
        assert calls == 11 and nshift == 8
        for i in range(nprob):
            ref = S.sinkhorn_knopp(problems[i], eps, 10, fp64=True).numpy()
            assert np.all(np.abs(q[i] - ref[lo:hi]) <= 1e-3 * ref[lo:hi].max(axis=1, keepdims=True))
    one = TorchTableKernels(problems.clone(), eps, n)
    sharded.run_sinkhorn_table(one, 10, None)
    np.testing.assert_allclose(np.concatenate([out[0][2], out[1][2]], 1), one.codes().numpy(), rtol=1e-4, atol=1e-7)


def test_eight_ranks_uneven_shards_group_and_table_solvers_equal_the_oracle():
    """The shape of the real machine, rehearsed on CPU: world_size 8 over gloo, 100 003 rows (not a
    multiple of 8: shards of 12 501 and 12 500 rows), a step's worth of problems through BOTH exchange protocols —
    sharded.run_sinkhorn_group (per-problem kernels) and sharded.run_sinkhorn_table (the table solver) — against the
    reference's sinkhorn_knopp on the concatenation; 11 collectives per protocol for all problems together, whatever the
    number of ranks."""
    from oracle import swav_ref as S
    torch.manual_seed(11)
    n, kk, eps, nprob, world = 100003, 16, 0.02, 4, 8
    problems = 0.3 * torch.randn(nprob, n, kk)
    mgr = mp.Manager()
    out_g, out_t = mgr.dict(), mgr.dict()
    mp.spawn(_group_worker, args=(world, _free_port(), [problems[i] for i in range(nprob)], eps, out_g), nprocs=world, join=True)
    mp.spawn(_table_worker, args=(world, _free_port(), problems, eps, out_t), nprocs=world, join=True)
    sizes = sorted(out_g[r][1] - out_g[r][0] for r in range(world))
    assert sizes[0] == 12500 and sizes[-1] == 12501 and sum(sizes) == n
    for i in range(nprob):
        ref = S.sinkhorn_knopp(problems[i], eps, 10, fp64=True).numpy()
        tol = 1e-3 * ref.max(axis=1, keepdims=True)
        qg, qt = np.zeros_like(ref), np.zeros_like(ref)
        for r in range(world):
            lo, hi, qs, calls = out_g[r]
            assert calls == 11
            qg[lo:hi] = qs[i]
            lo, hi, q, calls, nshift = out_t[r]
            assert calls == 11 and nshift == 8
            qt[lo:hi] = q[i]
        assert np.all(np.abs(qg - ref) <= tol) and np.all(np.abs(qt - ref) <= tol)
        # a transport plan over ALL ranks' rows: every prototype column carries 1 / K of the mass up to Sinkhorn's residual
        assert abs(float(qt.sum()) - n) < 1e-2 * n


def test_shard_range_covers_everything():
    for n in (1, 7, 16, 33):
        for w in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_sinkhorn_two_ranks_equals_oracle_on_concatenation():
    from oracle import swav_ref as S
    torch.manual_seed(5)
    n, k, eps = 301, 24, 0.02
    scores = 0.3 * torch.randn(n, k)
    ref = S.sinkhorn_knopp(scores, eps, 10, fp64=True).numpy()
    # single-rank run of the same protocol
    k1 = TorchSinkhornKernels(scores.clone(), eps, n)
    sharded.run_sinkhorn(k1, 10, None)
    q1 = k1.codes().numpy()
    assert np.all(np.abs(q1 - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, scores, eps, out), nprocs=2, join=True)
    q = np.zeros_like(ref)
    for r in range(2):
        lo, hi, qr = out[r]
        q[lo:hi] = qr
    assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))
    np.testing.assert_allclose(q, q1, rtol=2e-4, atol=1e-7)


# ----------------------------------------------------------------------------- bench.py's own launcher
def _bench(*args, env=None, timeout=120):
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], cwd=root, capture_output=True, text=True,
                       timeout=timeout, env=dict(os.environ, **(env or {})))
    return p, time.time() - t0


def test_bench_self_launch_rendezvous_two_ranks():
    """`python bench.py --gpus 2` starts its own two ranks (fresh child processes); with --dry-rendezvous they only join
    the gloo group, pass one barrier and leave: rc 0, ONE JSON line from rank 0, the other ranks' chatter on stderr."""
    import json
    p, _ = _bench("--gpus", "2", "--dry-rendezvous")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_rendezvous": True, "world": 2}


def test_bench_self_launch_rendezvous_eight_ranks():
    """the driver's largest launch, rehearsed: `bench.py --gpus 8 --backend gloo --dry-rendezvous` — eight fresh child processes
    join the group on 127.0.0.1, pass one barrier and leave; no GPU is touched"""
    import json
    p, _ = _bench("--gpus", "8", "--backend", "gloo", "--dry-rendezvous", timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_rendezvous": True, "world": 8}


def test_bench_self_launch_returns_fast_when_a_rank_dies():
    """A rank that dies during start-up must not leave its peers (and the parent) waiting in the rendezvous until the
    backend's timeout: the launcher polls every child, stops the others on the first non-zero exit and returns that
    code within seconds; the dead rank's stderr is relayed with its rank prefix."""
    p, dt = _bench("--gpus", "2", "--dry-rendezvous", env={"GX_BENCH_FAIL_RANK": "1"})
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert dt < 30.0, dt
    assert "[rank 1] rank 1: GX_BENCH_FAIL_RANK set" in p.stderr and "stopping the other ranks" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
