"""Multi-GPU protocol of the hot path (SURVEY.md par. 8e): one process per GPU, images sharded
over ranks, prototypes / projection replicated.

Backend-agnostic host logic: the collectives and their order live here; the per-rank compute
is supplied by a kernel object (HIP: `ops.Sinkhorn`; the gloo CPU tests plug in a torch
stand-in), so the N>1 control flow is exercised by world_size-2 tests without a GPU.

Exchange steps (everything else is rank-local):
  Sinkhorn, per call :  all_reduce(MAX) of the K column maxima, then per iteration one
                        all_reduce(SUM) of the K column sums (20 KB at K=5000; latency-bound); the problems of a step
                        exchange together (run_sinkhorn_group; on score tables run_sinkhorn_table)
  optimizer step     :  all_reduce(SUM) of {dWc, db, loss} (issued before the pyramid backward) and of
                        every pyramid level's column block of dWp as soon as it is produced, all
                        asynchronous: only the coarsest levels' message is not hidden behind GEMMs
"""
import torch


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def make_reduce(group):
    """reduce(tensor, 'max'|'sum') -> in-place all-reduce over `group` (RCCL on GPUs, gloo in tests)."""
    import torch.distributed as dist

    def reduce(t, op):
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=group)
    return reduce


def run_sinkhorn(k, niters, reduce=None):
    """Drive one log-domain Sinkhorn over a row-sharded score matrix.

    `k` provides: colmax() ; cmax_get() -> tensor(K) ; cmax_set(tensor) ; sweep(first: bool) ;
    colsums() -> tensor(K) view that update() reads ; update(first: bool).
    With reduce=None it is the single-rank algorithm (no exchange)."""
    k.colmax()
    if reduce is not None:
        cm = k.cmax_get()
        reduce(cm, "max")
        k.cmax_set(cm)
    for it in range(niters):
        k.sweep(it == 0)
        if reduce is not None:
            reduce(k.colsums(), "sum")
        k.update(it == 0)
    return k


def run_sinkhorn_group(ks, niters, reduce=None):
    """run_sinkhorn over several independent problems of one K at once: the exchanges of all
    problems travel together — ONE max all-reduce of len(ks)*K column maxima and `niters` sum
    all-reduces of len(ks)*K column sums (a step's 2 views x 5 patches: 11 collectives, not 110).
    Kernel objects additionally take cmax_get(out=row) and update(first, u=row)."""
    if reduce is None or len(ks) == 1:
        for k in ks:
            run_sinkhorn(k, niters, reduce)
        return ks
    for k in ks:
        k.colmax()
    cm = torch.stack([k.cmax_get() for k in ks])
    reduce(cm, "max")
    for i, k in enumerate(ks):
        k.cmax_set(cm[i])
    for it in range(niters):
        for k in ks:
            k.sweep(it == 0)
        u = torch.stack([k.colsums() for k in ks])
        reduce(u, "sum")
        for i, k in enumerate(ks):
            k.update(it == 0, u=u[i])
    return ks


def run_sinkhorn_table(k, niters, reduce=None):
    """Drive the table solver (every problem of a step on the per-pixel score tables, ops.SinkhornTable) over row-sharded
    tables: the problems' exchanges travel together — ONE max all-reduce of their column maxima, then `niters` sum all-reduces of
    their column sums (the first pass's sums are re-based on the all-reduced maxima first).

    `k` provides: first() (leaves k.cmax, k.u relative to it) ; rebase(cm) (k.u re-based on the maxima cm) ; update(first, cmax=)
    (new scaling vectors from k.u, and — first=False — the step statistics the next sweep may rely on) ; sweep(shifted) (k.u =
    this rank's column sums; shifted: the sweep before it ran on the same state).  The statistics come from all-reduced sums,
    so every rank takes the same form of every sweep.  With reduce=None it is the single-rank algorithm."""
    k.first()
    cm = k.cmax
    if reduce is not None:
        cm = k.cmax.clone()
        reduce(cm, "max")
        k.rebase(cm)
        reduce(k.u, "sum")
    k.update(True, cmax=cm)
    for it in range(1, niters):
        k.sweep(shifted=it > 1)
        if reduce is not None:
            reduce(k.u, "sum")
        k.update(False)
    return k


def allreduce_flat(tensors, group):
    """Sum-all-reduce a list of tensors as ONE message (gradients + loss: 21 MB at K=5000)."""
    import torch.distributed as dist
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, group=group)
    o = 0
    for t in tensors:
        t.copy_(flat[o:o + t.numel()].reshape(t.shape))
        o += t.numel()
    return tensors


def allreduce_flat_async(tensors, group):
    """allreduce_flat issued with async_op=True: the message travels on the backend's communication
    stream while the caller keeps queueing kernels; the returned closure makes the current stream wait for
    it and scatters the sums back."""
    import torch.distributed as dist
    flat = torch.cat([t.reshape(-1) for t in tensors])
    work = dist.all_reduce(flat, group=group, async_op=True)

    def finish():
        work.wait()
        o = 0
        for t in tensors:
            t.copy_(flat[o:o + t.numel()].reshape(t.shape))
            o += t.numel()
    return finish


def allreduce_cols_async(mat, cols, group):
    """Async sum-all-reduce of the column blocks `cols` = [(offset, width), ...] of a row-major matrix
    (the slices of dWp one pyramid level has just produced): staged into one contiguous message."""
    import torch.distributed as dist
    cols = sorted(set(cols))
    stage = torch.cat([mat[:, o:o + w] for o, w in cols], dim=1).contiguous()
    work = dist.all_reduce(stage, group=group, async_op=True)

    def finish():
        work.wait()
        c = 0
        for o, w in cols:
            mat[:, o:o + w].copy_(stage[:, c:c + w])
            c += w
    return finish
