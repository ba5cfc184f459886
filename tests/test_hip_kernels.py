"""Parity of each HIP kernel (through the C ABI) against the oracle / golden vectors.
Runs on a real MI355X only."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops as O
    O.lib()
    return O


@pytest.fixture(params=[0, 1], ids=["f32mfma", "bf16x6"])
def engine(request, ops):
    """both contraction engines: exact-fp32 MFMA and bf16x6 (the default)"""
    prev = ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(prev)


def dev(x):
    if isinstance(x, np.ndarray):
        x = T(x)
    return x.cuda().contiguous()


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (200, 72, 100), (1, 512, 512), (333, 129, 37), (64, 8, 516)])
def test_gemm_layouts(ops, engine, ta, tb, m, n, k):
    torch.manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(k, m) if ta else torch.randn(m, k)
    b = torch.randn(n, k) if tb else torch.randn(k, n)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    out = ops.gemm(dev(a), dev(b), ta, tb)
    assert rel_err(out, ref) < 2e-6


def test_gemm_asymmetric_identity(ops, engine):
    # A = I with an ASYMMETRIC B catches a transposed C write
    n = 160
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) * 1e-3
    out = ops.gemm(dev(torch.eye(n)), dev(b), False, False)
    assert torch.equal(out.cpu(), b)
    out = ops.gemm(dev(torch.eye(n)), dev(b), False, True)
    assert torch.equal(out.cpu(), b.t().contiguous())


def test_gemm_epilogue_and_splitk(ops, engine):
    torch.manual_seed(3)
    m, n, k = 300, 200, 5000
    a, b = torch.randn(k, m), torch.randn(k, n)
    rs, cb, c0 = torch.rand(m) + 0.5, torch.randn(n), torch.randn(m, n)
    ref = 0.7 * rs[:, None].double() * (a.t().double() @ b.double()) + cb.double() + 0.3 * c0.double()
    for sk in (1, 4, 7):
        out = dev(c0.clone())
        ops.gemm(dev(a), dev(b), True, False, out=out, alpha=0.7, beta=0.3, row_scale=dev(rs), col_bias=dev(cb),
                 split_k=sk)
        assert rel_err(out, ref) < 3e-6
    # matches the k-ordered fp32 chain closely enough to be compared with a torch fp32 matmul
    ref32 = (a.t() @ b)
    out = ops.gemm(dev(a), dev(b), True, False, split_k=1)
    assert rel_err(out, ref32) < 1e-5


def test_gemm_big_ffhq_shape(ops, engine):
    # ffhq projection shape at reduced rows: (N, 5376) x (512, 5376)^T
    torch.manual_seed(5)
    x = torch.randn(1000, 5376)
    w = torch.randn(512, 5376) / 73.0
    out = ops.gemm(dev(x), dev(w), False, True)
    assert rel_err(out, x.double() @ w.double().t()) < 3e-6


# ------------------------------------------------------------------------------------- GEMM on the bf16 cores
@pytest.fixture
def x6(ops):
    prev = ops.set_gemm_mode(1)
    yield
    ops.set_gemm_mode(prev)


@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (200, 72, 100), (1, 512, 512), (333, 129, 37), (64, 8, 516),
                                   (260, 260, 16), (129, 4, 8)])
def test_gemm_x6_layouts(ops, x6, ta, tb, m, n, k):
    """bf16x6 engine (three-way bf16 split, six MFMA products): every layout / ragged edge"""
    torch.manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(k, m) if ta else torch.randn(m, k)
    b = torch.randn(n, k) if tb else torch.randn(k, n)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    out = ops.gemm(dev(a), dev(b), ta, tb)
    assert rel_err(out, ref) < 2e-6


def test_gemm_random_shapes_both_engines(ops, engine):
    """40 random (layout, M, N, K, split, ld padding) cases per engine, incl. the wave-specialised
    form (M > 128, K >= 768) and ragged edges, against fp64"""
    rng = np.random.RandomState(1234 + engine)
    for case in range(40):
        ta, tb = bool(rng.randint(2)), bool(rng.randint(2))
        m = int(rng.choice([1, 7, 64, 129, 200, 257, 300, 513]))
        n = int(rng.choice([1, 4, 12, 100, 128, 130, 512]))
        k = int(rng.choice([3, 16, 100, 256, 800, 1030, 2049]))
        sk = int(rng.choice([1, 1, 2, 5]))
        pad_a, pad_b = int(rng.choice([0, 4, 5])), int(rng.choice([0, 4, 7]))
        a_full = torch.randn((k, m + pad_a) if ta else (m, k + pad_a))
        b_full = torch.randn((n, k + pad_b) if tb else (k, n + pad_b))
        a = a_full[:, :m] if ta else a_full[:, :k]
        b = b_full[:, :k] if tb else b_full[:, :n]
        ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
        ad, bd = dev(a_full), dev(b_full)
        out = ops.gemm(ad[:, :m] if ta else ad[:, :k], bd[:, :k] if tb else bd[:, :n], ta, tb, split_k=sk)
        assert out.shape == (m, n)
        assert rel_err(out, ref) < 5e-6, (case, ta, tb, m, n, k, sk, pad_a, pad_b)


def test_gemm_x6_accuracy(ops):
    """fp32-level error: against fp64 the bf16x6 engine is no worse than the exact-fp32 MFMA engine
    on the projection shape (K = 5376) and on badly scaled operands; A = I reproduces B bit for bit
    (h + m + l == x exactly); split-K, epilogue and the concatenated-A form work on it."""
    torch.manual_seed(21)
    x = torch.randn(1024, 5376)
    w = torch.randn(512, 5376) / 73.0
    ref = x.double() @ w.double().t()
    prev = ops.set_gemm_mode(0)
    try:
        e32 = rel_err(ops.gemm(dev(x), dev(w), False, True), ref)
        ops.set_gemm_mode(1)
        ex6 = rel_err(ops.gemm(dev(x), dev(w), False, True), ref)
        assert ex6 < 1.5 * e32 + 5e-8, (ex6, e32)
        # 12 orders of magnitude of dynamic range inside one reduction
        scale = torch.exp(torch.linspace(-14, 14, 5376))
        xs, ws = x * scale, w / scale
        ex6s = rel_err(ops.gemm(dev(xs), dev(ws), False, True), ref)
        ops.set_gemm_mode(0)
        e32s = rel_err(ops.gemm(dev(xs), dev(ws), False, True), ref)
        assert ex6s < 1.5 * e32s + 5e-8, (ex6s, e32s)
        ops.set_gemm_mode(1)
        n = 160
        b = torch.randn(n, n)
        assert torch.equal(ops.gemm(dev(torch.eye(n)), dev(b), False, False).cpu(), b)
        assert torch.equal(ops.gemm(dev(torch.eye(n)), dev(b), False, True).cpu(), b.t().contiguous())
        assert torch.equal(ops.gemm(dev(torch.eye(n)), dev(b), True, False).cpu(), b)
        # epilogue + split-K
        m, nn, k = 300, 200, 5000
        a, bb = torch.randn(k, m), torch.randn(k, nn)
        rs, cb, c0 = torch.rand(m) + 0.5, torch.randn(nn), torch.randn(m, nn)
        want = 0.7 * rs[:, None].double() * (a.t().double() @ bb.double()) + cb.double() + 0.3 * c0.double()
        for sk in (1, 4, 7):
            out = dev(c0.clone())
            ops.gemm(dev(a), dev(bb), True, False, out=out, alpha=0.7, beta=0.3, row_scale=dev(rs), col_bias=dev(cb),
                     split_k=sk)
            assert rel_err(out, want) < 3e-6
        # [A1 | A2] concatenated along K
        a1, a2, wc = torch.randn(500, 128), torch.randn(500, 96), torch.randn(64, 224)
        out = torch.empty(500, 64, device="cuda")
        ops.gemm_cat(dev(a1), dev(a2), dev(wc), out)
        assert rel_err(out, torch.cat([a1, a2], 1).double() @ wc.double().t()) < 2e-6
    finally:
        ops.set_gemm_mode(prev)


# ------------------------------------------------------------------------------------- Sinkhorn
def _sk_cases():
    d = load_golden("gv7_sinkhorn")
    return d


@pytest.mark.parametrize("tag", ["uniform_e05", "uniform_e005"])
def test_sinkhorn_golden_uniform(ops, tag):
    d = _sk_cases()
    s = T(d["scores"]) * float(d[f"scale_{tag}"])
    sk = ops.Sinkhorn(dev(s), float(d[f"eps_{tag}"]), int(d["niters"])).run()
    q = sk.codes().cpu().numpy()
    ref = d[f"q_{tag}"]
    assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))
    np.testing.assert_allclose(q.sum(1), 1.0, rtol=1e-5)


def test_sinkhorn_golden_big_and_overflow(ops):
    from oracle import swav_ref as S
    d = _sk_cases()
    sk = ops.Sinkhorn(dev(d["scores_big"]), 0.01, 10).run()
    q = sk.codes().cpu().numpy()
    ref = d["q_big_uniform_e01"]
    assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))
    # where the reference's fp32 exp() overflows, the fp64 run of the same algorithm is the checker
    s = T(d["scores"]) * float(d["overflow_scale"])
    ref = S.sinkhorn_knopp(s, 0.005, 10, fp64=True).numpy()
    q = ops.Sinkhorn(dev(s), 0.005, 10).run().codes().cpu().numpy()
    assert np.isfinite(q).all()
    assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))


@pytest.mark.parametrize("tag", ["image_e05", "image_e003"])
def test_sinkhorn_image_pdf(ops, tag):
    from oracle import swav_ref as S
    d = _sk_cases()
    s = T(d["scores"]) * float(d[f"scale_{tag}"])
    img = T(d["img"])
    n, k = s.shape
    r = ops.hist_marginal(dev(img), k)
    c = ops.hist_marginal(dev(img), n)
    rk, cb = S.image_marginals(img, k, n)
    np.testing.assert_allclose(r.cpu().numpy(), rk.numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(c.cpu().numpy(), cb.numpy(), rtol=1e-5, atol=1e-9)
    q = ops.Sinkhorn(dev(s), float(d[f"eps_{tag}"]), 10, r=r, c=c).run().codes().cpu().numpy()
    ref = d[f"q_{tag}"]
    assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))


@pytest.mark.parametrize("bins", [16, 4096, 4097])
def test_hist_marginal_counts_at_norm_map_size(ops, bins):
    """gx_hist_marginal on a norm map of the bench's size (a million values) against torch.histc + the reference's fix-ups
    (swav_clustering.py:524-530): few bins (counted per block in LDS: the cat configuration's 16) and many (global atomics)"""
    torch.manual_seed(bins)
    x = torch.rand(1_000_003) * 3.0 + torch.randn(1_000_003).abs()
    h = torch.histc(x, bins) + 1e-9
    h[0] = h[1]
    want = (h / h.sum()).numpy()
    got = ops.hist_marginal(dev(x), bins).cpu().numpy()
    # a value within an ulp of a bin edge may fall on either side (the kernel divides where torch multiplies by a reciprocal):
    # about n * bins * 2^-23 such values, each moving 1 / n between two neighbouring bins
    assert np.abs(got - want).sum() < 2e-5 + 1e-6 * bins and np.abs(got - want).max() < 6e-6


@pytest.mark.parametrize("bins", [4097, 8192, 8193, 320000])
def test_hist_counts_many_bins_exact(ops, bins):
    """the range-owning histogram (a block counts a range of 8192 bins in LDS while it scans all values, plain stores) against
    torch.bincount of the same fp32 bin arithmetic, int((x - lo) / (hi - lo) * bins): exact counts, odd lengths, unaligned views"""
    torch.manual_seed(bins)
    for n, off in ((1_048_576, 0), (100_003, 1)):
        xb = dev(torch.rand(n + off) * 2.0 + torch.randn(n + off).abs())
        x = xb[off:]
        lo, hi = x.min(), x.max()
        pos = ((x - lo) / (hi - lo) * float(bins)).to(torch.int64).clamp_(0, bins - 1)
        want = torch.bincount(pos, minlength=bins).double() + 1e-9
        want[0] = want[1]
        want = want / want.sum()
        got = ops.hist_marginal(x.contiguous() if off == 0 else x, bins).double()
        assert float((got - want).abs().max()) <= 1e-6 * float(want.max())


def test_hist_marginal_many_bins_two_launch_normalisation(ops):
    """one bin per sample (320 000 at the bench batch): gx_hist_marginal normalises in two launches over all CUs; the staged form
    (gx_hist_range / _counts / _finish: one block) on the same data gives the same marginal, and it sums to one"""
    torch.manual_seed(5)
    x = dev(torch.rand(1_048_576) * 2.0 + torch.randn(1_048_576).abs())
    for bins in (16384, 320000):
        a = ops.hist_marginal(x, bins)
        b = ops.hist_marginal(x, bins, reduce=lambda t, op: None)
        assert abs(float(a.double().sum()) - 1.0) < 1e-5
        assert float((a - b).abs().max()) <= 1e-6 * float(b.max())
        assert float(a[0]) == float(a[1])


def test_hist_marginal_sharded_equals_single(ops):
    torch.manual_seed(14)
    x = torch.rand(4, 32, 32) * 5 + 0.3
    whole = ops.hist_marginal(dev(x), 300).cpu()
    parts = [dev(x[:2]), dev(x[2:])]
    import threading
    barrier, slots, outs = threading.Barrier(2), [None, None], [None, None]

    def run(rank):
        def reduce(t, op):
            torch.cuda.synchronize()
            slots[rank] = t.clone()
            barrier.wait()
            st = torch.stack(slots)
            red = st.max(0)[0] if op == "max" else st.sum(0)
            barrier.wait()
            t.copy_(red)
        outs[rank] = ops.hist_marginal(parts[rank], 300, reduce).cpu()
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert torch.equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0].numpy(), whole.numpy(), rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("n,k,eps", [(2000, 8, 0.005), (777, 16, 0.003), (500, 1000, 0.01), (64, 5000, 0.005), (67, 8000, 0.005),
                                     (301, 6148, 0.01), (131, 6000, 0.005)])
def test_sinkhorn_vs_oracle_shapes(ops, n, k, eps):
    from oracle import swav_ref as S
    torch.manual_seed(n + k)
    z = torch.nn.functional.normalize(torch.randn(n, 32), dim=1)
    w = torch.nn.functional.normalize(torch.randn(k, 32), dim=1)
    s = 0.4 * z @ w.t()
    ref = S.sinkhorn_knopp(s, eps, 10, fp64=True).numpy()
    q = ops.Sinkhorn(dev(s), eps, 10).run().codes().cpu().numpy()
    assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))
    # argmax agreement wherever the reference's top-2 gap is not a numerical tie
    top2 = np.sort(ref, axis=1)[:, -2:]
    decided = (top2[:, 1] - top2[:, 0]) > 1e-2 * top2[:, 1]
    assert (q.argmax(1) == ref.argmax(1))[decided].all()


@pytest.mark.parametrize("n,k,eps,pdf", [(5000, 8, 0.005, "uniform"), (9000, 40, 0.01, "uniform"),
                                         (4100, 16, 0.003, "image"), (320000, 8, 0.005, "uniform"),
                                         (5001, 12, 0.005, "uniform"), (4500, 4, 0.01, "image"), (4097, 6, 0.01, "uniform")])
def test_sinkhorn_small_fused_path(ops, n, k, eps, pdf):
    """rows >= 4096 and K <= 64 take gx_sinkhorn_small (one launch per iteration): same codes as
    the generic 3-kernel sequence and as the oracle"""
    from oracle import swav_ref as S
    torch.manual_seed(n + k)
    s = 0.3 * torch.nn.functional.normalize(torch.randn(n, 16), dim=1) @ \
        torch.nn.functional.normalize(torch.randn(k, 16), dim=1).t()
    sd = dev(s)
    r = c = None
    img = torch.rand(1, 64, 64) * 2 + 0.1
    if pdf == "image":
        r, c = ops.hist_marginal(dev(img), k), ops.hist_marginal(dev(img), n)
    q = ops.Sinkhorn(sd, eps, 10, r=r, c=c).run().codes().cpu().numpy()
    generic = ops.Sinkhorn(sd, eps, 10, r=r, c=c, tag="skg")
    from ganecdotes_amd.sharded import run_sinkhorn
    run_sinkhorn(generic, 10, None)
    qg = generic.codes().cpu().numpy()
    np.testing.assert_allclose(q, qg, rtol=2e-4, atol=1e-7)
    if n <= 10000:
        ref = S.sinkhorn_knopp(s, eps, 10, pdf, img, fp64=True).numpy()
        assert np.all(np.abs(q - ref) <= 1e-3 * ref.max(axis=1, keepdims=True))


@pytest.mark.parametrize("nprob,n,k,pdf", [(10, 20000, 8, "uniform"), (3, 4100, 16, "image"), (4, 5000, 40, "uniform"),
                                           (2, 300, 12, "uniform"), (2, 4200, 80, "uniform"),
                                           (20, 703, 300, "uniform"), (17, 1030, 5000, "uniform"), (16, 400, 132, "image"),
                                           (16, 333, 8000, "uniform")])
def test_sinkhorn_batch_equals_per_problem(ops, nprob, n, k, pdf):
    """SinkhornBatch (one launch per iteration over all problems when K <= 64, rows >= 4096, and — many prototypes,
    >= 16 problems, the per-image scope at K = 5000 — gx_sinkhorn_bigk_batch; per-problem kernels otherwise) == each
    problem solved alone; and with a (faked 2-rank)
    reduce the grouped exchange gives the 1-rank codes of the concatenation."""
    torch.manual_seed(nprob * n + k)
    eps = 0.005
    s = dev(0.3 * torch.nn.functional.normalize(torch.randn(nprob, n, 16), dim=2) @
            torch.nn.functional.normalize(torch.randn(k, 16), dim=1).t())
    r = c = None
    if pdf == "image":
        imgs = [dev(torch.rand(1, 64, 64) * 2 + 0.1) for _ in range(nprob)]
        r = torch.stack([ops.hist_marginal(im, k) for im in imgs])
        c = torch.stack([ops.hist_marginal(im, n) for im in imgs])
    skb = ops.SinkhornBatch(s, eps, 10, r=r, c=c).run()
    for i in range(nprob):
        one = ops.Sinkhorn(s[i], eps, 10, r=None if r is None else r[i], c=None if c is None else c[i],
                           tag="one").run()
        np.testing.assert_allclose(skb[i].codes().cpu().numpy(), one.codes().cpu().numpy(), rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(skb.la[i].cpu().numpy(), one.la.cpu().numpy(), rtol=1e-5, atol=1e-5)
    if pdf == "uniform":
        # two "ranks" hold the two halves of every problem; the fake reduce sums / maxes the pair
        h = n // 2
        halves = [ops.SinkhornBatch(s[:, :h].contiguous(), eps, 10, c_uniform=1.0 / n, tag="ha"),
                  ops.SinkhornBatch(s[:, h:].contiguous(), eps, 10, c_uniform=1.0 / n, tag="hb")]
        calls = []

        def lockstep():
            """drive both rank states through run_sinkhorn_group's schedule with a shared exchange"""
            A, Bk = halves[0].probs, halves[1].probs
            for kk in A + Bk:
                kk.colmax()
            cm = torch.maximum(torch.stack([kk.cmax_get() for kk in A]), torch.stack([kk.cmax_get() for kk in Bk]))
            calls.append("max")
            for i2 in range(nprob):
                A[i2].cmax_set(cm[i2]); Bk[i2].cmax_set(cm[i2])
            for it in range(10):
                for kk in A + Bk:
                    kk.sweep(it == 0)
                u = torch.stack([kk.colsums() for kk in A]) + torch.stack([kk.colsums() for kk in Bk])
                calls.append("sum")
                for i2 in range(nprob):
                    A[i2].update(it == 0, u=u[i2]); Bk[i2].update(it == 0, u=u[i2])
        lockstep()
        assert len(calls) == 11
        for i in range(nprob):
            q = torch.cat([halves[0][i].codes(), halves[1][i].codes()]).cpu().numpy()
            np.testing.assert_allclose(q, skb[i].codes().cpu().numpy(), rtol=5e-4, atol=1e-7)


@pytest.mark.parametrize("nprob,n,k,pdf", [(10, 20000, 8, "uniform"), (4, 9000, 16, "image"), (2, 600, 12, "uniform")])
def test_sinkhorn_batch_two_ranks_equal_one(ops, nprob, n, k, pdf):
    """SinkhornBatch.run(reduce) on two row-shards (two threads standing in for two ranks, the
    reduce exchanging through host memory) == the 1-rank batch on the concatenation; the batched
    sharded path makes 1 + niters exchanges for the whole batch."""
    import threading
    torch.manual_seed(nprob + n + k)
    eps = 0.005
    s = dev(0.3 * torch.nn.functional.normalize(torch.randn(nprob, n, 16), dim=2) @
            torch.nn.functional.normalize(torch.randn(k, 16), dim=1).t())
    r = c = None
    if pdf == "image":
        imgs = [dev(torch.rand(1, 64, 64) * 2 + 0.1) for _ in range(nprob)]
        r = torch.stack([ops.hist_marginal(im, k) for im in imgs])
        c = torch.stack([ops.hist_marginal(im, n) for im in imgs])
    whole = ops.SinkhornBatch(s, eps, 10, r=r, c=c, tag="whole").run()
    h = n // 2
    spans = [(0, h), (h, n)]
    barrier, slots, outs, calls = threading.Barrier(2), [None, None], [None, None], [0, 0]

    def run(rank):
        lo, hi = spans[rank]
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            def reduce(t, op):
                calls[rank] += 1
                torch.cuda.current_stream().synchronize()
                slots[rank] = t.clone()
                barrier.wait()
                st = torch.stack(slots)
                red = st.max(0)[0] if op == "max" else st.sum(0)
                barrier.wait()
                t.copy_(red)
            part = ops.SinkhornBatch(s[:, lo:hi].contiguous(), eps, 10, r=r, c=None if c is None else c[:, lo:hi].contiguous(),
                                     c_uniform=1.0 / n, tag=f"rank{rank}").run(reduce)
            outs[rank] = [part[i].codes().cpu() for i in range(nprob)]
            stream.synchronize()
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert calls == [11, 11]
    for i in range(nprob):
        q = torch.cat([outs[0][i], outs[1][i]]).numpy()
        np.testing.assert_allclose(q, whole[i].codes().cpu().numpy(), rtol=5e-4, atol=1e-7)


def test_sinkhorn_sharded_equals_single(ops):
    """W-rank result == 1-rank result on the concatenated pixels (ranks faked)."""
    torch.manual_seed(11)
    n, k, eps = 1200, 24, 0.01
    s = dev(0.3 * torch.randn(n, k))
    single = ops.Sinkhorn(s, eps, 10).run()
    shards = [s[i * 300:(i + 1) * 300].contiguous() for i in range(4)]
    states = [ops.Sinkhorn(x, eps, 10, c_uniform=1.0 / n, tag=f"sk{i}") for i, x in enumerate(shards)]
    # emulate the collectives: run the rank loops in lock-step through the reduce hook
    import threading
    nr = len(states)
    barrier = threading.Barrier(nr)
    slots = [None] * nr

    def make_reduce(rank):
        def reduce(t, op):
            torch.cuda.synchronize()
            slots[rank] = t.clone()
            barrier.wait()
            st = torch.stack(slots)
            red = st.max(0)[0] if op == "max" else st.sum(0)
            barrier.wait()
            t.copy_(red)
        return reduce

    ths = [threading.Thread(target=lambda i=i: states[i].run(make_reduce(i))) for i in range(nr)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    q = torch.cat([x.codes() for x in states]).cpu().numpy()
    ref = single.codes().cpu().numpy()
    assert np.all(np.abs(q - ref) <= 1e-4 * ref.max(axis=1, keepdims=True))


# ------------------------------------------------------------------------------------- CE + grads
def test_swapped_ce_golden_gv8(ops):
    d = load_golden("gv8_loss_grads")
    ss, st = dev(d["scores_s"]), dev(d["scores_t"])
    n, k = ss.shape
    eps, tt = float(d["eps"]), float(d["temperature"])
    sk_s = ops.Sinkhorn(ss, eps, 10).run()
    sk_t = ops.Sinkhorn(st, eps, 10, tag="sk2").run()
    loss = torch.zeros(1, device="cuda")
    ds, dt = torch.empty_like(ss), torch.empty_like(st)
    ops.swapped_ce(sk_s, sk_t, tt, 1.0 / n, loss, 1.0 / n, 0.0, ds, dt)
    np.testing.assert_allclose(loss.item(), float(d["loss"]), rtol=1e-4)
    # d(scores) against autograd of the oracle loss with the GOLDEN codes
    from oracle import swav_ref as S
    s1 = T(d["scores_s"]).clone().requires_grad_(True)
    t1 = T(d["scores_t"]).clone().requires_grad_(True)
    S.swapped_prediction_loss(s1 / tt, t1 / tt, T(d["q_s"]), T(d["q_t"])).backward()
    assert rel_err(ds, s1.grad) < 1e-3
    assert rel_err(dt, t1.grad) < 1e-3


@pytest.mark.parametrize("n,k", [(1000, 8), (300, 700), (131, 8000)])
def test_swapped_ce_vs_oracle(ops, n, k):
    from oracle import swav_ref as S
    torch.manual_seed(n)
    s = 0.3 * torch.randn(n, k)
    t = s + 0.1 * torch.randn(n, k)
    eps, tt = 0.01, 0.05
    qs = S.sinkhorn_knopp(s, eps, 10, fp64=True)
    qt = S.sinkhorn_knopp(t, eps, 10, fp64=True)
    s1, t1 = s.clone().requires_grad_(True), t.clone().requires_grad_(True)
    ref = S.swapped_prediction_loss(s1 / tt, t1 / tt, qs, qt)
    ref.backward()
    sk_s = ops.Sinkhorn(dev(s), eps, 10).run()
    sk_t = ops.Sinkhorn(dev(t), eps, 10, tag="sk2").run()
    loss = torch.zeros(1, device="cuda")
    ds, dt = torch.empty(n, k, device="cuda"), torch.empty(n, k, device="cuda")
    ops.swapped_ce(sk_s, sk_t, tt, 1.0 / n, loss, 1.0 / n, 0.0, ds, dt)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-3)
    assert rel_err(ds, s1.grad) < 1e-3 and rel_err(dt, t1.grad) < 1e-3


@pytest.mark.parametrize("n,k", [(301, 700), (1030, 5000), (64, 68), (130, 6000)])
def test_swapped_ce_fused_many_prototypes(ops, n, k):
    """gx_swapped_ce_fused (one pass over both score matrices, bias gradient from the same pass): against the oracle's
    loss / autograd, and against the two-kernel gx_swapped_ce + gx_colsum on the same inputs; odd row counts, K not
    a multiple of the 1024-column register tile, accumulation into gb"""
    from oracle import swav_ref as S
    torch.manual_seed(n + k)
    s = 0.3 * torch.randn(n, k)
    t = s + 0.1 * torch.randn(n, k)
    eps, tt = 0.01, 0.05
    qs = S.sinkhorn_knopp(s, eps, 10, fp64=True)
    qt = S.sinkhorn_knopp(t, eps, 10, fp64=True)
    s1, t1 = s.clone().requires_grad_(True), t.clone().requires_grad_(True)
    ref = S.swapped_prediction_loss(s1 / tt, t1 / tt, qs, qt)
    ref.backward()
    sk_s = ops.Sinkhorn(dev(s), eps, 10).run()
    sk_t = ops.Sinkhorn(dev(t), eps, 10, tag="sk2").run()
    ds, dt = torch.empty(n, k, device="cuda"), torch.empty(n, k, device="cuda")
    assert ops.swapped_ce_fused_ok(sk_s, ds)
    loss = torch.zeros(1, device="cuda")
    gb0 = torch.randn(k)
    gb = dev(gb0.clone())
    ops.swapped_ce_fused(sk_s, sk_t, tt, 1.0 / n, loss, 1.0 / n, 0.0, ds, dt, gb, 1.0)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-3)
    assert rel_err(ds, s1.grad) < 1e-3 and rel_err(dt, t1.grad) < 1e-3
    want_gb = gb0.double() + s1.grad.double().sum(0) + t1.grad.double().sum(0)
    assert float((gb.cpu().double() - want_gb).abs().max()) < 1e-3 * float(s1.grad.abs().sum(0).max()) + 1e-6
    # the two-kernel form on the same Sinkhorn states
    ds2, dt2 = torch.empty_like(ds), torch.empty_like(dt)
    loss2 = torch.zeros(1, device="cuda")
    ops.swapped_ce(sk_s, sk_t, tt, 1.0 / n, loss2, 1.0 / n, 0.0, ds2, dt2)
    assert rel_err(ds, ds2) < 1e-5 and rel_err(dt, dt2) < 1e-5 and abs(loss.item() - loss2.item()) < 1e-5 * abs(loss2.item())
    gb2 = torch.empty(k, device="cuda")
    ops.colsum(ds2, gb2, beta=0.0)
    ops.colsum(dt2, gb2, beta=1.0)
    ops.swapped_ce_fused(sk_s, sk_t, tt, 1.0 / n, loss, 1.0 / n, 0.0, ds, dt, gb, 0.0)         # beta = 0 overwrites
    assert float((gb - gb2).abs().max()) < 1e-5 * float(gb2.abs().max()) + 1e-9
    with pytest.raises(RuntimeError):                                                          # K <= 64: not this kernel
        sk8 = ops.Sinkhorn(dev(0.3 * torch.randn(100, 8)), eps, 3).run()
        ops.swapped_ce_fused(sk8, sk8, tt, 1.0, loss, 1.0, 0.0, torch.empty(100, 8, device="cuda"),
                             torch.empty(100, 8, device="cuda"), torch.zeros(8, device="cuda"), 0.0)


# ------------------------------------------------------------------------------------- gather
def test_gather_golden_gv5(ops):
    d = load_golden("gv5_pixel_features")
    taps = [T(d[f"tap_{i}"]) for i in range(4)]
    ts = ops.TapSet([dev(t.permute(0, 2, 3, 1)) for t in taps])
    hw = 16 * 16
    picks = torch.randperm(hw, dtype=torch.int32).reshape(1, hw).cuda()
    for hlen, key in ((37, "stack"), (42, "stack_full")):
        x = ops.gather_taps(ts, picks, hlen).cpu()
        ref = T(d[key]).reshape(hlen, hw).t()[picks.cpu().long().reshape(-1)]
        assert torch.equal(x, ref)


def test_gather_rotflip_batch(ops):
    from oracle import swav_ref as S
    torch.manual_seed(9)
    b, hw = 3, 32
    taps = [torch.randn(b, 8, 4, 4), torch.randn(b, 12, 16, 16), torch.randn(b, 4, 32, 32)]
    ts = ops.TapSet([dev(t.permute(0, 2, 3, 1)) for t in taps])
    angles, flips = [7.3, -4.1, 0.0], [True, False, True]
    maps = torch.stack([ops.rotflip_map(a, f, hw, hw, "cuda") for a, f in zip(angles, flips)])
    picks = torch.stack([torch.randperm(hw * hw)[:200] for _ in range(b)]).int().cuda()
    x = ops.gather_taps(ts, picks, 24, maps).cpu().reshape(b, 200, 24)
    img = ops.norm_map(ts, 24, maps).cpu()
    for i in range(b):
        omap = S.rotflip_index_map(angles[i], flips[i], hw, hw)
        agree = (omap == maps[i].cpu().long()).float().mean()
        assert agree > 0.999                                  # nearest-rounding ties only
        stack = S.pixel_features([t[i:i + 1] for t in taps], 24)
        stack = S.apply_index_map(stack, maps[i].cpu().long())   # same map -> exact gather parity
        ref = S.sample_pixels(stack, picks[i].cpu().long(), 200)
        assert torch.equal(x[i], ref)
        np.testing.assert_allclose(img[i].numpy(), S.norm_map(stack)[0].numpy(), rtol=1e-5, atol=1e-6)


def test_gather_and_norm_map_bilinear_vs_torch_interpolate(ops):
    """swav_args['hf_interp'] = 'bilinear' (the `mode` of the reference's F.interpolate over the taps,
    hfc_with_swav/swav_clustering.py:123-124): gx_gather_taps(interp=1) and gx_stack_norm_map_bilinear against the stack torch
    itself makes (oracle pixel_features -> F.interpolate(mode='bilinear'), align_corners=False), with rotate / flip maps and an
    hlen that cuts a tap; taps at 4, 16 and the full 32 (a tap at full resolution is taken as it is)."""
    from oracle import swav_ref as S
    torch.manual_seed(19)
    b, hw, hlen = 3, 32, 22
    taps = [torch.randn(b, 8, 4, 4), torch.randn(b, 12, 16, 16), torch.randn(b, 4, 32, 32)]
    ts = ops.TapSet([dev(t.permute(0, 2, 3, 1)) for t in taps])
    angles, flips = [7.3, -4.1, 0.0], [True, False, True]
    maps = torch.stack([ops.rotflip_map(a, f, hw, hw, "cuda") for a, f in zip(angles, flips)])
    picks = torch.stack([torch.randperm(hw * hw)[:300] for _ in range(b)]).int().cuda()
    for use_maps in (True, False):
        m = maps if use_maps else None
        x = ops.gather_taps(ts, picks, hlen, m, interp="bilinear").cpu().reshape(b, 300, hlen)
        img = ops.norm_map(ts, hlen, m, interp="bilinear").cpu()
        for i in range(b):
            stack = S.pixel_features([t[i:i + 1] for t in taps], hlen, "bilinear")
            if use_maps:
                stack = S.apply_index_map(stack, maps[i].cpu().long())
            ref = S.sample_pixels(stack, picks[i].cpu().long(), 300)
            np.testing.assert_allclose(x[i].numpy(), ref.numpy(), rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(img[i].numpy(), S.norm_map(stack)[0].numpy(), rtol=1e-5, atol=1e-6)
    # and it is NOT the nearest stack
    xn = ops.gather_taps(ts, picks, hlen, None).cpu()
    assert float((xn - ops.gather_taps(ts, picks, hlen, None, interp="bilinear").cpu()).abs().max()) > 0.1


def test_gemm_cat_and_strided_views(ops, engine):
    torch.manual_seed(8)
    m, k1, k2, n = 300, 64, 40, 72
    a1, a2 = torch.randn(m, k1), torch.randn(m, k2 + 8)
    w = torch.randn(n, 500)
    out = torch.empty(m, n, device="cuda")
    wd = dev(w)
    ops.gemm_cat(dev(a1), dev(a2)[:, :k2], wd[:, 100:100 + k1 + k2], out)
    ref = torch.cat([a1, a2[:, :k2]], 1).double() @ w[:, 100:100 + k1 + k2].double().t()
    assert rel_err(out, ref) < 3e-6
    # TN with a column-sliced output (gradient block of the projection weight)
    g, t = torch.randn(700, 48), torch.randn(700, 96)
    gw = torch.zeros(48, 500, device="cuda")
    ops.gemm(dev(g), dev(t)[:, :80], True, False, out=gw[:, 200:280], split_k=3)
    assert rel_err(gw[:, 200:280], g.double().t() @ t[:, :80].double()) < 3e-6
    assert float(gw[:, :200].abs().max()) == 0 and float(gw[:, 280:].abs().max()) == 0


def test_gemm_upadd_fused_pyramid_level(ops, engine):
    """one pyramid level in one launch: [a1 | a2] . W^T + upsample2(coarse) (and the single-A / beta forms)"""
    torch.manual_seed(18)
    for b, rf, k1, k2, n in ((2, 8, 64, 40, 72), (3, 16, 32, 0, 256), (1, 4, 96, 32, 130)):
        m = b * rf * rf
        a1, a2 = torch.randn(m, k1), (torch.randn(m, k2 + 4) if k2 else None)
        w = torch.randn(n, k1 + k2 + 9)
        coarse = torch.randn(b, rf // 2, rf // 2, n)
        up = coarse.repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(m, n).double()
        acat = a1 if a2 is None else torch.cat([a1, a2[:, :k2]], 1)
        ref = acat.double() @ w[:, 5:5 + k1 + k2].double().t() + up
        out = torch.empty(m, n, device="cuda")
        ops.gemm_upadd(dev(a1), None if a2 is None else dev(a2)[:, :k2], dev(w)[:, 5:5 + k1 + k2], out,
                       dev(coarse.reshape(-1, n)), rf)
        assert rel_err(out, ref) < 3e-6
        prior = torch.randn(m, n)
        out2 = dev(prior.clone())
        ops.gemm_upadd(dev(a1), None if a2 is None else dev(a2)[:, :k2], dev(w)[:, 5:5 + k1 + k2], out2,
                       dev(coarse.reshape(-1, n)), rf, beta=1.0)
        assert rel_err(out2, ref + prior.double()) < 3e-6
    with pytest.raises(RuntimeError):                  # a non-power-of-two resolution has no shift form
        ops.gemm_upadd(dev(torch.randn(72, 32)), None, dev(torch.randn(8, 32)), torch.empty(72, 8, device="cuda"),
                       dev(torch.randn(18, 8)), 6)


def test_gemm_live_bounds_on_the_device(ops):
    """gx_gemm_f32_nt_planes_mlive / gx_gemm_f32_klive (DESIGN §4.15 d: the score tables hold the drawn pixels first and an
    int32 DEVICE scalar says how many): the live rows are BITWISE those of the unbounded launch over the same rows, the rows
    of whole dead 128-row tiles keep what they held, and the TN product equals the product over the live rows only —
    whatever garbage (here NaN) sits past the bound."""
    torch.manual_seed(61)
    prev = ops.set_gemm_mode(1)
    try:
        for (m, n, k, live_n) in ((4096, 260, 64, 2500), (1300, 512, 5008, 1300), (1024, 132, 32, 0), (900, 68, 48, 129)):
            a, w, cb = dev(torch.randn(m, k)), dev(torch.randn(n, k)), dev(torch.randn(n))
            bp = ops.split_gemm_weights(w)
            live = torch.tensor([live_n], dtype=torch.int32, device="cuda")
            full = torch.empty(m, n, device="cuda")
            ops.gemm_nt_planes(a, None, bp, full, col_bias=cb)
            a_dead = a.clone()
            a_dead[live_n:] = float("nan")                     # nothing past the bound may reach a live row
            out = torch.full((m, n), -7.0, device="cuda")
            ops.gemm_nt_planes_live(a_dead, bp, out, live, live_n / m, col_bias=cb)
            assert torch.equal(out[:live_n], full[:live_n]), (m, n, k, live_n)
            dead0 = -(-live_n // 128) * 128                    # rows of the tile the bound falls in are written (and unread)
            assert torch.equal(out[dead0:], torch.full_like(out[dead0:], -7.0))
            # TN: dS^T Z over the live rows, split-k over the LIVE range, accumulate form included
            ds, z = dev(torch.randn(m, n)), dev(torch.randn(m, 40))
            ds_dead, z_dead = ds.clone(), z.clone()
            ds_dead[live_n:] = float("nan")
            z_dead[live_n:] = float("nan")
            want = ds[:live_n].double().t() @ z[:live_n].double()
            for split in (1, 3, 8):
                prior = torch.randn(n, 40, device="cuda")
                got = prior.clone()
                ops.gemm(ds_dead, z_dead, True, False, out=got, beta=0.5, split_k=split, live=live, live_frac=live_n / m)
                assert rel_err(got, want + 0.5 * prior.double()) < 3e-6, (m, n, split, live_n)
                again = prior.clone()
                ops.gemm(ds_dead, z_dead, True, False, out=again, beta=0.5, split_k=split, live=live, live_frac=live_n / m)
                assert torch.equal(got, again)
        with pytest.raises(RuntimeError):                      # a host-side count is not what these entry points take
            ops.gemm_nt_planes_live(a, bp, out, torch.tensor([3], dtype=torch.int32), 1.0)
        with pytest.raises(RuntimeError):
            ops.gemm(a, w, False, True, live=live)                 # the k bound is the TN form's
    finally:
        ops.set_gemm_mode(prev)


def test_gemm_nt_planes_bitwise_and_fp64(ops):
    """gx_gemm_f32_nt_planes (weights pre-split once per step, both operands by LDS-DMA): against fp64 and BITWISE
    against gx_gemm_f32 / _cat / _upadd on the bf16x6 engine, incl. ragged M / N tails, strided views, every epilogue
    term; the exact-fp32 engine and unaligned shapes are refused loudly."""
    torch.manual_seed(28)
    prev = ops.set_gemm_mode(1)
    try:
        for (m, n, k1, k2, up) in ((300, 132, 32, 0, 0), (1024, 512, 64, 16, 8), (4160, 260, 128, 128, 0), (4096, 512, 256, 0, 16),
                                   (77, 8, 16, 0, 0)):
            a1 = torch.randn(m, k1 + 8)
            a2 = torch.randn(m, k2 + 4) if k2 else None
            w = torch.randn(n, k1 + k2 + 12)
            rs, cb = torch.rand(m) + 0.5, torch.randn(n)
            co = torch.randn(m // 4, n) if up else None
            a1d, a2d, wd = dev(a1)[:, :k1], (dev(a2)[:, :k2] if k2 else None), dev(w)[:, 4:4 + k1 + k2]
            bp = ops.split_gemm_weights(wd)
            acat = (a1[:, :k1] if not k2 else torch.cat([a1[:, :k1], a2[:, :k2]], 1)).double()
            base = acat @ w[:, 4:4 + k1 + k2].double().t()
            prior = torch.randn(m, n)
            ref, out = dev(prior.clone()), dev(prior.clone())
            if up:
                upd = co.reshape(-1, up // 2, up // 2, n).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(m, n).double()
                ops.gemm_upadd(a1d, a2d, wd, ref, dev(co), up)
                ops.gemm_nt_planes(a1d, a2d, bp, out, coarse=dev(co), fine_res=up)
                want = base + upd
            elif k2:
                ops.gemm_cat(a1d, a2d, wd, ref, alpha=0.7, beta=0.25)
                ops.gemm_nt_planes(a1d, a2d, bp, out, alpha=0.7, beta=0.25)
                want = 0.7 * base + 0.25 * prior.double()
            else:
                ops.gemm(a1d, wd, False, True, out=ref, alpha=1.3, row_scale=dev(rs), col_bias=dev(cb))
                ops.gemm_nt_planes(a1d, None, bp, out, alpha=1.3, row_scale=dev(rs), col_bias=dev(cb))
                want = 1.3 * rs.double()[:, None] * base + cb.double()[None, :]
            assert rel_err(out, want) < 3e-6
            assert torch.equal(out, ref), (m, n, k1, k2, up)
        # column maxima of the written matrix from the epilogue (the Sinkhorn stabiliser), ragged M and N
        for (m, n, k) in ((300, 132, 32), (4097, 5000, 64), (31, 68, 16)):
            a, w, cb = dev(torch.randn(m, k)), dev(torch.randn(n, k)), dev(torch.randn(n))
            out = torch.empty(m, n, device="cuda")
            _, cm = ops.gemm_nt_planes(a, None, ops.split_gemm_weights(w), out, col_bias=cb, want_colmax=True)
            assert torch.equal(cm, out.amax(0))
            sk1 = ops.Sinkhorn(out, 0.05, 5).run()
            sk2 = ops.Sinkhorn(out, 0.05, 5, tag="skp")
            sk2.cmax_preset = cm
            sk2.run()
            np.testing.assert_allclose(sk2.la.cpu().numpy(), sk1.la.cpu().numpy(), rtol=1e-5, atol=1e-5)
            # ... and the FIRST Sinkhorn pass from the same epilogue: u_k = sum_n exp((out_nk - max_k) / eps), merged from
            # per-32-row sums relative to their own maxima; the solver fed with both skips two sweeps of `out`
            for eps in (0.05, 0.005):
                _, cm2, uf = ops.gemm_nt_planes(a, None, ops.split_gemm_weights(w), out, col_bias=cb, want_colmax=True, sk_eps=eps)
                assert torch.equal(cm2, cm)
                want = torch.exp((out.double() - cm.double()) / eps).sum(0)
                assert rel_err(uf, want) < 2e-5
                sk0 = ops.Sinkhorn(out, eps, 5).run()
                sk3 = ops.Sinkhorn(out, eps, 5, tag="skf")
                sk3.cmax_preset, sk3.u_first_preset = cm2, uf
                sk3.run()
                np.testing.assert_allclose(sk3.la.cpu().numpy(), sk0.la.cpu().numpy(), rtol=1e-5, atol=2e-5)
        # per row group (the per-image Sinkhorn scope): groups of whole 32-row slices, and the batched solver fed with them
        g, rows, n, k = 16, 96, 132, 32
        a, w = dev(torch.randn(g * rows, k)), dev(0.3 * torch.randn(n, k))
        out = torch.empty(g * rows, n, device="cuda")
        _, cm = ops.gemm_nt_planes(a, None, ops.split_gemm_weights(w), out, want_colmax=g)
        assert torch.equal(cm, out.view(g, rows, n).amax(1))
        plain = ops.SinkhornBatch(out.view(g, rows, n), 0.05, 5).run()
        fed = ops.SinkhornBatch(out.view(g, rows, n), 0.05, 5, tag="skfed")
        for i in range(g):
            fed[i].cmax_preset = cm[i]
        fed.run()
        np.testing.assert_allclose(fed.la.cpu().numpy(), plain.la.cpu().numpy(), rtol=1e-5, atol=1e-5)
        _, cm2, uf = ops.gemm_nt_planes(a, None, ops.split_gemm_weights(w), out, want_colmax=g, sk_eps=0.05)
        assert torch.equal(cm2, cm)
        assert rel_err(uf, torch.exp((out.view(g, rows, n).double() - cm.double()[:, None]) / 0.05).sum(1)) < 2e-5
        fed2 = ops.SinkhornBatch(out.view(g, rows, n), 0.05, 5, tag="skfed2")
        for i in range(g):
            fed2[i].cmax_preset, fed2[i].u_first_preset = cm2[i], uf[i]
        fed2.run()
        np.testing.assert_allclose(fed2.la.cpu().numpy(), plain.la.cpu().numpy(), rtol=1e-5, atol=2e-5)
        a, w = dev(torch.randn(64, 24)), dev(torch.randn(16, 24))            # K % 16 != 0
        assert not ops.gemm_planes_ok(a, None, 16, torch.empty(64, 16, device="cuda"))
        with pytest.raises(RuntimeError):
            ops.gemm_nt_planes(a, None, ops.split_gemm_weights(w), torch.empty(64, 16, device="cuda"))
        ops.set_gemm_mode(0)                                                  # planes belong to the bf16x6 engine
        a, w = dev(torch.randn(64, 32)), dev(torch.randn(16, 32))
        assert not ops.gemm_planes_ok(a, None, 16, torch.empty(64, 16, device="cuda"))
        with pytest.raises(RuntimeError):
            ops.gemm_nt_planes(a, None, ops.split_gemm_weights(w), torch.empty(64, 16, device="cuda"))
    finally:
        ops.set_gemm_mode(prev)


def test_gemm_nt_planes_f16x3_form_is_fp32_class(ops):
    """The f16x3 form of gx_gemm_f32_nt_planes (two fp16 planes of the weights scaled by the power of two found on the device,
    the A elements scaled from the caller's bound a_range, three MFMA products instead of six): against fp64 its error is that
    of the bf16x6 form on the same data — unit-scale, tiny (gradient-like, 1e-6) and large (1e3) operands, weights of 0.01,
    every epilogue term, ragged tails, two A sources, the upsample-add, the column maxima / first Sinkhorn pass, the device-
    counted live rows; a bound 2^10 above the data keeps fp32-class error (2^20: not any more); an element past 4 x a_range comes out
    NaN (loud), a missing a_range is refused."""
    torch.manual_seed(29)
    prev = ops.set_gemm_mode(1)
    was = ops.GEMM_H2
    ops.GEMM_H2 = True
    try:
        for (m, n, k1, k2, up, sa, sw) in ((300, 132, 32, 0, 0, 1.0, 1.0), (1024, 512, 64, 16, 8, 1e-6, 0.01),
                                           (4160, 260, 128, 128, 0, 1e3, 3.0), (4096, 512, 256, 0, 16, 1.0, 0.0136),
                                           (77, 8, 16, 0, 0, 1.0, 1.0), (2048, 640, 5376, 0, 0, 1.0, 0.0136)):
            a1 = torch.randn(m, k1 + 8) * sa
            a2 = torch.randn(m, k2 + 4) * sa if k2 else None
            w = torch.randn(n, k1 + k2 + 12) * sw
            rs, cb = torch.rand(m) + 0.5, torch.randn(n) * sa * sw
            co = torch.randn(m // 4, n) * sa * sw if up else None
            a1d, a2d, wd = dev(a1)[:, :k1], (dev(a2)[:, :k2] if k2 else None), dev(w)[:, 4:4 + k1 + k2]
            bp6, bp3 = ops.split_gemm_weights(wd), ops.split_gemm_weights(wd, h2=True)
            assert bp3.h2 and not bp6.h2
            acat = (a1[:, :k1] if not k2 else torch.cat([a1[:, :k1], a2[:, :k2]], 1)).double()
            base = acat @ w[:, 4:4 + k1 + k2].double().t()
            prior = torch.randn(m, n) * sa * sw
            o6, o3 = dev(prior.clone()), dev(prior.clone())
            rng = float(acat.abs().max()) * 1.5
            if up:
                upd = co.reshape(-1, up // 2, up // 2, n).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(m, n).double()
                ops.gemm_nt_planes(a1d, a2d, bp6, o6, coarse=dev(co), fine_res=up)
                ops.gemm_nt_planes(a1d, a2d, bp3, o3, coarse=dev(co), fine_res=up, a_range=rng)
                want = base + upd
            elif k2:
                ops.gemm_nt_planes(a1d, a2d, bp6, o6, alpha=0.7, beta=0.25)
                ops.gemm_nt_planes(a1d, a2d, bp3, o3, alpha=0.7, beta=0.25, a_range=rng)
                want = 0.7 * base + 0.25 * prior.double()
            else:
                ops.gemm_nt_planes(a1d, None, bp6, o6, alpha=1.3, row_scale=dev(rs), col_bias=dev(cb))
                ops.gemm_nt_planes(a1d, None, bp3, o3, alpha=1.3, row_scale=dev(rs), col_bias=dev(cb), a_range=rng)
                want = 1.3 * rs.double()[:, None] * base + cb.double()[None, :]
            e6 = float((o6.double().cpu() - want).norm() / want.norm())
            e3 = float((o3.double().cpu() - want).norm() / want.norm())
            assert e3 < 1.5 * e6 + 1e-7, (m, n, k1, k2, e3, e6)
            assert rel_err(o3, want) < 3e-6
        # a loose bound: 2^10 above the data still fp32-class; every element keeps an ABSOLUTE error of a_range 2^-38, so 2^20
        # above, the error is 2^-18 of the data's scale over sqrt(K)-fold averaging — the bound has to be within ~2^14 of the RMS
        a, w = dev(torch.randn(512, 256)), dev(torch.randn(256, 256))
        want = a.double().cpu() @ w.double().cpu().t()
        out = torch.empty(512, 256, device="cuda")
        bp3 = ops.split_gemm_weights(w, h2=True)
        ops.gemm_nt_planes(a, None, bp3, out, a_range=4.0 * 2 ** 10)
        assert rel_err(out, want) < 3e-6
        ops.gemm_nt_planes(a, None, bp3, out, a_range=4.0 * 2 ** 20)
        assert 3e-6 < rel_err(out, want) < 3e-5
        # the epilogue's by-products and the live-row form
        cb = dev(torch.randn(256))
        _, cm, uf = ops.gemm_nt_planes(a, None, bp3, out, col_bias=cb, want_colmax=True, sk_eps=0.05, a_range=8.0)
        assert torch.equal(cm, out.amax(0))
        assert rel_err(uf, torch.exp((out.double() - cm.double()) / 0.05).sum(0)) < 2e-5
        live = torch.tensor([300], dtype=torch.int32, device="cuda")
        out2 = torch.full((512, 256), 7.0, device="cuda")
        ops.gemm_nt_planes_live(a, bp3, out2, live, 300 / 512, col_bias=cb, a_range=8.0)
        assert torch.equal(out2[:256], out[:256]) and bool((out2[384:] == 7.0).all())
        # large launches take 256-row tiles (a wave 64 rows: the 128-row form is LDS-bound); same products in the same order per
        # element -> BITWISE the 128-row result, ragged tail, per-slice by-products and the live-row form included
        from ganecdotes_amd import _lib
        L = _lib.load()
        for (m, n, k) in ((33000, 512, 64), (16500, 1028, 32)):
            a2, w2, cb2 = dev(torch.randn(m, k)), dev(torch.randn(n, k) * 0.05), dev(torch.randn(n))
            bp = ops.split_gemm_weights(w2, h2=True)
            res = []
            for tall in (1, 0):
                was_t = L.gx_set_gemm_tall(tall)
                o = torch.empty(m, n, device="cuda")
                _, cm_, uf_ = ops.gemm_nt_planes(a2, None, bp, o, col_bias=cb2, want_colmax=True, sk_eps=0.05, a_range=8.0)
                o2 = torch.full((m, n), 3.0, device="cuda")
                lv = torch.tensor([m - 5000], dtype=torch.int32, device="cuda")
                ops.gemm_nt_planes_live(a2, bp, o2, lv, 0.8, a_range=8.0)
                o3 = dev(torch.ones(m, n))
                ops.gemm_nt_planes(a2, None, bp, o3, alpha=0.5, beta=2.0, a_range=8.0)
                res.append((o, cm_, uf_, o2, o3))
                L.gx_set_gemm_tall(was_t)
            for i, (x, y) in enumerate(zip(res[0], res[1])):
                if i == 3:      # live rows: a tile is skipped whole, so rows past the count inside the last live tile differ by tile height
                    x, y = x[:m - 5000], y[:m - 5000]
                assert torch.equal(x, y), (m, n, k, i, float((x - y).abs().max()))
            assert torch.equal(res[0][1], res[0][0].amax(0))
            assert rel_err(res[0][0], a2.double().cpu() @ w2.double().cpu().t() + cb2.double().cpu()) < 3e-6
            assert bool((res[0][3][m - 5000 + 256:] == 3.0).all())
        # out of range: loud
        a[3, 5] = 100.0
        ops.gemm_nt_planes(a, None, bp3, out, a_range=1.0)
        assert bool(torch.isnan(out[3]).all()) and not bool(torch.isnan(out[4]).any())
        with pytest.raises(RuntimeError, match="a_range"):
            ops.gemm_nt_planes(a, None, bp3, out)
        # GX_GEMM_PLANES=0: every such call falls back to the bf16 planes
        ops.GEMM_H2 = False
        assert not ops.split_gemm_weights(w, h2=True).h2
    finally:
        ops.GEMM_H2 = was
        ops.set_gemm_mode(prev)


def test_gemm_ranged_f16x3_form_of_the_long_k_gemms(ops):
    """gx_gemm_f32_ranged: the wave-specialised kernel's f16x3 form (the TN prototype-gradient GEMM over the pixels at many
    prototypes: A = d(scores), bounded by P gscale / T, B = unit rows) against fp64 beside the bf16x6 form — gradient-sized
    operands, split-K, a device-counted k range, NT / NN / TN layouts; short-k launches ignore the bounds (bitwise bf16x6)."""
    torch.manual_seed(30)
    prev = ops.set_gemm_mode(1)
    was = ops.GEMM_H2
    ops.GEMM_H2 = True
    try:
        for (ta, tb, m, n, k, sa, sb, split) in ((True, False, 640, 512, 40000, 3e-7, 0.044, 8), (False, True, 1000, 260, 4096, 1.0, 0.02, 1),
                                                 (False, False, 512, 384, 2048, 50.0, 1.0, 2), (True, False, 5000, 512, 20000, 1e-5, 0.044, 4)):
            a = torch.randn((k, m) if ta else (m, k)) * sa
            b = torch.randn((n, k) if tb else (k, n)) * sb
            want = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
            ad, bd = dev(a), dev(b)
            o6 = ops.gemm(ad, bd, ta, tb, split_k=split)
            o3 = ops.gemm(ad, bd, ta, tb, split_k=split, a_range=float(a.abs().max()) * 2, b_range=float(b.abs().max()) * 3)
            e6 = float((o6.double().cpu() - want).norm() / want.norm())
            e3 = float((o3.double().cpu() - want).norm() / want.norm())
            assert e3 < 1.5 * e6 + 1e-7 and e3 < 3e-6, (ta, tb, m, n, k, e3, e6)
            assert not torch.equal(o3, o6)                                   # (it did take the other form)
        # the device-counted k range
        k, m, n = 30000, 640, 512
        a, b = dev(torch.randn(k, m) * 1e-4), dev(torch.randn(k, n) * 0.05)
        live = torch.tensor([17000], dtype=torch.int32, device="cuda")
        o3 = ops.gemm(a, b, True, False, split_k=4, live=live, live_frac=17 / 30, a_range=1e-3, b_range=0.5)
        want = a[:17000].double().cpu().t() @ b[:17000].double().cpu()
        assert rel_err(o3, want) < 3e-6
        # A's bound on the device (max |A| left by its producer, times a host factor), also for the two-source TN form
        k, m = 24000, 512
        a, b1, b2 = dev(torch.randn(k, m) * 3e-6), dev(torch.randn(k, 256)), dev(torch.randn(k, 128) * 2)
        amax = (a.abs().max() / 4).reshape(1).contiguous()
        want = a.double().cpu().t() @ torch.cat([b1, b2], 1).double().cpu()
        o3 = ops.gemm_tn_cat(a, b1, b2, torch.empty(m, 384, device="cuda"), split_k=4, a_amax=amax, a_mul=4.0, b_range=16.0)
        o6 = ops.gemm_tn_cat(a, b1, b2, torch.empty(m, 384, device="cuda"), split_k=4)
        assert rel_err(o3, want) < 3e-6 and not torch.equal(o3, o6)
        o3 = ops.gemm(a, b1, True, False, split_k=4, a_amax=amax, a_range=4.0, b_range=8.0)
        assert rel_err(o3, a.double().cpu().t() @ b1.double().cpu()) < 3e-6
        # short k: the 4-wave kernel, bounds ignored
        a, b = dev(torch.randn(300, 64)), dev(torch.randn(200, 64))
        assert torch.equal(ops.gemm(a, b, a_range=8.0, b_range=8.0), ops.gemm(a, b))
    finally:
        ops.GEMM_H2 = was
        ops.set_gemm_mode(prev)


def test_device_side_bounds_of_the_f16x3_forms(ops):
    """gx_absmax (max |x| of a strided matrix / a strided column, optionally of sqrt |x|) and the a_amax / b_amax arguments of the
    f16x3 GEMM forms: a bound read from the device gives the bits of the same bound passed as a host number."""
    torch.manual_seed(33)
    prev = ops.set_gemm_mode(1)
    was = ops.GEMM_H2
    ops.GEMM_H2 = True
    try:
        x = dev(torch.randn(1000, 300) * 7)
        assert float(ops.absmax(x)) == float(x.abs().max())
        assert float(ops.absmax(x[:, 5:77])) == float(x[:, 5:77].abs().max())
        assert float(ops.absmax(x[:, 13])) == float(x[:, 13].abs().max())
        assert float(ops.absmax(x[:, 13], sqrt=True)) == float(x[:, 13].abs().sqrt().max())
        assert float(ops.absmax(torch.zeros(64, 8, device="cuda"))) == 0.0
        big = dev(torch.randn(1 << 18, 280))
        sq = (big * big).sum(1)
        assert float(ops.absmax(sq, sqrt=True)) >= float(big.abs().max())           # a row norm bounds the row's elements
        # the weight-plane GEMM: A's bound on the device
        for (m, n, k, sa) in ((4096, 256, 512, 1.0), (1000, 132, 64, 2e-6)):
            a, w = dev(torch.randn(m, k) * sa), dev(torch.randn(n, k) * 0.05)
            ph = ops.split_gemm_weights(w, h2=True)
            amax = ops.absmax(a)
            o_dev = ops.gemm_nt_planes(a, None, ph, torch.empty(m, n, device="cuda"), alpha=2.0, a_amax=amax)
            o_host = ops.gemm_nt_planes(a, None, ph, torch.empty(m, n, device="cuda"), alpha=2.0, a_range=float(amax))
            assert torch.equal(o_dev, o_host)
            assert rel_err(o_dev, 2.0 * a.double().cpu() @ w.double().cpu().t()) < 3e-6
            o_mul = ops.gemm_nt_planes(a, None, ph, torch.empty(m, n, device="cuda"), alpha=2.0, a_amax=amax / 4, a_range=4.0)
            assert torch.equal(o_mul, o_host)
        with pytest.raises(RuntimeError):
            ops.gemm_nt_planes(a, None, ph, torch.empty(m, n, device="cuda"))           # f16x3 planes, no bound
        # the long-k TN GEMM: both bounds on the device
        k, m, n = 24000, 512, 256
        a, b = dev(torch.randn(k, m) * 0.3), dev(torch.randn(k, n) * 4e-6)
        am, bm = ops.absmax(a), ops.absmax(b)
        o_dev = ops.gemm(a, b, True, False, split_k=4, a_amax=am, b_amax=bm)
        o_host = ops.gemm(a, b, True, False, split_k=4, a_range=float(am), b_range=float(bm))
        assert torch.equal(o_dev, o_host)
        assert rel_err(o_dev, a.double().cpu().t() @ b.double().cpu()) < 3e-6
        assert not torch.equal(o_dev, ops.gemm(a, b, True, False, split_k=4))
    finally:
        ops.GEMM_H2 = was
        ops.set_gemm_mode(prev)


def test_gemm_tn_cat_one_pass_over_dz(ops, engine):
    """gx_gemm_f32_tn_cat: A^T [B1 | B2] in one launch == the two per-tap TN GEMMs (bitwise at equal split) == fp64"""
    torch.manual_seed(29)
    for k, m, n1, n2, split in ((700, 48, 128, 40, 3), (4096, 512, 256, 256, 8), (1030, 130, 128, 128, 1)):
        a, b1, b2 = torch.randn(k, m), torch.randn(k, n1 + 4), torch.randn(k, n2 + 8)
        ad, b1d, b2d = dev(a), dev(b1)[:, :n1], dev(b2)[:, :n2]
        prior = torch.randn(m, n1 + n2 + 6)
        out, ref = dev(prior.clone()), dev(prior.clone())
        ops.gemm_tn_cat(ad, b1d, b2d, out[:, 3:3 + n1 + n2], alpha=0.5, beta=1.0, split_k=split)
        ops.gemm(ad, b1d, True, False, out=ref[:, 3:3 + n1], alpha=0.5, beta=1.0, split_k=split)
        ops.gemm(ad, b2d, True, False, out=ref[:, 3 + n1:3 + n1 + n2], alpha=0.5, beta=1.0, split_k=split)
        want = prior.double()
        want[:, 3:3 + n1 + n2] += 0.5 * (a.double().t() @ torch.cat([b1[:, :n1], b2[:, :n2]], 1).double())
        assert rel_err(out, want) < 3e-6
        assert torch.equal(out, ref)
    with pytest.raises(RuntimeError):                      # N1 must be a whole number of 128-column tiles
        ops.gemm_tn_cat(dev(torch.randn(64, 32)), dev(torch.randn(64, 40)), dev(torch.randn(64, 8)),
                        torch.empty(32, 48, device="cuda"))


def test_pyramid_pieces(ops):
    torch.manual_seed(12)
    b, r, d = 2, 8, 24
    coarse, fine = torch.randn(b, r, r, d), torch.randn(b, 2 * r, 2 * r, d)
    fd = dev(fine.clone())
    ops.upsample2_add_(dev(coarse), fd)
    up = coarse.repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert torch.equal(fd.cpu(), fine + up)
    pooled = ops.sumpool2(dev(fine)).cpu()
    ref = fine.reshape(b, r, 2, r, 2, d).sum(dim=(2, 4))
    assert rel_err(pooled, ref) < 1e-6
    # adjointness <up(c), f> == <c, pool(f)>
    assert abs(float((up * fine).sum()) - float((coarse * ref).sum())) < 1e-3
    # gather + normalise and its scatter adjoint
    hw = 2 * r
    z = dev(fine)
    picks = torch.stack([torch.randperm(hw * hw)[:100] for _ in range(b)]).int().cuda()
    maps = torch.stack([ops.rotflip_map(9.0, True, hw, hw, "cuda"), ops.rotflip_map(-3.0, False, hw, hw, "cuda")])
    zh, rn = ops.gather_rows_normalize(z, picks, maps)
    src = torch.gather(maps.cpu().long(), 1, picks.cpu().long())                    # (b, 100)
    flat = fine.reshape(b, hw * hw, d)
    rows = torch.stack([flat[i][src[i].clamp(min=0)] * (src[i] >= 0).float()[:, None] for i in range(b)])
    ref = torch.nn.functional.normalize(rows.reshape(-1, d), dim=1)
    assert rel_err(zh, ref) < 1e-6
    gin = torch.randn(b * 100, d)
    x1 = rows.reshape(-1, d).clone().requires_grad_(True)
    torch.nn.functional.normalize(x1, dim=1).backward(gin)
    gfull = torch.zeros(b, hw, hw, d, device="cuda")
    amax = torch.zeros(1, device="cuda")
    ops.normalize_bwd_scatter_(dev(gin), zh, rn, picks, maps, gfull, amax=amax)
    want = torch.zeros(b, hw * hw, d)
    for i in range(b):
        ok = src[i] >= 0
        want[i].index_add_(0, src[i][ok], x1.grad.reshape(b, 100, d)[i][ok])
    assert rel_err(gfull.reshape(b, hw * hw, d), want) < 1e-5
    # amax: the largest single row added (zero-filled picks add nothing) — x the rows that can meet in a pixel it bounds |G|
    single = x1.grad.reshape(b, 100, d) * (src >= 0).float()[:, :, None]
    assert abs(float(amax) - float(single.abs().max())) <= 1e-6 * float(single.abs().max())
    from ganecdotes_amd.swav_clustering import SCATTER_MEET
    assert float(gfull.abs().max()) <= SCATTER_MEET * float(amax)
    # how many output pixels of a rotated view share a source pixel: never more than the factor the bound allows
    for ang in (0.0, 9.0, 33.0, 45.0, 61.0, 90.0):
        m = ops.rotflip_map(ang, ang > 30, 64, 64, "cuda")
        cnt = torch.bincount(m[m >= 0].long(), minlength=64 * 64)
        assert int(cnt.max()) <= int(SCATTER_MEET)


@pytest.mark.parametrize("d", [256, 512, 768, 1024, 260])
def test_gather_normalize_and_its_scatter_vector_rows(ops, d):
    """gx_gather_rows_normalize on rows of 256 NV floats (read once, 16 bytes per lane and load; 260: the dword form) and
    gx_normalize_bwd_scatter against torch: rotated / flipped views (zero-filled picks, source pixels drawn twice), autograd's gradient"""
    torch.manual_seed(d)
    b, hw, n = 2, 12, 120
    zc = torch.randn(b, hw, hw, d)
    z = dev(zc)
    picks = torch.stack([torch.randperm(hw * hw)[:n] for _ in range(b)]).int().cuda()
    maps = torch.stack([ops.rotflip_map(10.0, True, hw, hw, "cuda"), ops.rotflip_map(-7.0, False, hw, hw, "cuda")])
    zh, rn = ops.gather_rows_normalize(z, picks, maps)
    src = torch.gather(maps.cpu().long(), 1, picks.cpu().long())
    flat = zc.reshape(b, hw * hw, d)
    rows = torch.stack([flat[i][src[i].clamp(min=0)] * (src[i] >= 0).float()[:, None] for i in range(b)])
    assert rel_err(zh, torch.nn.functional.normalize(rows.reshape(-1, d), dim=1)) < 1e-6
    assert int((src < 0).sum()) > 0 and any(len(set(src[i][src[i] >= 0].tolist())) < int((src[i] >= 0).sum()) for i in range(b))
    gin = torch.randn(b * n, d)
    x1 = rows.reshape(-1, d).clone().requires_grad_(True)
    torch.nn.functional.normalize(x1, dim=1).backward(gin)
    gfull = torch.zeros(b, hw, hw, d, device="cuda")
    ops.normalize_bwd_scatter_(dev(gin), zh, rn, picks, maps, gfull)
    want = torch.zeros(b, hw * hw, d)
    for i in range(b):
        ok = src[i] >= 0
        want[i].index_add_(0, src[i][ok], x1.grad.reshape(b, n, d)[i][ok])
    assert rel_err(gfull.reshape(b, hw * hw, d), want) < 1e-5


@pytest.mark.parametrize("d,k", [(256, 8), (512, 12), (512, 16), (256, 3)])
def test_small_prototype_fused_kernels(ops, d, k):
    """fused K<=16 kernels == the unfused gather/normalise/GEMM/colsum/scatter chain"""
    torch.manual_seed(d + k)
    b, hw, n = 2, 16, 150
    z = dev(torch.randn(b, hw, hw, d))
    wc = dev(torch.nn.functional.normalize(torch.randn(k, d), dim=1))
    bias = dev(0.01 * torch.randn(k))
    picks = torch.stack([torch.randperm(hw * hw)[:n] for _ in range(b)]).int().cuda()
    maps = torch.stack([ops.rotflip_map(8.0, False, hw, hw, "cuda"), ops.rotflip_map(-5.0, True, hw, hw, "cuda")])
    assert ops.proto_small_ok(k, d)
    zh, rn, sc = ops.proto_fwd_small(z, picks, maps, wc, bias)
    zh0, rn0 = ops.gather_rows_normalize(z, picks, maps)
    sc0 = ops.gemm(zh0, wc, False, True, col_bias=bias)
    assert rel_err(zh, zh0) < 1e-6 and rel_err(rn, rn0) < 1e-6
    assert rel_err(sc, sc0) < 2e-6
    ds = dev(1e-3 * torch.randn(b * n, k))
    g1 = torch.zeros(b, hw, hw, d, device="cuda")
    gw1, gb1 = torch.full((k, d), 0.5, device="cuda"), torch.full((k,), 0.25, device="cuda")
    ops.proto_bwd_small_(ds, zh, rn, picks, maps, wc, g1, gw1, gb1, 1.0)
    g0 = torch.zeros(b, hw, hw, d, device="cuda")
    gw0, gb0 = torch.full((k, d), 0.5, device="cuda"), torch.full((k,), 0.25, device="cuda")
    ops.gemm(ds, zh0, True, False, out=gw0, beta=1.0, split_k=1)
    ops.colsum(ds, gb0, alpha=1.0, beta=1.0)
    dz = ops.gemm(ds, wc, False, False)
    ops.normalize_bwd_scatter_(dz, zh0, rn0, picks, maps, g0)
    assert rel_err(gw1, gw0) < 1e-5 and rel_err(gb1, gb0) < 1e-5
    assert rel_err(g1, g0) < 1e-5


# ------------------------------------------------------------------------------------- rows / optimizer
def test_row_ops(ops):
    torch.manual_seed(2)
    z = torch.randn(1001, 512)
    zd = dev(z.clone())
    rn = ops.row_l2_normalize_(zd)
    ref = torch.nn.functional.normalize(z, dim=1)
    assert rel_err(zd, ref) < 1e-6
    np.testing.assert_allclose(rn.cpu().numpy(), (1 / z.norm(dim=1)).numpy(), rtol=1e-5)
    # backward of normalize
    z1 = z.clone().requires_grad_(True)
    g = torch.randn(1001, 512)
    torch.nn.functional.normalize(z1, dim=1).backward(g)
    gd = dev(g.clone())
    ops.l2_normalize_bwd_(gd, zd, rn)
    assert rel_err(gd, z1.grad) < 1e-5
    assert torch.equal(ops.row_argmax(dev(z)).cpu(), z.max(1)[1])
    out = torch.zeros(512, device="cuda")
    ops.colsum(dev(z), out)
    assert rel_err(out, z.double().sum(0)) < 1e-5
    # row sums of squares over the first `cused` columns: every lane-team width of the 16-byte form, and the dword form (odd widths)
    for c, cused in ((24, 24), (32, 20), (64, 64), (128, 128), (256, 100), (512, 512), (512, 448), (77, 77), (64, 30), (1024, 1024)):
        x = torch.randn(1037, c)
        got = ops.row_sumsq(dev(x), cused)
        assert rel_err(got, (x[:, :cused].double() ** 2).sum(1)) < 1e-6


def test_larc_sgd(ops):
    from oracle import swav_ref as S
    torch.manual_seed(4)
    p, g = torch.randn(512, 777), torch.randn(512, 777) * 1e-3
    pd, gd, bd = dev(p.clone()), dev(g), torch.zeros(512, 777, device="cuda")
    bufs = [None]
    for step in range(3):
        bufs = S.larc_sgd_step([p], [g], bufs, lr=0.01, momentum=0.9, trust=0.01)
        ops.larc_sgd_(pd, gd, bd, 0.01, 0.9, 0.01, first_step=(step == 0))
        assert rel_err(pd, p) < 1e-6 and rel_err(bd, bufs[0]) < 1e-5


# ------------------------------------------------------------------------------------- generator layers
def _prep_conv(weight):
    """(1,Cout,Cin,k,k) -> wt (k*k,Cout,Cin), wsq (Cout,Cin)"""
    w = weight[0]
    cout, cin, k, _ = w.shape
    return w.permute(2, 3, 0, 1).reshape(k * k, cout, cin).contiguous(), w.pow(2).sum(dim=(2, 3)).contiguous()


def _style(ops, style, mod_w, mod_b):
    s = ops.gemm(dev(style), dev(mod_w), False, True, alpha=1.0 / math.sqrt(mod_w.shape[1]), col_bias=dev(mod_b))
    return s


def test_modconv_golden_gv2(ops, engine):
    d = load_golden("gv2_modconv")
    kf = dev(torch.flip(T(load_golden("gv1_upfirdn2d")["kernel"]) * 4, [0, 1]))
    for tag in ("plain", "up", "rgb"):
        w = T(d[f"{tag}_weight"])
        wt, wsq = _prep_conv(w)
        cin, k = w.shape[2], w.shape[3]
        scale = 1.0 / math.sqrt(cin * k * k)
        s = _style(ops, T(d[f"{tag}_style"]), T(d[f"{tag}_mod_weight"]), T(d[f"{tag}_mod_bias"]))
        x = dev(T(d[f"{tag}_x"]).permute(0, 2, 3, 1))
        ref = T(d[f"{tag}_y"])
        if tag == "plain":
            y = ops.modconv(x, dev(wt), 3, s, ops.demod(s, dev(wsq), scale), None, None, None, act=False)
        elif tag == "up":
            y = ops.modconv_up(x, dev(wt), s, ops.demod(s, dev(wsq), scale), kf, None, None, None, act=False)
        else:
            y = ops.modconv(x, dev(wt), 1, s, None, None, None, None, act=False) * scale
        assert rel_err(y.permute(0, 3, 1, 2), ref) < 5e-6, tag


def test_styled_conv_and_torgb_golden_gv3(ops, engine):
    d = load_golden("gv3_layers")
    kf = dev(torch.flip(T(d["up_sd_conv.blur.kernel"]), [0, 1]))
    for tag in ("plain", "up"):
        sd = {k[len(tag) + 4:]: T(v) for k, v in d.items() if k.startswith(f"{tag}_sd_")}
        wt, wsq = _prep_conv(sd["conv.weight"])
        cin = sd["conv.weight"].shape[2]
        scale = 1.0 / math.sqrt(cin * 9)
        s = _style(ops, T(d[f"{tag}_style"]), sd["conv.modulation.weight"], sd["conv.modulation.bias"])
        dm = ops.demod(s, dev(wsq), scale)
        x = dev(T(d[f"{tag}_x"]).permute(0, 2, 3, 1))
        args = (dev(T(d[f"{tag}_noise"])), dev(sd["noise.weight"]), dev(sd["activate.bias"]))
        y = ops.modconv(x, dev(wt), 3, s, dm, *args) if tag == "plain" else ops.modconv_up(x, dev(wt), s, dm, kf, *args)
        assert rel_err(y.permute(0, 3, 1, 2), T(d[f"{tag}_y"])) < 5e-6, tag
    sd = {k[7:]: T(v) for k, v in d.items() if k.startswith("rgb_sd_")}
    w3 = sd["conv.weight"][0, :, :, 0, 0].contiguous()
    s = _style(ops, T(d["rgb_style"]), sd["conv.modulation.weight"], sd["conv.modulation.bias"])
    x = dev(T(d["rgb_x"]).permute(0, 2, 3, 1))
    kfu = dev(torch.flip(sd["upsample.kernel"], [0, 1]))
    b3 = dev(sd["bias"].reshape(3))
    scale = 1.0 / math.sqrt(w3.shape[1])
    y = ops.torgb(x, dev(w3), s, scale, b3, dev(T(d["rgb_skip"])), kfu)
    assert rel_err(y, T(d["rgb_y_skip"])) < 5e-6
    y = ops.torgb(x, dev(w3), s, scale, b3, None, None)
    assert rel_err(y, T(d["rgb_y_noskip"])) < 5e-6


def test_mapping_pieces_golden_gv3(ops):
    d = load_golden("gv3_layers")
    x = T(d["el_x"])
    assert rel_err(ops.pixel_norm(dev(x)), T(d["pn_y"])) < 1e-6
    y = ops.gemm(dev(x), dev(d["el_weight"]), False, True)
    ops.bias_act_(y, alpha=0.01 / math.sqrt(16), bias=dev(d["el_bias"]), bscale=0.01, act=True)
    assert rel_err(y, T(d["el_y"])) < 1e-5


@pytest.mark.parametrize("k,rows,eps,mapped", [(200, 1000, 0.05, True), (5000, 9001, 0.005, True), (132, 37, 0.005, False),
                                               (1000, 40000, 0.05, True)])
def test_sinkhorn_first_rows_one_sweep(ops, k, rows, eps, mapped):
    """gx_sinkhorn_first_rows: column maxima + first Sinkhorn pass of a problem living on the rows rowmap[n] of a score table, in
    one sweep (online log-sum-exp per block, merged exactly) — against fp64; rows outside the problem carry huge scores that
    would underflow everything if they leaked into the reference; and a Sinkhorn problem started from it equals the
    gathered-copy problem"""
    torch.manual_seed(k + rows)
    ntab = rows * 3 if mapped else rows
    tab = (torch.randn(ntab, k) * 0.2)
    rowmap = None
    if mapped:
        idx = torch.randperm(ntab)[:rows]
        mask = torch.ones(ntab, dtype=torch.bool)
        mask[idx] = False
        tab[mask] += 50.0                               # not part of the problem
        rowmap = idx.to(torch.int32).cuda()
        sub = tab[idx]
    else:
        sub = tab
    tabd = tab.cuda()
    cm, u = ops.sinkhorn_first_rows(tabd, eps, rowmap)
    want_cm = sub.double().max(0).values
    want_u = torch.exp((sub.double() - want_cm) / eps).sum(0)
    assert torch.equal(cm.cpu().double(), want_cm)
    assert rel_err(u, want_u) < 2e-5
    a = ops.Sinkhorn(tabd, eps, 5, rowmap=rowmap)
    a.cmax_preset, a.u_first_preset = cm, u
    a.run()
    b = ops.Sinkhorn(sub.cuda().contiguous(), eps, 5).run()
    assert rel_err(a.la, b.la) < 1e-5 or float((a.la - b.la).abs().max()) < 1e-4
    assert rel_err(a.codes(), b.codes()) < 1e-4


@pytest.mark.parametrize("k,ngroups", [(200, 1), (5000, 1), (132, 3)])
def test_swapped_ce_pixels_equals_per_pair_ce_summed_per_pixel(ops, k, ngroups):
    """gx_swapped_ce_pixels (every patch of a step in one pixel-major pass, a pixel's d(scores) summed over the patches that drew
    it, tables written once) against gx_swapped_ce_fused per pair + an index_add per pixel: tables, loss and bias gradient"""
    torch.manual_seed(k)
    P, nimg, npp, n = 3, ngroups if ngroups > 1 else 2, 97, 60          # images, pixels per image, samples per image and patch
    rows, npix, kp = nimg * n, nimg * npp, (k + 15) // 16 * 16
    S = (torch.randn(P, 2, rows, k) * 0.1).cuda()
    skb = ops.SinkhornBatch(S.reshape(P * 2 * ngroups, rows // ngroups, k), 0.05, 5)
    skb.run()
    inv = torch.full((P, npix), -1, dtype=torch.int32)
    rows_to = []
    for p in range(P):
        idx = torch.cat([torch.randperm(npp)[:n] + b * npp for b in range(nimg)])
        inv[p, idx] = torch.arange(rows, dtype=torch.int32)
        rows_to.append(idx)
    inv = inv.cuda()
    ds, dt = torch.full((npix, kp), 7.0).cuda(), torch.full((npix, kp), 7.0).cuda()      # overwritten, padding zeroed
    loss, gb = torch.zeros(1).cuda(), torch.zeros(k).cuda()
    ops.swapped_ce_pixels(S, skb, ngroups, inv, 0.1, 1e-3, loss, 2.0, 0.0, ds, dt, kp, gb, 0.0)
    want_s, want_t = torch.zeros(npix, k, dtype=torch.float64), torch.zeros(npix, k, dtype=torch.float64)
    loss2, gb2 = torch.zeros(1).cuda(), torch.zeros(k).cuda()
    grows = rows // ngroups
    for p in range(P):
        d_s, d_t = torch.empty(rows, k).cuda(), torch.empty(rows, k).cuda()
        for g in range(ngroups):
            r0, r1 = g * grows, (g + 1) * grows
            ops.swapped_ce_fused(skb[(p * 2) * ngroups + g], skb[(p * 2 + 1) * ngroups + g], 0.1, 1e-3, loss2, 2.0, 1.0,
                                 d_s[r0:r1], d_t[r0:r1], gb2, 1.0)
        want_s.index_add_(0, rows_to[p], d_s.double().cpu())
        want_t.index_add_(0, rows_to[p], d_t.double().cpu())
    assert rel_err(ds[:, :k], want_s) < 2e-6 and rel_err(dt[:, :k], want_t) < 2e-6
    assert torch.all(ds[:, k:] == 0) and torch.all(dt[:, k:] == 0)
    assert abs(loss.item() - loss2.item()) < 1e-5 * abs(loss2.item())
    assert rel_err(gb, gb2) < 1e-5


@pytest.mark.parametrize("k,rows,nprob", [(200, 1000, 5), (5000, 333, 3), (132, 4097, 2), (132, 64, 100)])
def test_sinkhorn_problems_of_one_table_side_by_side(ops, k, rows, nprob):
    """gx_sinkhorn_rows_batch (the patches of a view as sorted row selections of one score table, one launch per pass, blocks
    of the problems interleaved) against the same problems solved one by one through gx_sinkhorn_pass_rows, and against the
    plain solver on gathered copies of their rows; 100 problems per table: two launches per pass"""
    torch.manual_seed(k + rows)
    npix = 2 * rows
    tabs = [(torch.randn(npix, k) * 0.2).cuda() for _ in range(2)]
    maps = [torch.sort(torch.randperm(npix)[:rows]).values.to(torch.int32).cuda() for _ in range(nprob)]
    tl = [tabs[v] for _ in range(nprob) for v in range(2)]
    ml = [maps[p] for p in range(nprob) for _ in range(2)]
    for eps in (0.05, 0.005):
        a = ops.SinkhornBatch.from_rows(tl, ml, eps, 6, 1.0 / rows, sorted_rows=True)
        b = ops.SinkhornBatch.from_rows(tl, ml, eps, 6, 1.0 / rows)
        for skb in (a, b):
            for sk_ in skb.probs:
                sk_.cmax_preset, sk_.u_first_preset = ops.sinkhorn_first_rows(sk_.S, eps, sk_.rowmap)
            skb.run()
        assert float((a.la - b.la).abs().max()) < 2e-4
        # without presets the batch makes its own column maxima and first pass (the per-image scope's many small problems);
        # more than 96 problems of a table go in several launches
        c = ops.SinkhornBatch.from_rows(tl, ml, eps, 6, 1.0 / rows, sorted_rows=True).run()
        assert float((c.la - b.la).abs().max()) < 2e-4
        for i in (0, 2 * nprob - 1):
            ref = ops.Sinkhorn(tl[i][ml[i].long()].contiguous(), eps, 6).run()
            assert float((a.la[i] - ref.la).abs().max()) < 2e-4
            assert rel_err(a[i].codes(), ref.codes()) < 1e-4


@pytest.mark.parametrize("k,npix,rows,P,live_n,pdf", [(200, 1500, 600, 5, 1300, False), (5000, 700, 333, 3, None, False),
                                                     (132, 9000, 4097, 2, 8000, True), (4000, 2600, 900, 5, 2401, True),
                                                     (6144, 300, 120, 5, 290, False), (1024, 500, 77, 1, None, False),
                                                     # the small forms (round 5): 64 / 128 / 256-thread workgroups up to 512 / 1024 / 2048 columns
                                                     (512, 900, 400, 5, 850, False), (700, 1100, 500, 4, None, True), (1600, 800, 300, 5, 700, True),
                                                     (2048, 600, 250, 3, 580, False),
                                                     # the wide form (16 columns per thread): the generic configuration's 8000 prototypes
                                                     (8000, 403, 150, 5, 390, False), (6400, 350, 100, 3, None, True),
                                                     (8192, 200, 64, 4, 190, False)])
def test_sinkhorn_table_all_patches_in_one_pass(ops, k, npix, rows, P, live_n, pdf):
    """gx_sinkhorn_table (every problem of a step — patches x views — solved on the two per-pixel score tables, a table row
    read once per pass for all the patches that drew it) against the same problems solved one by one on gathered copies of
    their rows, and its first pass against fp64; rows past `live` and rows no patch drew carry scores that would swamp
    everything if they leaked in; with the 'image' pdf's row / prototype marginals; and split over two faked ranks (rows of
    every problem in two halves, the exchanges as sums / maxima of the two) = the single-rank solve"""
    torch.manual_seed(k + npix)
    drawn = live_n if live_n is not None else npix
    eps, niters = 0.005 if k >= 1000 else 0.05, 6
    tabs = [(torch.randn(npix, k) * 0.2) for _ in range(2)]
    rowmaps = [torch.sort(torch.randperm(drawn)[:rows]).values for _ in range(P)]
    used = torch.zeros(npix, dtype=torch.bool)
    for m in rowmaps:
        used[m] = True
    for t in tabs:
        t[~used] += 40.0                                   # rows of no problem (drawn by no patch, or past the live bound)
    tabd = [t.cuda() for t in tabs]
    inv = torch.full((P, npix), -1, dtype=torch.int32)
    for p in range(P):
        inv[p, rowmaps[p]] = torch.arange(rows, dtype=torch.int32)
    inv = inv.cuda()
    live = None if live_n is None else torch.tensor([live_n], dtype=torch.int32, device="cuda")
    r = c = None
    if pdf:
        r = torch.rand(2 * P, k) + 0.1
        r = (r / r.sum(1, keepdim=True)).cuda()
        c = torch.rand(2 * P, rows) + 0.1
        c = (c / c.sum(1, keepdim=True)).cuda()
    assert ops.sinkhorn_table_ok(P, k)
    st = ops.SinkhornTable(tabd[0], tabd[1], inv, live, eps, niters, 1.0 / rows, rows, r=r, c=c,
                           rowmaps=[m.to(torch.int32).cuda() for m in rowmaps])
    st.first()
    for i in (0, 2 * P - 1):
        sub = tabs[i % 2][rowmaps[i // 2]].double()
        want_cm = sub.max(0).values
        assert torch.equal(st.cmax[i].cpu().double(), want_cm)
        assert rel_err(st.u[i], torch.exp((sub - want_cm) / eps).sum(0)) < 2e-5
    st.run()
    for i in range(2 * P):
        sub = tabd[i % 2][rowmaps[i // 2].cuda()].contiguous()
        ref = ops.Sinkhorn(sub, eps, niters, r=None if r is None else r[i].contiguous(), c=None if c is None else c[i].contiguous()).run()
        assert float((st.la[i] - ref.la).abs().max()) < 2e-4, i
        if i in (0, 2 * P - 1):
            assert rel_err(st[i].codes(), ref.codes()) < 1e-4
    # two faked ranks: each holds half of the table rows (its own tables, inv numbering its own half of every problem)
    half = [torch.arange(npix) % 2 == h for h in range(2)]
    parts = []
    for h in range(2):
        rows_h = torch.nonzero(half[h]).squeeze(1)
        invh = torch.full((P, rows_h.numel()), -1, dtype=torch.int32)
        ch = None if c is None else torch.zeros(2 * P, rows, device="cuda")
        for p in range(P):
            sel = inv[p].cpu()[rows_h]
            pos = torch.nonzero(sel >= 0).squeeze(1)
            invh[p, pos] = torch.arange(pos.numel(), dtype=torch.int32)
            if c is not None:
                for v in range(2):
                    ch[p * 2 + v, :pos.numel()] = c[p * 2 + v][sel[pos].long()]
        parts.append(ops.SinkhornTable(tabd[0][rows_h.cuda()].contiguous(), tabd[1][rows_h.cuda()].contiguous(), invh.cuda(), None, eps,
                                       niters, 1.0 / rows, rows, r=r, c=ch, tag=f"skt_half{h}"))
    for q in parts:
        q.first()
    cm = torch.maximum(parts[0].cmax, parts[1].cmax)
    for q in parts:
        q.rebase(cm)
    usum = parts[0].u + parts[1].u
    for q in parts:
        q.u.copy_(usum)
        q.update(True, cmax=cm)
    for it in range(1, niters):
        for q in parts:
            q.sweep(shifted=it > 1)
        usum = parts[0].u + parts[1].u
        for q in parts:
            q.u.copy_(usum)
            q.update(False)
    assert float((parts[0].la - st.la).abs().max()) < 2e-4 and torch.equal(parts[0].la, parts[1].la)
    # every sweep in its exact form (row maxima first) = the solve above, whose later sweeps ran in the shifted form wherever
    # the update vouched for it; and a problem the update does NOT vouch for (its steps spread over more than 64 log2 units)
    # falls back to the exact form inside the same call
    ex = ops.SinkhornTable(tabd[0], tabd[1], inv, live, eps, niters, 1.0 / rows, rows, r=r, c=c, tag="skt_exact")
    ex.first()
    ex.update(True)
    for _ in range(1, niters):
        ex.sweep(shifted=False)
        ex.update(False)
    assert float((ex.la - st.la).abs().max()) < 2e-4
    assert st.stats[:, 1].view(torch.int32).min().item() == 1          # (this data: every problem ran shifted)
    ex.stats.copy_(st.stats)
    ex.stats[0, 1] = 0.0                                               # problem 0 not vouched for
    la_before = ex.la.clone()
    ex.sweep(shifted=True)
    u_mixed = ex.u.clone()
    ex.la.copy_(la_before)
    ex.sweep(shifted=False)
    assert rel_err(u_mixed, ex.u) < 1e-5


@pytest.mark.parametrize("k,B,nl,N,P,pdf", [(200, 3, 500, 200, 5, False), (5000, 2, 150, 64, 3, True), (132, 5, 1000, 999, 2, False),
                                            (8000, 2, 120, 50, 5, False), (700, 3, 300, 120, 4, True), (1600, 2, 200, 90, 5, False)])
def test_sinkhorn_table_per_image_problems(ops, k, B, nl, N, P, pdf):
    """gx_sinkhorn_table with one problem per image too (the per-image Sinkhorn scope: the reference's batch-1 semantics): image
    b's live rows are the run [seg[b], seg[b + 1]) of the table, a workgroup stays inside one image; against every problem
    solved on a gathered copy of its rows"""
    torch.manual_seed(k + B)
    npix = B * nl + 37                                      # the tail: pixels no patch drew
    eps, niters = 0.005 if k >= 1000 else 0.05, 6
    tabs = [(torch.randn(npix, k) * 0.2) for _ in range(2)]
    for t in tabs:
        t[B * nl:] += 40.0
    rowmaps = [torch.cat([torch.sort(torch.randperm(nl)[:N]).values + b * nl for b in range(B)]) for _ in range(P)]
    used = torch.zeros(npix, dtype=torch.bool)
    for m in rowmaps:
        used[m] = True
    for t in tabs:
        t[~used] += 40.0
    tabd = [t.cuda() for t in tabs]
    rows = B * N
    inv = torch.full((P, npix), -1, dtype=torch.int32)
    for p in range(P):
        inv[p, rowmaps[p]] = torch.arange(rows, dtype=torch.int32)
    seg = (torch.arange(B + 1, dtype=torch.int32) * nl).cuda()
    live = torch.tensor([B * nl], dtype=torch.int32, device="cuda")
    r = c = None
    if pdf:
        r = torch.rand(2 * P * B, k) + 0.1
        r = (r / r.sum(1, keepdim=True)).cuda()
        c = torch.rand(2 * P, B, N) + 0.1
        c = (c / c.sum(2, keepdim=True)).reshape(2 * P, rows).cuda()
    st = ops.SinkhornTable(tabd[0], tabd[1], inv.cuda(), live, eps, niters, 1.0 / N, rows, r=r, c=c,
                           rowmaps=[m.to(torch.int32).cuda() for m in rowmaps], ngroups=B, seg=seg)
    st.run()
    assert st.la.shape == (2 * P * B, k)
    for i in range(2 * P * B):
        pv, b = divmod(i, B)
        sub = tabd[pv % 2][rowmaps[pv // 2][b * N:(b + 1) * N].cuda()].contiguous()
        ref = ops.Sinkhorn(sub, eps, niters, r=None if r is None else r[i].contiguous(),
                           c=None if c is None else c[pv][b * N:(b + 1) * N].contiguous()).run()
        assert float((st.la[i] - ref.la).abs().max()) < 2e-4, i
    assert rel_err(st[2 * P * B - 1].codes(), ref.codes()) < 1e-4
    with pytest.raises(RuntimeError):
        ops.SinkhornTable(tabd[0], tabd[1], inv.cuda(), live, eps, niters, 1.0 / N, rows, ngroups=B)          # no seg


@pytest.mark.parametrize("k,live_n", [(200, 150), (5000, 233), (132, None), (8000, 190), (6000, None)])
def test_swapped_ce_pixels_on_score_tables_equals_the_gathered_form(ops, k, live_n):
    """gx_swapped_ce_pixels_tab (scores as per-pixel tables, Sinkhorn problems row-mapped onto them) against
    gx_swapped_ce_pixels on the gathered score rows with the same row potentials; with a device-side live bound the rows of
    the d(scores) tables past it keep what they held and nothing of them reaches the loss or the bias gradient"""
    torch.manual_seed(k + 1)
    P, npix, rows, kp = 3, 233, 60, (k + 15) // 16 * 16
    drawn = live_n if live_n is not None else npix
    tabs = [(torch.randn(npix, k) * 0.1).cuda() for _ in range(2)]
    rowmaps = [torch.randperm(drawn)[:rows].to(torch.int32).cuda() for _ in range(P)]
    skb = ops.SinkhornBatch.from_rows([tabs[v] for _ in range(P) for v in range(2)], [rowmaps[p] for p in range(P) for _ in range(2)],
                                      0.05, 5, 1.0 / k)
    skb.run()
    inv = torch.full((P, npix), -1, dtype=torch.int32, device="cuda")
    for p in range(P):
        inv[p].scatter_(0, rowmaps[p].long(), torch.arange(rows, dtype=torch.int32, device="cuda"))
    live = None if live_n is None else torch.tensor([live_n], dtype=torch.int32, device="cuda")
    ds, dt = torch.full((npix, kp), 7.0).cuda(), torch.full((npix, kp), 7.0).cuda()
    loss, gb = torch.zeros(1).cuda(), torch.zeros(k).cuda()
    ops.swapped_ce_pixels_tab(tabs[0], tabs[1], skb, 1, inv, 0.1, 1e-3, loss, 2.0, 0.0, ds, dt, kp, gb, 0.0, live=live)
    S = torch.stack([torch.stack([tabs[v][rowmaps[p].long()] for v in range(2)]) for p in range(P)]).contiguous()
    ds2, dt2 = torch.full((npix, kp), 7.0).cuda(), torch.full((npix, kp), 7.0).cuda()
    loss2, gb2 = torch.zeros(1).cuda(), torch.zeros(k).cuda()
    ops.swapped_ce_pixels(S, skb, 1, inv, 0.1, 1e-3, loss2, 2.0, 0.0, ds2, dt2, kp, gb2, 0.0)
    assert rel_err(ds[:drawn], ds2[:drawn]) < 1e-6 and rel_err(dt[:drawn], dt2[:drawn]) < 1e-6
    assert abs(loss.item() - loss2.item()) < 1e-6 * abs(loss2.item()) and rel_err(gb, gb2) < 1e-6
    if live_n is not None:
        assert torch.all(ds[live_n:] == 7.0) and torch.all(dt[live_n:] == 7.0)     # untouched: no GEMM reads them
        assert torch.all(ds2[live_n:] == 0.0)                                      # (the unbounded form writes zeros there)
    with pytest.raises(RuntimeError):
        ops.swapped_ce_pixels_tab(tabs[0], tabs[1], skb, 1, inv, 0.1, 1e-3, loss, 2.0, 0.0, ds, dt, kp, gb, 0.0,
                                  live=torch.tensor([3], dtype=torch.int32))


@pytest.mark.parametrize("rows,k,n,act", [(1, 512, 512, True), (5, 16, 16, True), (48, 512, 512, True), (200, 512, 100, False),
                                          (13, 64, 70, True)])
def test_equal_linear_skinny_vs_fp64_and_gv3(ops, rows, k, n, act):
    """the one-launch EqualLinear (+ fused_leaky_relu) of the mapping network on a batch of latents: against fp64, against
    the GEMM + bias_act route it replaces, and against the reference's own EqualLinear output (GV3)"""
    torch.manual_seed(rows + n)
    x = torch.randn(rows, k)
    w = torch.randn(n, k)
    b = torch.randn(n)
    alpha, bscale = 0.01 / math.sqrt(k), 0.01
    pre = (x.double() @ w.double().t()) * alpha + b.double() * bscale
    want = torch.nn.functional.leaky_relu(pre, 0.2) * math.sqrt(2.0) if act else pre
    got = ops.equal_linear_skinny(dev(x), dev(w.t().contiguous()), alpha, dev(b), bscale, act=act)
    assert got.shape == (rows, n) and rel_err(got, want) < 2e-6
    y = ops.gemm(dev(x), dev(w), False, True)
    ops.bias_act_(y, alpha=alpha, bias=dev(b), bscale=bscale, act=act)
    assert rel_err(got, y) < 1e-5
    if (rows, k) == (5, 16):
        d = load_golden("gv3_layers")
        got = ops.equal_linear_skinny(dev(T(d["el_x"])), dev(T(d["el_weight"]).t().contiguous()), 0.01 / math.sqrt(16),
                                      dev(d["el_bias"]), 0.01, act=True)
        assert rel_err(got, T(d["el_y"])) < 1e-5


def test_mapping_network_skinny_equals_gemm_route(ops):
    """MappingNetwork.forward: 48 latents through the one-launch layers == through the split-K GEMM layers; the weight's
    transposed copy follows the parameter (in-place update -> remade)"""
    from ganecdotes_amd import generator as Gm
    torch.manual_seed(3)
    net = Gm.MappingNetwork(512, 8, 0.01).cuda()
    z = torch.randn(48, 512, device="cuda")
    a = net(z)
    prev, Gm.SKINNY_ROWS = Gm.SKINNY_ROWS, 0
    try:
        b = net(z)
        assert rel_err(a, b) < 2e-5
        with torch.no_grad():
            getattr(net, "1").weight.mul_(1.5)
        b2 = net(z)
    finally:
        Gm.SKINNY_ROWS = prev
    assert rel_err(net(z), b2) < 2e-5 and rel_err(b2, b) > 1e-3


def test_modconv_ffhq_shapes(ops, engine):
    """A real-width layer (512->512 @ 16x16, batch 2) against the oracle."""
    from oracle import stylegan2_ref as G
    torch.manual_seed(21)
    b, cin, cout, h = 2, 512, 512, 16
    w = torch.randn(1, cout, cin, 3, 3)
    mod_w, mod_b = torch.randn(cin, 512), torch.ones(cin)
    x, style = torch.randn(b, cin, h, h), torch.randn(b, 512)
    blur = G.fir_kernel() * 4
    wt, wsq = _prep_conv(w)
    scale = 1.0 / math.sqrt(cin * 9)
    s = _style(ops, style, mod_w, mod_b)
    dm = ops.demod(s, dev(wsq), scale)
    xd = dev(x.permute(0, 2, 3, 1))
    y = ops.modconv(xd, dev(wt), 3, s, dm, None, None, None, act=False)
    ref = G.modulated_conv(x, style, w, mod_w, mod_b)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-5
    y = ops.modconv_up(xd, dev(wt), s, dm, dev(torch.flip(blur, [0, 1])), None, None, None, act=False)
    ref = G.modulated_conv(x, style, w, mod_w, mod_b, upsample=True, blur_kernel=blur)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 1e-5


def test_modconv_random_shapes_both_engines(ops, engine):
    """12 random layers per engine (batch, channels incl. non-multiples of 16 / 128, resolution, plain or
    upsampling, with and without the pre-split weight planes, noise + bias + activation) against the oracle"""
    from oracle import stylegan2_ref as G
    rng = np.random.RandomState(77 + engine)
    blur = G.fir_kernel() * 4
    for case in range(12):
        b = int(rng.choice([1, 2, 3]))
        cin = int(rng.choice([8, 16, 24, 40, 128, 136]))
        cout = int(rng.choice([8, 12, 64, 128, 132, 200]))
        h = int(rng.choice([4, 5, 8, 16, 33]))
        up = bool(rng.randint(2)) and h <= 16
        planes = bool(rng.randint(2))
        torch.manual_seed(1000 + case)
        w = torch.randn(1, cout, cin, 3, 3)
        mod_w, mod_b = torch.randn(cin, 32), torch.ones(cin)
        x, style = torch.randn(b, cin, h, h), torch.randn(b, 32)
        oh = 2 * h if up else h
        noise, nw, bias = torch.randn(b, 1, oh, oh), torch.tensor([0.3]), 0.1 * torch.randn(cout)
        wt, wsq = _prep_conv(w)
        s = _style(ops, style, mod_w, mod_b)
        dm = ops.demod(s, dev(wsq), 1.0 / math.sqrt(cin * 9))
        xd, wd = dev(x.permute(0, 2, 3, 1)), dev(wt)
        wp = ops.split_conv_weights(wd) if planes else None
        if up:
            y = ops.modconv_up(xd, wd, s, dm, dev(torch.flip(blur, [0, 1])), dev(noise), dev(nw), dev(bias), True, wt_planes=wp)
            ref = G.modulated_conv(x, style, w, mod_w, mod_b, upsample=True, blur_kernel=blur)
        else:
            y = ops.modconv(xd, wd, 3, s, dm, dev(noise), dev(nw), dev(bias), True, wt_planes=wp)
            ref = G.modulated_conv(x, style, w, mod_w, mod_b)
        ref = G.lrelu_bias_scale(ref + nw * noise, bias)
        assert rel_err(y.permute(0, 3, 1, 2), ref) < 2e-5, (case, b, cin, cout, h, up, planes)


@pytest.mark.parametrize("rows,hw,n,quant", [(80, 65536, 20000, 0), (3, 1024, 300, 0), (4, 65536, 20000, 64),
                                             (2, 262144, 20000, 0), (2, 1000, 1000, 8), (5, 4097, 1, 0)])
def test_random_subset_picks(ops, rows, hw, n, quant):
    """radix-select picks == the n smallest keys of every row (the set argsort(keys)[:n] selects), distinct,
    exactly n, in ascending pixel order (keys below the n-th smallest first, then the tied ones) — also with
    heavily tied keys (quant levels), n == HW and n == 1"""
    torch.manual_seed(rows * hw + n)
    keys = torch.rand(rows, hw, device="cuda")
    if quant:
        keys = torch.floor(keys * quant) / quant
    picks = ops.random_subset_picks(keys, n).cpu().long()
    kc = keys.cpu()
    for r in range(rows):
        p = picks[r]
        assert p.shape == (n,) and int(p.min()) >= 0 and int(p.max()) < hw
        assert len(torch.unique(p)) == n
        picked = torch.zeros(hw, dtype=torch.bool)
        picked[p] = True
        if n < hw:
            assert float(kc[r][picked].max()) <= float(kc[r][~picked].min())
        if not quant:
            assert torch.equal(torch.sort(p)[0], torch.sort(torch.argsort(kc[r])[:n])[0])
            assert torch.equal(p[:-1], torch.sort(p[:-1])[0])    # ascending; the n-th smallest key's pixel comes last
    # uniformity smoke check: every pixel is picked with probability n / hw
    if rows >= 80:
        freq = torch.zeros(hw)
        for r in range(rows):
            freq[picks[r]] += 1
        assert abs(float(freq.mean()) / rows - n / hw) < 1e-6


# ------------------------------------------------------------------------------------- the reference's native-op signatures
def test_upfirdn2d_reference_signature_gv1_gv16(ops):
    """gx_upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    (lib/gan/optim/upfirdn2d.cpp:18-35): GV1's own outputs for the generator's modes 1 and 3, and GV16's
    general parameter sets (down 2, mixed up/down, non-square kernel, negative pads, minor > 1)."""
    d = load_golden("gv1_upfirdn2d")
    k = dev(T(d["kernel"]) * 4)
    y1 = ops.upfirdn2d(dev(T(d["x_mode1"])), k, up=1, down=1, pad=(1, 1))
    y3 = ops.upfirdn2d(dev(T(d["x_mode3"])), k, up=2, down=1, pad=(2, 1))
    assert tuple(y1.shape) == d["y_mode1"].shape and tuple(y3.shape) == d["y_mode3"].shape
    assert rel_err(y1, T(d["y_mode1"])) < 1e-6 and rel_err(y3, T(d["y_mode3"])) < 1e-6
    g = load_golden("gv16_native_ops")
    for i, case in enumerate(g["upfirdn_cases"]):
        rest = [int(v) for v in case[6:]]
        y = ops.upfirdn2d_native(dev(T(g[f"up{i}_x"])), dev(T(g[f"up{i}_k"])), *rest)
        assert tuple(y.shape) == g[f"up{i}_y"].shape, i
        assert rel_err(y, T(g[f"up{i}_y"])) < 2e-6, i
    with pytest.raises(RuntimeError):                      # CHECK_CUDA of the reference boundary (upfirdn2d.cpp:9-15)
        ops.upfirdn2d_native(T(g["up0_x"]), T(g["up0_k"]), 1, 1, 1, 1, 0, 0, 0, 0)
    with pytest.raises(RuntimeError):                      # CHECK_CONTIGUOUS
        ops.upfirdn2d_native(dev(T(g["up3_x"])).transpose(1, 2), dev(T(g["up3_k"])), 1, 1, 1, 1, 0, 0, 0, 0)


def test_fused_bias_act_reference_signature_gv3_gv16(ops):
    """gx_fused_bias_act(input, bias, refer, act, grad, alpha, scale) (fused_bias_act.cpp:18-31): forward and the
    grad = 1 / 2 modes against GV16 (fused_leaky_relu + its autograd), the EqualLinear(activation) output
    of GV3 through it, and the linear activation."""
    g = load_golden("gv16_native_ops")
    x, b = dev(T(g["fl_x"])), dev(T(g["fl_b"]))
    y = ops.fused_leaky_relu(x, b)
    assert rel_err(y, T(g["fl_y"])) < 1e-6
    gx = ops.fused_bias_act(dev(T(g["fl_gy"])), None, y, 3, 1, 0.2, 2 ** 0.5)
    assert rel_err(gx, T(g["fl_gx"])) < 1e-6
    assert float(ops.fused_bias_act(x, None, y, 3, 2, 0.2, 1.0).abs().max()) == 0.0
    assert rel_err(ops.fused_bias_act(dev(T(g["fl2_x"])), dev(T(g["fl2_b"])), None, 3, 0, 0.2, 1.5), T(g["fl2_y"])) < 1e-6
    lin = ops.fused_bias_act(x, b, None, 1, 0, 0.2, 3.0)
    assert rel_err(lin, (T(g["fl_x"]) + T(g["fl_b"]).reshape(1, -1, 1, 1)) * 3.0) < 1e-6
    # GV3: EqualLinear(lr_mul = 0.01, activation='fused_lrelu') = fused_leaky_relu(F.linear(x, W * scale), bias * lr_mul)
    # with scale = lr_mul / sqrt(in)   (model.py:242-252)
    d = load_golden("gv3_layers")
    w, xin = T(d["el_weight"]), T(d["el_x"])
    lin = ops.gemm(dev(xin), dev(w), False, True, alpha=0.01 / math.sqrt(w.shape[1]))
    assert rel_err(ops.fused_leaky_relu(lin, dev(T(d["el_bias"]) * 0.01)), T(d["el_y"])) < 2e-6


# ------------------------------------------------------------------------------------- x-planes conv path
def test_modconv_xp_golden_gv2_gv3(ops):
    """The pre-split-activation conv (gx_modsplit + gx_modconv_fwd_xp / gx_modconv_up_convt_xp: both operands by
    LDS-DMA) on the reference's own ModulatedConv2d / StyledConv outputs (GV2, GV3): Cin = 8 (zero channel
    padding to 16), Cout = 12 / 6 (tile tails), one 128-row tile spanning both images."""
    prev = ops.set_gemm_mode(1)
    try:
        d = load_golden("gv2_modconv")
        kf = dev(torch.flip(T(load_golden("gv1_upfirdn2d")["kernel"]) * 4, [0, 1]))
        for tag in ("plain", "up", "rgb"):
            w = T(d[f"{tag}_weight"])
            wt, wsq = _prep_conv(w)
            cin, k = w.shape[2], w.shape[3]
            scale = 1.0 / math.sqrt(cin * k * k)
            s = _style(ops, T(d[f"{tag}_style"]), T(d[f"{tag}_mod_weight"]), T(d[f"{tag}_mod_bias"]))
            x = dev(T(d[f"{tag}_x"]).permute(0, 2, 3, 1))
            wtd = dev(wt)
            wtp, xp = ops.split_conv_weights(wtd), ops.modsplit(x, s)
            if tag == "plain":
                y = ops.modconv(x, wtd, 3, s, ops.demod(s, dev(wsq), scale), None, None, None, act=False, wt_planes=wtp,
                                x_planes=xp)
            elif tag == "up":
                y = ops.modconv_up(x, wtd, s, ops.demod(s, dev(wsq), scale), kf, None, None, None, act=False,
                                   wt_planes=wtp, x_planes=xp)
            else:
                y = ops.modconv(x, wtd, 1, s, None, None, None, None, act=False, wt_planes=wtp, x_planes=xp) * scale
            assert rel_err(y.permute(0, 3, 1, 2), T(d[f"{tag}_y"])) < 5e-6, tag
        d = load_golden("gv3_layers")
        kf = dev(torch.flip(T(d["up_sd_conv.blur.kernel"]), [0, 1]))
        for tag in ("plain", "up"):
            sd = {k[len(tag) + 4:]: T(v) for k, v in d.items() if k.startswith(f"{tag}_sd_")}
            wt, wsq = _prep_conv(sd["conv.weight"])
            scale = 1.0 / math.sqrt(sd["conv.weight"].shape[2] * 9)
            s = _style(ops, T(d[f"{tag}_style"]), sd["conv.modulation.weight"], sd["conv.modulation.bias"])
            dm = ops.demod(s, dev(wsq), scale)
            x = dev(T(d[f"{tag}_x"]).permute(0, 2, 3, 1))
            wtd = dev(wt)
            kw = dict(wt_planes=ops.split_conv_weights(wtd), x_planes=ops.modsplit(x, s))
            args = (dev(T(d[f"{tag}_noise"])), dev(sd["noise.weight"]), dev(sd["activate.bias"]))
            y = (ops.modconv(x, wtd, 3, s, dm, *args, **kw) if tag == "plain"
                 else ops.modconv_up(x, wtd, s, dm, kf, *args, **kw))
            assert rel_err(y.permute(0, 3, 1, 2), T(d[f"{tag}_y"])) < 5e-6, tag
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("b,h,cout,edge", [(16, 64, 256, False), (8, 128, 128, True), (32, 32, 512, False), (8, 64, 128, False)])   # the last: too few strips for the large form (both runs take the same kernel)
def test_blur_pass_with_lds_dma_rows_equals_the_register_form(ops, b, h, cout, edge):
    """The blur + activation pass of an up-sampling layer with its T rows arriving by LDS-DMA five rows ahead (gx_conv.hip
    blur_act_dma_kernel: counted vmcnt waits across the interleaved stores, raw barriers, the ring read by inline asm) against
    the register form it replaces on the large launches (gx_set_blur_dma(0)): the same FIR sums in the same order — fp32 taps
    and the three bf16 planes of out * s_next bitwise equal, image borders (zero rows / columns outside T) included."""
    from ganecdotes_amd import _lib
    torch.manual_seed(b + h)
    cin = 64
    x = dev(torch.randn(b, h, h, cin))
    wt = dev(torch.randn(9, cout, cin) / math.sqrt(9 * cin))
    s = dev(torch.rand(b, cin) + 0.5)
    dm = dev(torch.rand(b, cout) + 0.5)
    sn = dev(torch.rand(b, cout) + 0.5)
    kf = dev(torch.flip(T(load_golden("gv1_upfirdn2d")["kernel"]) * 4, [0, 1]))
    noise, nw, bias = dev(torch.randn(1, 1, 2 * h, 2 * h)), dev(torch.tensor([0.3])), dev(torch.randn(cout))
    if edge:                                   # values that make a wrong halo visible: large at the image border
        x[:, 0] *= 50.0; x[:, -1] *= 50.0; x[:, :, 0] *= 50.0; x[:, :, -1] *= 50.0
    wtp = ops.split_conv_weights(wt)
    prev = ops.set_gemm_mode(1)
    L = _lib.load()
    try:
        outs = []
        for dma in (1, 0):
            was = L.gx_set_blur_dma(dma)
            planes = ops.xplanes_buffer(b, 2 * h, 2 * h, cout, x.device, slot=f"blur_dma_test{dma}")
            planes.zero_()
            xp = ops.modsplit(x, s)
            y = ops.modconv_up(x, wt, s, dm, kf, noise, nw, bias, True, wt_planes=wtp, x_planes=xp, next_style=sn, next_planes=planes)
            outs.append((y.clone(), planes.clone()))
            L.gx_set_blur_dma(was)
        assert torch.equal(outs[0][0], outs[1][0]), float((outs[0][0] - outs[1][0]).abs().max())
        assert torch.equal(outs[0][1], outs[1][1])
        assert float(outs[0][0].abs().max()) > 0
    finally:
        ops.set_gemm_mode(prev)
        L.gx_set_blur_dma(1)


@pytest.mark.parametrize("b,h,cin,cout,up,xs", [(4, 64, 128, 128, False, 1.0), (2, 64, 256, 128, True, 1.0), (16, 128, 64, 128, False, 1.0),
                                                (4, 64, 128, 128, False, 300.0), (4, 64, 128, 128, False, 0.02)])
def test_modconv_plane_formats_against_the_exact_engine(ops, b, h, cin, cout, up, xs):
    """The x-planes conv under its two plane formats — three bf16 planes / six products, two fp16 planes of the scaled operands /
    three products (gx_set_conv_planes) — against the exact-fp32 MFMA engine (bitwise an fmaf chain) on the same inputs: plain
    and up-sampling layers, 128- and 256-row tiles, activations of scale 1, 300 and 0.02 (inside the f16x3 window): the f16x3
    error is within 1.5 x the bf16x6 one and both stay at fp32 summation-order level (K = 9 Cin up to 2304)."""
    torch.manual_seed(b + h + cin)
    x = dev(torch.randn(b, h, h, cin) * xs)
    wt = dev(torch.randn(9, cout, cin) / math.sqrt(9 * cin))
    s = dev(torch.rand(b, cin) + 0.5)
    dm = dev((torch.rand(b, cout) + 0.5) / xs)
    kf = dev(torch.flip(T(load_golden("gv1_upfirdn2d")["kernel"]) * 4, [0, 1]))
    oh = 2 * h if up else h
    noise, nw, bias = dev(torch.randn(1, 1, oh, oh)), dev(torch.tensor([0.3])), dev(torch.randn(cout))
    prev, prev_fmt = ops.set_gemm_mode(0), ops.get_conv_planes()
    try:
        if up:
            ref = ops.modconv_up(x, wt, s, dm, kf, noise, nw, bias, True)
        else:
            ref = ops.modconv(x, wt, 3, s, dm, noise, nw, bias, True)
        ops.set_gemm_mode(1)
        err = {}
        for fmt in (0, 1):
            ops.set_conv_planes(fmt)
            wtp = ops.split_conv_weights(wt)
            xp = ops.modsplit(x, s)
            if up:
                got = ops.modconv_up(x, wt, s, dm, kf, noise, nw, bias, True, wt_planes=wtp, x_planes=xp)
            else:
                got = ops.modconv(x, wt, 3, s, dm, noise, nw, bias, True, wt_planes=wtp, x_planes=xp)
            err[fmt] = float((got.double() - ref.double()).norm() / ref.double().norm())
        assert err[1] < 1.5 * err[0] + 5e-8 and err[1] < 1e-6, err
    finally:
        ops.set_conv_planes(prev_fmt)
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("b,h,cin,cout,up", [(3, 16, 64, 160, False), (2, 24, 40, 128, True), (5, 8, 512, 512, False),
                                             (1, 64, 128, 96, True), (2, 33, 20, 72, False),
                                             # Cout <= 64: the tall 256-pixel x 64-channel tile (the 512^2 layers of a 512^2 generator)
                                             (2, 48, 128, 64, False), (2, 40, 128, 64, True), (3, 21, 64, 32, False),
                                             (1, 131, 64, 64, False), (1, 9, 24, 16, True)])
def test_modconv_xp_equals_in_kernel_split(ops, b, h, cin, cout, up):
    """random shapes (odd sizes, several images per tile, Cin not a multiple of 16, Cout tails, all four parity
    classes of the transposed conv): the x-planes path computes the bf16x6 sums of the in-kernel-split path — the same
    six products per fp32 product in the same k order (pairs of them inside one 16x16x32 MFMA) — so the two agree to
    fp32 summation-order rounding."""
    prev = ops.set_gemm_mode(1)
    try:
        torch.manual_seed(b * 100 + h)
        x = dev(torch.randn(b, h, h + 1, cin))
        wt = dev(torch.randn(9, cout, cin) / math.sqrt(9 * cin))
        s = dev(torch.rand(b, cin) + 0.5)
        dm = dev(torch.rand(b, cout) + 0.5)
        wtp = ops.split_conv_weights(wt)
        kf = dev(torch.flip(T(load_golden("gv1_upfirdn2d")["kernel"]) * 4, [0, 1]))
        oh, ow = (2 * h, 2 * (h + 1)) if up else (h, h + 1)
        noise, nw, bias = dev(torch.randn(1, 1, oh, ow)), dev(torch.tensor([0.3])), dev(torch.randn(cout))
        xp = ops.modsplit(x, s)
        if up:
            ref = ops.modconv_up(x, wt, s, dm, kf, noise, nw, bias, True, wt_planes=wtp)
            got = ops.modconv_up(x, wt, s, dm, kf, noise, nw, bias, True, wt_planes=wtp, x_planes=xp)
        else:
            ref = ops.modconv(x, wt, 3, s, dm, noise, nw, bias, True, wt_planes=wtp)
            got = ops.modconv(x, wt, 3, s, dm, noise, nw, bias, True, wt_planes=wtp, x_planes=xp)
        assert rel_err(got, ref) < (1e-6 if ops.conv_tiles(b, h, h + 1, cout, up) >= 256 else 5e-6)   # small layers: the
        # in-kernel-split path runs split-K there (another summation order)
        # the border of the planes buffer is still zero after the kernels ran
        planes = xp.view(-1, 3, b, h + 2, h + 3, 16)
        assert int(planes[:, :, :, 0].abs().max()) == 0 and int(planes[:, :, :, :, -1].abs().max()) == 0
    finally:
        ops.set_gemm_mode(prev)
