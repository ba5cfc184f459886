"""Thin torch-tensor front end of the C ABI (include/ganecdotes_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
computation below is a hand-written gfx950 kernel reached through ctypes.  No function in
this module has a CPU or eager-PyTorch fallback.
"""
import ctypes
import os
import math

import torch

from . import _lib

_WS = {}
PROFILE = None      # bench.py sets this to a dict: {tag: [(start_event, end_event, flops, bytes)]}


class record:
    """HIP-event bracket on the launch stream around one kernel (or a run of launches of the same
    kernel), with the algorithmic FLOPs / bytes it performs.  Inactive unless ops.PROFILE is a dict."""

    def __init__(self, tag, flops=0.0, nbytes=0.0, launches=1):
        self.tag, self.flops, self.nbytes, self.launches = tag, flops, nbytes, launches

    def __enter__(self):
        if PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.b.record()
            PROFILE.setdefault(self.tag, []).append((self.a, self.b, self.flops, self.nbytes, self.launches))
        return False


def lib():
    return _lib.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("ganecdotes_amd ops need HIP (cuda) tensors: no CPU fallback exists")
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected float32, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("expected a contiguous tensor")


def workspace(tag, nbytes, device):
    """Grow-only scratch arena per (tag, device, current stream): the users of one arena are ordered on one stream (the views
    of the next batch run on their own stream beside the step, ViewMixin.prefetch_views: the two must not share scratch)."""
    key = (tag, str(device), torch.cuda.current_stream(device).cuda_stream)
    n = max(1, (int(nbytes) + 3) // 4)
    t = _WS.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(int(n * 1.25) + 64, dtype=torch.float32, device=device)
        _WS[key] = t
    return t


# ------------------------------------------------------------------------------------- GEMM
def set_gemm_mode(mode):
    """0: exact-fp32 MFMA; 1: bf16x6 on the bf16 matrix cores.  Returns the previous mode."""
    return lib().gx_set_gemm_mode(int(mode))


def get_gemm_mode():
    return lib().gx_get_gemm_mode()


RANGE_HINT = ("a value left the fp16 range of the f16x3 contractions (|activation * style| < 32752 in the modulated convs, the "
              "stated bounds of the weight-plane GEMMs) or the computation diverged: the matrix cores then return NaN instead "
              "of a wrong number.  GX_CONV_PLANES=0 GX_GEMM_PLANES=0 select the bf16x6 forms, which have fp32's range")


def check_finite(x, what):
    """Raise FloatingPointError when `x` (a HIP tensor or a float already read back) holds a NaN / inf — the loud end of the
    f16x3 range contract.  A tensor costs one reduction and one read-back: call it where the host waits anyway."""
    ok = math.isfinite(x) if isinstance(x, float) else bool(torch.isfinite(x).all())
    if not ok:
        raise FloatingPointError(f"{what}: not finite — {RANGE_HINT}")


def set_conv_planes(fmt):
    """Plane format of the x-planes conv path (engine 1): 0 = three bf16 planes / six products (bf16x6), 1 = two fp16 planes /
    three products (f16x3).  Frozen weight planes are re-made when the format changes (Generator._prepare).  Returns the
    previous format."""
    return lib().gx_set_conv_planes(int(fmt))


def get_conv_planes():
    return lib().gx_get_conv_planes()


def _chk2d(*ts):
    """2-D fp32 HIP tensors with unit inner stride (row-strided views such as W[:, a:b] are fine)."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError("expected a 2-D float32 HIP tensor with unit inner stride")


def gemm(a, b, trans_a=False, trans_b=True, out=None, alpha=1.0, beta=0.0, row_scale=None, col_bias=None,
         split_k=1, live=None, live_frac=1.0, a_range=None, b_range=None, a_amax=None, b_amax=None):
    """out = alpha * row_scale[:,None] * op(a) @ op(b) + col_bias + beta*out on the fp32 MFMA.
    a_range / b_range (both, optional): upper bounds of |a|, |b| — the long-k launches then run the f16x3 form
    (gx_gemm_f32_ranged: half the matrix-core work; each bound within ~2^14 of its operand's RMS).  a_amax: a one-element HIP
    tensor holding max |a| (left by a's producer); a_range is then the factor it is multiplied by (default 1).  b_amax / b_range
    likewise.
    op(a) = a (M,K) or a.T for a stored (K,M); op(b) = b (K,N) or b.T for b stored (N,K).
    a, b, out may be row-strided views (leading dimension = stride(0)).
    live (trans_a only): int32 device scalar, the k rows that carry data (the rest is not read; the launch covers K with no
    host sync); live_frac = its expected share of K, for the FLOP accounting only."""
    _chk2d(a, b, out)
    _chk(row_scale, col_bias)
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    n, kb = (b.shape[0], b.shape[1]) if trans_b else (b.shape[1], b.shape[0])
    if k != kb:
        raise RuntimeError(f"gemm: inner dimensions differ ({k} vs {kb})")
    if out is None:
        if beta != 0.0:
            raise RuntimeError("gemm: beta != 0 needs `out`")
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    elif tuple(out.shape) != (m, n):
        raise RuntimeError("gemm: bad `out` shape")
    L = lib()
    ws, wsb = None, 0
    if split_k > 1:
        wsb = L.gx_gemm_f32_workspace_bytes(m, n, split_k)
        ws = workspace("gemm", wsb, a.device)
    if live is not None and (not trans_a or live.dtype != torch.int32 or not live.is_cuda or live.numel() != 1):
        raise RuntimeError("gemm: `live` is an int32 HIP scalar and needs trans_a")
    if a_amax is not None and a_range is None:
        a_range = 1.0
    if b_amax is not None and b_range is None:
        b_range = 1.0
    if GEMM_H2 and a_range is not None and b_range is not None and a_range > 0 and b_range > 0:
        with record("gemm_f32_kernel", 2.0 * m * n * k * (live_frac if live is not None else 1.0)):
            _lib.check(L.gx_gemm_f32_ranged(int(trans_a), int(trans_b), m, n, k, _p(a), a.stride(0), _p(b), b.stride(0),
                                            _p(out), out.stride(0), alpha, beta, _p(row_scale), _p(col_bias), split_k, _p(ws),
                                            wsb, _p(live), float(a_range), float(b_range), _p(a_amax), _p(b_amax), _stream()),
                       "gx_gemm_f32_ranged")
        return out
    if live is not None:
        with record("gemm_f32_kernel", 2.0 * m * n * k * live_frac):
            _lib.check(L.gx_gemm_f32_klive(int(trans_a), int(trans_b), m, n, k, _p(a), a.stride(0), _p(b), b.stride(0),
                                           _p(out), out.stride(0), alpha, beta, _p(row_scale), _p(col_bias), split_k, _p(ws),
                                           wsb, _p(live), _stream()), "gx_gemm_f32_klive")
        return out
    with record("gemm_f32_kernel", 2.0 * m * n * k):
        _lib.check(L.gx_gemm_f32(int(trans_a), int(trans_b), m, n, k, _p(a), a.stride(0), _p(b), b.stride(0),
                                 _p(out), out.stride(0), alpha, beta, _p(row_scale), _p(col_bias), split_k, _p(ws),
                                 wsb, _stream()), "gx_gemm_f32")
    return out


def gemm_cat(a1, a2, b, out, trans_b=True, alpha=1.0, beta=0.0):
    """out = alpha * [a1 | a2] @ op(b) + beta*out; a1 (M,K1), a2 (M,K2), K1 % 32 == 0."""
    _chk2d(a1, a2, b, out)
    m, k1 = a1.shape
    k2 = a2.shape[1]
    n = b.shape[0] if trans_b else b.shape[1]
    if a2.shape[0] != m or tuple(out.shape) != (m, n) or (b.shape[1] if trans_b else b.shape[0]) != k1 + k2:
        raise RuntimeError("gemm_cat: shape mismatch")
    with record("gemm_f32_kernel", 2.0 * m * n * (k1 + k2)):
        _lib.check(lib().gx_gemm_f32_cat(int(trans_b), m, n, k1, _p(a1), a1.stride(0), k2, _p(a2), a2.stride(0),
                                         _p(b), b.stride(0), _p(out), out.stride(0), alpha, beta, None, _stream()),
                   "gx_gemm_f32_cat")
    return out


def gemm_upadd(a1, a2, b, out, coarse, fine_res, beta=0.0):
    """out = [a1 | a2] @ b.T + beta*out + upsample2(coarse): one pyramid level in one launch.  out's rows are the
    pixels of fine_res x fine_res images, coarse (rows/4, N) the finished level below; a2 may be None."""
    _chk2d(a1, a2, b, out, coarse)
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = b.shape[0]
    if tuple(out.shape) != (m, n) or b.shape[1] != k1 + k2 or tuple(coarse.shape) != (m // 4, n) or coarse.stride(0) != n:
        raise RuntimeError("gemm_upadd: shape mismatch")
    with record("gemm_f32_kernel", 2.0 * m * n * (k1 + k2)):
        _lib.check(lib().gx_gemm_f32_upadd(1, m, n, k1, _p(a1), a1.stride(0), k2, _p(a2), 0 if a2 is None else a2.stride(0),
                                           _p(b), b.stride(0), _p(out), out.stride(0), beta, _p(coarse), fine_res,
                                           _stream()), "gx_gemm_f32_upadd")
    return out


def gemm_tn_cat(a, b1, b2, out, alpha=1.0, beta=0.0, split_k=1, a_amax=None, a_mul=1.0, b_range=None):
    """out (M, N1+N2) = alpha * a.T @ [b1 | b2] + beta*out for a stored (K, M), b1 (K, N1), b2 (K, N2); N1 % 128 == 0.
    a_amax (one-element HIP tensor: max |a| / a_mul) and b_range: the bounds of the f16x3 form (gemm's a_amax / b_range)."""
    _chk2d(a, b1, b2, out)
    k, m = a.shape
    n1, n2 = b1.shape[1], b2.shape[1]
    if b1.shape[0] != k or b2.shape[0] != k or tuple(out.shape) != (m, n1 + n2):
        raise RuntimeError("gemm_tn_cat: shape mismatch")
    L = lib()
    ws, wsb = None, 0
    if split_k > 1:
        wsb = L.gx_gemm_f32_workspace_bytes(m, n1 + n2, split_k)
        ws = workspace("gemm", wsb, a.device)
    if GEMM_H2 and a_amax is not None and b_range is not None:
        with record("gemm_f32_kernel", 2.0 * m * (n1 + n2) * k):
            _lib.check(L.gx_gemm_f32_tn_cat_ranged(m, n1, n2, k, _p(a), a.stride(0), _p(b1), b1.stride(0), _p(b2), b2.stride(0),
                                                   _p(out), out.stride(0), alpha, beta, split_k, _p(ws), wsb, _p(a_amax),
                                                   float(a_mul), float(b_range), _stream()), "gx_gemm_f32_tn_cat_ranged")
        return out
    with record("gemm_f32_kernel", 2.0 * m * (n1 + n2) * k):
        _lib.check(L.gx_gemm_f32_tn_cat(m, n1, n2, k, _p(a), a.stride(0), _p(b1), b1.stride(0), _p(b2), b2.stride(0), _p(out),
                                        out.stride(0), alpha, beta, split_k, _p(ws), wsb, _stream()), "gx_gemm_f32_tn_cat")
    return out


FEATURE_RANGE = 4096.0       # the a_range of generator taps: the x-planes convs' own contract (|x * style| < 32752, x 2 = scale 2)
GEMM_H2 = os.environ.get("GX_GEMM_PLANES", "1") != "0"     # the f16x3 form of the weight-plane GEMMs where the caller knows |A|'s range


class WeightPlanes:
    """Planes of a weight operand (N, K) for gemm_nt_planes: .buf (int16), .h2 (False: three bf16 planes, bf16x6; True: two
    fp16 planes scaled on the device + the max-|B| trailer, f16x3)"""
    __slots__ = ("buf", "h2")

    def __init__(self, buf, h2):
        self.buf, self.h2 = buf, h2

    def numel(self):
        return self.buf.numel()


def split_gemm_weights(b, h2=False):
    """Planes of the weight operand b (N, K) of gemm_nt_planes (row-strided views allowed).  Made once per optimizer step: the
    projection / prototype weights only change there.  h2: the f16x3 form (a call that passes a_range) — honoured only while
    ops.GEMM_H2 (env GX_GEMM_PLANES=0 switches every such call back to bf16x6)."""
    _chk2d(b)
    n, k = b.shape
    L = lib()
    h2 = bool(h2 and GEMM_H2)
    planes = torch.empty(L.gx_gemm_weight_planes_bytes(n, k) // 2, dtype=torch.int16, device=b.device)
    if h2:
        _lib.check(L.gx_split_gemm_weights_h2(_p(b), n, k, b.stride(0), _p(planes), _stream()), "gx_split_gemm_weights_h2")
    else:
        _lib.check(L.gx_split_gemm_weights(_p(b), n, k, b.stride(0), _p(planes), _stream()), "gx_split_gemm_weights")
    return WeightPlanes(planes, h2)


def _a_range(b_planes, a_range, who, a_amax=None):
    """the a_range argument of the C call: 0 for bf16 planes; for f16x3 planes the caller's bound on |A| is mandatory — a host
    number, or a one-element HIP tensor a_amax (a_range is then the factor it is multiplied by, default 1)"""
    if not b_planes.h2:
        return 0.0
    if a_amax is not None and a_range is None:
        a_range = 1.0
    if a_range is None or not a_range > 0:
        raise RuntimeError(f"{who}: f16x3 weight planes need a_range, an upper bound of |A|")
    return float(a_range)


def absmax(x, sqrt=False, out=None):
    """one-element HIP tensor: max |x| of a 1-D / 2-D fp32 tensor (rows may be strided; a 1-D tensor any stride); sqrt: max sqrt |x|.
    The device-side bound of the f16x3 GEMM forms (a_amax / b_amax) where no producer leaves one behind."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() not in (1, 2) or (x.dim() == 2 and x.stride(1) != 1):
        raise RuntimeError("absmax: a 1-D, or a 2-D fp32 HIP tensor with unit inner stride")
    rows, cols, ld = (x.shape[0], 1, max(1, x.stride(0))) if x.dim() == 1 else (x.shape[0], x.shape[1], x.stride(0))
    out = torch.empty(1, dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(lib().gx_absmax(_p(x), rows, cols, ld, int(bool(sqrt)), _p(out), _stream()), "gx_absmax")
    return out


def gemm_planes_ok(a1, a2, n, out):
    """shapes gemm_nt_planes takes (bf16x6 engine, 16-channel chunks, 16-B aligned rows)"""
    if lib().gx_get_gemm_mode() != 1 or n % 4 or out.stride(0) % 4 or out.data_ptr() % 16:
        return False
    for a in (a1, a2):
        if a is not None and (a.shape[1] % 16 or a.stride(0) % 4 or a.data_ptr() % 16 or a.stride(0) >= (1 << 22)):
            return False
    return True


def gemm_nt_planes_live(a, b_planes, out, live, live_frac, col_bias=None, alpha=1.0, beta=0.0, a_range=None):
    """gemm_nt_planes(a, None, b_planes, out, col_bias=...) for an `a` whose rows carry data only below the int32 device scalar
    `live`: the 128-row tiles at or past it are skipped (their rows of `out` stay as they are); live_frac: expected share, for
    the FLOP accounting"""
    _chk2d(a, out)
    _chk(col_bias)
    m, k1 = a.shape
    n = out.shape[1]
    L = lib()
    if out.shape[0] != m or b_planes.numel() * 2 != L.gx_gemm_weight_planes_bytes(n, k1):
        raise RuntimeError("gemm_nt_planes_live: shape mismatch")
    if live.dtype != torch.int32 or not live.is_cuda or live.numel() != 1:
        raise RuntimeError("gemm_nt_planes_live: `live` is an int32 HIP scalar")
    with record("gemm_f32_kernel", 2.0 * m * n * k1 * live_frac):
        _lib.check(L.gx_gemm_f32_nt_planes_mlive(m, n, k1, _p(a), a.stride(0), _p(b_planes.buf), _p(out), out.stride(0), alpha, beta,
                                                 _p(col_bias), _p(live), _a_range(b_planes, a_range, "gemm_nt_planes_live"),
                                                 _stream()), "gx_gemm_f32_nt_planes_mlive")
    return out


def gemm_nt_planes(a1, a2, b_planes, out, alpha=1.0, beta=0.0, row_scale=None, col_bias=None, coarse=None, fine_res=0,
                   want_colmax=False, sk_eps=None, a_range=None, a_amax=None):
    """out = alpha * row_scale[:,None] * [a1 | a2] @ B.T + col_bias + beta*out (+ upsample2(coarse)) with B given as
    split_gemm_weights planes; a2 / row_scale / col_bias / coarse may be None.  Bitwise gemm / gemm_cat / gemm_upadd.
    want_colmax: also return out.amax(0), reduced from per-wave partials of the epilogue (no pass over out); an int g > 1:
    the maxima of g equal row groups, (g, N) (rows per group a multiple of 32).
    sk_eps (with want_colmax): also return the FIRST Sinkhorn pass, u = exp((out - colmax) / sk_eps).sum(0) (per group), from
    per-slice sums of the same epilogue: (out, colmax, u).
    a_range (f16x3 planes): the bound of |a1|, |a2|; with a_amax (one-element HIP tensor: max |a| as its producer left it) the
    factor that multiplies it (default 1)."""
    _chk2d(a1, a2, out, coarse)
    _chk(row_scale, col_bias)
    m, k1 = a1.shape
    k2 = 0 if a2 is None else a2.shape[1]
    n = out.shape[1]
    if out.shape[0] != m or (a2 is not None and a2.shape[0] != m):
        raise RuntimeError("gemm_nt_planes: shape mismatch")
    if coarse is not None and (tuple(coarse.shape) != (m // 4, n) or coarse.stride(0) != n):
        raise RuntimeError("gemm_nt_planes: bad coarse table")
    L = lib()
    if b_planes.numel() * 2 != L.gx_gemm_weight_planes_bytes(n, k1 + k2):
        raise RuntimeError("gemm_nt_planes: planes do not match (N, K)")
    part = upart = None
    if sk_eps is not None and not want_colmax:
        raise RuntimeError("gemm_nt_planes: sk_eps needs want_colmax")
    if want_colmax:
        pb = L.gx_gemm_colmax_part_bytes(m, n)
        part = workspace("gemm_colmax", pb, a1.device)[:pb // 4]
        if sk_eps is not None:
            upart = workspace("gemm_usum", pb, a1.device)[:pb // 4]
    with record("gemm_f32_kernel", 2.0 * m * n * (k1 + k2)):
        _lib.check(L.gx_gemm_f32_nt_planes(m, n, k1, _p(a1), a1.stride(0), k2, _p(a2), 0 if a2 is None else a2.stride(0),
                                           _p(b_planes.buf), _p(out), out.stride(0), alpha, beta, _p(row_scale), _p(col_bias),
                                           _p(coarse), fine_res, _p(part), _p(upart), 0.0 if sk_eps is None else float(sk_eps),
                                           _a_range(b_planes, a_range, "gemm_nt_planes", a_amax),
                                           _p(a_amax) if b_planes.h2 else None, _stream()), "gx_gemm_f32_nt_planes")
    if want_colmax:
        g = 1 if want_colmax is True else int(want_colmax)
        if g > 1 and m % (32 * g):
            raise RuntimeError("gemm_nt_planes: row groups must be whole 32-row slices")
        if sk_eps is not None:          # column maxima and first-pass sums of every group, merged by log-sum-exp
            cm = torch.empty(g, n, dtype=torch.float32, device=out.device)
            u = torch.empty(g, n, dtype=torch.float32, device=out.device)
            wsb = L.gx_sinkhorn_first_from_parts_workspace_bytes(m, n)
            ws = workspace("sk_first", wsb, out.device)
            _lib.check(L.gx_sinkhorn_first_from_parts(_p(part), _p(upart), m, n, g, float(sk_eps), _p(cm), _p(u), _p(ws), wsb,
                                                      _stream()), "gx_sinkhorn_first_from_parts")
            return (out, cm[0], u[0]) if g == 1 else (out, cm, u)
        if g > 1:
            return out, part.view(-1, n)[:m // 32].view(g, m // (32 * g), n).amax(1)
        return out, part.view(-1, n).amax(0)
    return out


# ------------------------------------------------------------------------------------- pyramid projection
def upsample2_add_(coarse, fine):
    """fine (B, 2r, 2r, D) += nearest-upsampled coarse (B, r, r, D)."""
    _chk(coarse, fine)
    b, r, _, d = coarse.shape
    _lib.check(lib().gx_upsample2_add(_p(coarse), _p(fine), b, r, d, _stream()), "gx_upsample2_add")
    return fine


def sumpool2(fine):
    """(B, 2r, 2r, D) -> (B, r, r, D) sums of 2x2 blocks (adjoint of upsample2_add_)."""
    _chk(fine)
    b, rf, _, d = fine.shape
    out = torch.empty(b, rf // 2, rf // 2, d, dtype=torch.float32, device=fine.device)
    with record("sumpool2_kernel", 0.0, 5.0 * b * rf * rf * d):
        _lib.check(lib().gx_sumpool2(_p(fine), _p(out), b, rf // 2, d, _stream()), "gx_sumpool2")
    return out


def gather_rows_normalize(z, picks, maps=None):
    """z (B, H, W, D) table; picks (B, N) int32 -> zhat (B*N, D) L2-normalised rows, rnorm (B*N)."""
    _chk(z)
    b, h, w, d = z.shape
    n = picks.shape[1]
    zhat = torch.empty(b * n, d, dtype=torch.float32, device=z.device)
    rn = torch.empty(b * n, dtype=torch.float32, device=z.device)
    _lib.check(lib().gx_gather_rows_normalize(_p(z), b, h * w, d, _p(picks), n, _p(maps), _p(zhat), _p(rn), _stream()),
               "gx_gather_rows_normalize")
    return zhat, rn


def normalize_bwd_scatter_(dzhat, zhat, rnorm, picks, maps, g, amax=None):
    """g (B, H, W, D) += scatter of normalize-backward(dzhat) rows (float atomics).  amax (one-element HIP tensor the caller
    zeroed): raised to the largest single contribution (x the rows that can meet in one pixel: a bound of |g|)."""
    _chk(dzhat, zhat, rnorm, g)
    b, h, w, d = g.shape
    _lib.check(lib().gx_normalize_bwd_scatter(_p(dzhat), _p(zhat), _p(rnorm), b, h * w, d, _p(picks), picks.shape[1],
                                              _p(maps), _p(g), _p(amax), _stream()), "gx_normalize_bwd_scatter")
    return g


def proto_small_ok(k, d):
    return k <= 16 and d % 256 == 0 and d <= 512


def proto_fwd_small(z, picks, maps, wc, bias, out=None):
    """Fused row gather + normalise + scores for K <= 16 prototypes (scores into `out` if given)."""
    _chk(z, wc, bias, out)
    b, h, w, d = z.shape
    n, k = picks.shape[1], wc.shape[0]
    zhat = torch.empty(b * n, d, dtype=torch.float32, device=z.device)
    rn = torch.empty(b * n, dtype=torch.float32, device=z.device)
    sc = torch.empty(b * n, k, dtype=torch.float32, device=z.device) if out is None else out
    _lib.check(lib().gx_proto_fwd_small(_p(z), b, h * w, d, _p(picks), n, _p(maps), _p(wc), _p(bias), k, _p(zhat),
                                        _p(rn), _p(sc), _stream()), "gx_proto_fwd_small")
    return zhat, rn, sc


def proto_bwd_small_(ds, zhat, rnorm, picks, maps, wc, g, gwc, gb, beta):
    """Fused dz GEMM + dWc GEMM + db + normalise-backward + scatter for K <= 16 prototypes."""
    _chk(ds, zhat, rnorm, wc, g, gwc, gb)
    b, h, w, d = g.shape
    k = wc.shape[0]
    L = lib()
    wsb = L.gx_proto_small_workspace_bytes(k, d)
    ws = workspace("protosmall", wsb, g.device)
    _lib.check(L.gx_proto_bwd_small(_p(ds), _p(zhat), _p(rnorm), b, h * w, d, _p(picks), picks.shape[1], _p(maps),
                                    _p(wc), k, _p(g), _p(gwc), _p(gb), beta, _p(ws), wsb, _stream()),
               "gx_proto_bwd_small")


def lowrank_ok(k, d):
    return k <= 16 and d % 256 == 0 and d <= 512


def proto_fwd_lowrank(z, picks, maps, wc, bias, out=None):
    """scores (B*N, K), rnorm (B*N) from the projected-pixel table (zhat is never written).
    picks=None: the score / rnorm tables of every pixel in order (B*HW rows)."""
    _chk(z, wc, bias, out)
    b, h, w, d = z.shape
    n, k = (h * w if picks is None else picks.shape[1]), wc.shape[0]
    rn = torch.empty(b * n, dtype=torch.float32, device=z.device)
    sc = torch.empty(b * n, k, dtype=torch.float32, device=z.device) if out is None else out
    _lib.check(lib().gx_proto_fwd_lowrank(_p(z), b, h * w, d, _p(picks), n, _p(maps), _p(wc), _p(bias), k, _p(rn),
                                          _p(sc), _stream()), "gx_proto_fwd_lowrank")
    return rn, sc


def gather_scores(table, rn_table, bias, picks, maps, hw, out=None):
    """one patch's rows of the all-pixel score table: scores (B*N, K), rnorm (B*N)"""
    _chk(table, rn_table, bias, out)
    b, n = picks.shape
    k = table.shape[1]
    rn = torch.empty(b * n, dtype=torch.float32, device=table.device)
    sc = torch.empty(b * n, k, dtype=torch.float32, device=table.device) if out is None else out
    _lib.check(lib().gx_gather_scores(_p(table), _p(rn_table), _p(bias), b, hw, k, _p(picks), n, _p(maps), _p(sc),
                                      _p(rn), _stream()), "gx_gather_scores")
    return rn, sc


def gather_scores_batch(table, rn_table, bias, picks, maps, b, hw, out, shift=0):
    """all patches x both views: out (P, 2, B*N, K) scores, returns rnorm (P, 2, B*N); tables over the 2B batch,
    `shift` pyramid levels below the picks' resolution (2B * hw / 4**shift rows)"""
    _chk(table, rn_table, bias, out)
    p, pb, n = picks.shape
    k = table.shape[1]
    if table.shape[0] != 2 * b * (hw >> (2 * shift)) or rn_table.numel() != table.shape[0]:
        raise RuntimeError("gather_scores_batch: tables do not match (2B, hw, shift)")
    rn = torch.empty(p, 2, b * n, dtype=torch.float32, device=table.device)
    _lib.check(lib().gx_gather_scores_batch(_p(table), _p(rn_table), _p(bias), b, hw, k, _p(picks), p, n, _p(maps), shift,
                                            _p(out), _p(rn), _stream()), "gx_gather_scores_batch")
    return rn


def proto_bwd_lowrank_batch_(ds, scores, rnorm, bias, picks, maps, coeffs, b, hw, gb, beta, shift=0, coeffs_fix=None):
    """coeffs_fix (int64, coeffs' shape, zeroed once per step): the deterministic form — fixed-point accumulation, coeffs
    overwritten with its float value"""
    _chk(ds, scores, rnorm, bias, coeffs, gb)
    if coeffs_fix is not None and (coeffs_fix.dtype != torch.int64 or tuple(coeffs_fix.shape) != tuple(coeffs.shape) or
                                   not coeffs_fix.is_contiguous() or not coeffs_fix.is_cuda):
        raise RuntimeError("proto_bwd_lowrank_batch_: coeffs_fix must be a contiguous int64 HIP tensor of coeffs' shape")
    p, pb, n = picks.shape
    k = ds.shape[-1]
    if coeffs.shape[0] != 2 * b * (hw >> (2 * shift)):
        raise RuntimeError("proto_bwd_lowrank_batch_: coefficient grid does not match (2B, hw, shift)")
    ws = workspace("lowrank_b", 1024 * k * 4, ds.device)
    _lib.check(lib().gx_proto_bwd_lowrank_batch(_p(ds), _p(scores), _p(rnorm), _p(bias), b, hw, k, _p(picks), p, n, _p(maps),
                                                shift, _p(coeffs), _p(coeffs_fix), _p(gb), beta, _p(ws), 1024 * k * 4, _stream()),
               "gx_proto_bwd_lowrank_batch")


def swapped_ce_batch(scores, ngroups, skb, temperature, grad_scale, loss_out, loss_scale, loss_beta, d_scores):
    """scores / d_scores (P, 2, rows_per_patch, K); skb = the SinkhornBatch over its (P*2*ngroups) problems"""
    _chk(scores, d_scores, loss_out)
    p, _, rpp, k = scores.shape
    rows = rpp // ngroups
    wsb = p * ngroups * 512 * 4
    ws = workspace("ce_batch", wsb, scores.device)
    _lib.check(lib().gx_swapped_ce_batch(_p(scores), p, ngroups, rpp, rows, k, skb.eps, temperature, _p(skb.la), grad_scale,
                                         _p(d_scores), loss_scale, loss_beta, _p(loss_out), _p(ws), wsb, _stream()),
               "gx_swapped_ce_batch")
    return loss_out


def lowrank_coeffs(b, hw, k, device):
    """zeroed coefficient grid C (B*HW, stride(K))"""
    return torch.zeros(b * hw, lib().gx_lowrank_stride(k), dtype=torch.float32, device=device)


def proto_bwd_lowrank_(ds, scores, rnorm, bias, picks, maps, coeffs, hw, gb, beta):
    _chk(ds, scores, rnorm, bias, coeffs, gb)
    k = ds.shape[1]
    b = picks.shape[0]
    ws = workspace("lowrank_b", 1024 * k * 4, ds.device)
    _lib.check(lib().gx_proto_bwd_lowrank(_p(ds), _p(scores), _p(rnorm), _p(bias), b, hw, k, _p(picks), picks.shape[1],
                                          _p(maps), _p(coeffs), _p(gb), beta, _p(ws), 1024 * k * 4, _stream()),
               "gx_proto_bwd_lowrank")


def proto_lowrank_finish(coeffs, z, wc, gwc, beta, want_pool=False, add_to=None):
    """G (B,H,W,D) rebuilt from the coefficient grid; gwc = beta*gwc + C^T Z.
    want_pool: also return sumpool2(G) (B,H/2,W/2,D), produced in the same pass.
    add_to: a (B,H,W,D) table the rebuilt rows are ADDED to (it becomes G)."""
    _chk(coeffs, z, wc, gwc, add_to)
    b, h, w, d = z.shape
    k = wc.shape[0]
    L = lib()
    if add_to is not None and tuple(add_to.shape) != tuple(z.shape):
        raise RuntimeError("proto_lowrank_finish: add_to must have the table's shape")
    g = torch.empty_like(z) if add_to is None else add_to
    pool = None
    if want_pool:
        if h != w or h % 2:
            raise RuntimeError("proto_lowrank_finish: pooled output needs a square, even-sized table")
        pool = torch.empty(b, h // 2, w // 2, d, dtype=torch.float32, device=z.device)
    wsb = L.gx_proto_small_workspace_bytes(k, d)
    ws = workspace("protosmall", wsb, z.device)
    npix = b * h * w
    passes = 2.0 + (0.25 if want_pool else 0.0) + (1.0 if add_to is not None else 0.0)
    with record("proto_lowrank_finish_kernel", 0.0, 4.0 * npix * (d * passes + coeffs.shape[1])):
        _lib.check(L.gx_proto_lowrank_finish(_p(coeffs), _p(z), npix, h // 2, d, _p(wc), k, _p(g), int(add_to is not None),
                                             _p(pool), _p(gwc), beta, _p(ws), wsb, _stream()), "gx_proto_lowrank_finish")
    return (g, pool) if want_pool else g


def upsample2_add_scores_(coarse, fine, wc, bias):
    """fine (B,2r,2r,D) += upsample2(coarse (B,r,r,D)) and, in the same pass, the prototype score
    table (B*4r^2, K) and 1/||row|| table (B*4r^2) of the finished rows.  coarse=None: fine is
    already finished (gemm_upadd); only the tables are computed."""
    _chk(coarse, fine, wc, bias)
    b, r, d = fine.shape[0], fine.shape[1] // 2, fine.shape[3]
    k = wc.shape[0]
    rn = torch.empty(b * 4 * r * r, dtype=torch.float32, device=fine.device)
    sc = torch.empty(b * 4 * r * r, k, dtype=torch.float32, device=fine.device)
    npix = b * 4 * r * r
    with record("upsample2_add_scores_kernel", 0.0, 4.0 * npix * ((d if coarse is None else 2.25 * d) + k + 1)):
        _lib.check(lib().gx_upsample2_add_scores(_p(coarse), _p(fine), b, r, d, _p(wc), _p(bias), k, _p(rn), _p(sc),
                                                 _stream()), "gx_upsample2_add_scores")
    return rn, sc


# ---- the finest level of the low-rank path without its projected table (csrc/gx_fineq.hip) --------------------------
def fineq_ok(k1, k2, k):
    """taps (k1 + k2 channels) and prototype count the quadratic-form finest level takes (bf16x6 engine only)"""
    return get_gemm_mode() == 1 and bool(lib().gx_fineq_ok(int(k1), int(k2), int(k)))


FINEQ_NC_COL = 272          # the |Zc|^2 column of a fineq_ytable


def fineq_ytable(rows_coarse, device):
    """the coarse table of fineq_fwd: (rows/4, ld) fp32 — columns [0, 256) Y2 = 2 Zc Wf, [256, 256 + K) Ac = Zc Wc^T,
    272 nc = |Zc|^2 — so that every share of a parent pixel reaches the kernel as a 128-B column block of one row"""
    # (not zeroed: the pad columns travel to LDS with their block but nothing reads them)
    return torch.empty(rows_coarse, lib().gx_fineq_ytable_ld(), dtype=torch.float32, device=device)


def proto_raw_tables(z2d, wc, ytable=None):
    """z2d (rows, D) finished table -> (sumsq (rows,), dots (rows, K) = z2d @ wc.T): no normalisation, no bias.
    ytable: write them into columns 272 and [256, 256 + K) of a fineq_ytable instead (returns views of it)"""
    _chk(z2d, wc, ytable)
    rows, d = z2d.shape
    k = wc.shape[0]
    if ytable is None:
        ss = torch.empty(rows, dtype=torch.float32, device=z2d.device)
        dots = torch.empty(rows, k, dtype=torch.float32, device=z2d.device)
    else:
        if ytable.shape[0] != rows or ytable.shape[1] < 273:
            raise RuntimeError("proto_raw_tables: ytable does not match")
        ss, dots = ytable[:, 272], ytable[:, 256:256 + k]
    with record("proto_raw_tables_kernel", 0.0, 4.0 * rows * (d + k + 1)):
        _lib.check(lib().gx_proto_raw_tables(_p(z2d), rows, d, _p(wc), k, _p(ss), ss.stride(0), _p(dots), dots.stride(0),
                                             _stream()), "gx_proto_raw_tables")
    return ss, dots


def fineq_pack_weights(mg, wcf):
    """Bq (CF + 16, CF) = [tri(mg) ; wcf ; 0] for fineq_fwd: mg = Wf.T @ Wf (CF, CF), wcf = Wc @ Wf (K, CF)"""
    _chk(mg, wcf)
    cf, k = mg.shape[0], wcf.shape[0]
    bq = torch.empty(cf + 16, cf, dtype=torch.float32, device=mg.device)
    _lib.check(lib().gx_fineq_pack_weights(_p(mg), _p(wcf), cf, k, _p(bq), _stream()), "gx_fineq_pack_weights")
    return bq


def fineq_fwd(t1, t2, bq_planes, ytable, bias, k, fine_res, a_range=None, want_amax=False):
    """the finest level's score / rnorm tables from its taps t1 (rows, K1) [| t2 (rows, K2)] and the coarse table of the
    level below (fineq_ytable: y2 = 2 Zc Wf | ac = Zc Wc^T | nc = |Zc|^2): returns (rn_table (rows,), sc_table (rows, K))"""
    _chk2d(t1, t2)
    _chk(ytable, bias)
    rows, k1 = t1.shape
    k2 = 0 if t2 is None else t2.shape[1]
    L = lib()
    if tuple(ytable.shape) != (rows // 4, L.gx_fineq_ytable_ld()) or (t2 is not None and t2.shape[0] != rows):
        raise RuntimeError("fineq_fwd: shape mismatch")
    if bq_planes.numel() * 2 != L.gx_gemm_weight_planes_bytes(k1 + k2 + 16, k1 + k2):
        raise RuntimeError("fineq_fwd: planes do not match (CF + 16, CF)")
    rn = torch.empty(rows, dtype=torch.float32, device=t1.device)
    sc = torch.empty(rows, k, dtype=torch.float32, device=t1.device)
    t_amax = torch.empty(1, dtype=torch.float32, device=t1.device) if want_amax else None      # max |tap| as the pass met them
    cf = k1 + k2
    flops = 2.0 * rows * 16 * 32 * (sum(cf // 32 - (t >> 1) for t in range(cf // 16)) + cf // 16)   # blocks on / below the diagonal + scores
    with record("fineq_fwd_kernel", flops):
        _lib.check(L.gx_fineq_fwd(rows, fine_res, k1, _p(t1), t1.stride(0), k2, _p(t2), 0 if t2 is None else t2.stride(0),
                                  _p(bq_planes.buf), _p(ytable), _p(bias), k, _p(rn), _p(sc), _a_range(bq_planes, a_range, "fineq_fwd"),
                                  _p(t_amax), _stream()), "gx_fineq_fwd")
    return (rn, sc, t_amax) if want_amax else (rn, sc)


def lowrank_coeff_amax(coeffs, k):
    """(2,) HIP tensor: max |coeffs[:, :k]|, max |coeffs[:, k]| of a low-rank coefficient grid"""
    _chk(coeffs)
    out = torch.empty(2, dtype=torch.float32, device=coeffs.device)
    _lib.check(lib().gx_lowrank_coeff_amax(_p(coeffs), coeffs.shape[0], k, _p(out), _stream()), "gx_lowrank_coeff_amax")
    return out


def fineq_bwd(coeffs, t1, t2, k, fine_res, want_amax=False, t_amax=None):
    """coefficient grid (rows, stride) + the finest taps, ONE pass over the taps -> S (256, 256) = sum_q e[q] t t^T,
    P (K, 256) = C[:, :K]^T t, Tt (rows/4, 256), Cpool (rows/4, stride) (+ max |Tt| as a one-element HIP tensor with want_amax).
    t_amax (one-element HIP tensor: max |tap|, fineq_fwd's): the f16x3 form of the pass, the coefficient bounds found here."""
    _chk(coeffs)
    _chk2d(t1, t2)
    rows, k1 = t1.shape
    k2 = 0 if t2 is None else t2.shape[1]
    cf = k1 + k2
    L = lib()
    if coeffs.shape[0] != rows or coeffs.shape[1] != L.gx_lowrank_stride(k) or cf != 256:
        raise RuntimeError("fineq_bwd: coefficient grid / taps do not match (rows, K, 256 channels)")
    dev = t1.device
    s = torch.empty(cf, cf, dtype=torch.float32, device=dev)
    pm = torch.empty(k, cf, dtype=torch.float32, device=dev)
    tt = torch.empty(rows // 4, cf, dtype=torch.float32, device=dev)
    cpool = torch.empty(rows // 4, coeffs.shape[1], dtype=torch.float32, device=dev)
    wsb = L.gx_fineq_bwd_workspace_bytes()
    ws = workspace("fineq_bwd", wsb, dev)
    tt_amax = torch.empty(1, dtype=torch.float32, device=dev) if want_amax else None
    c_amax = lowrank_coeff_amax(coeffs, k) if t_amax is not None else None
    # the 10 of 16 64x64 blocks of S on / above the diagonal + the 32 coefficient rows (K of them live)
    with record("fineq_bwd_kernel", 2.0 * rows * 64 * 64 * 12, 4.0 * rows * (cf * 1.25 + coeffs.shape[1] * 1.25)):
        _lib.check(L.gx_fineq_bwd(_p(coeffs), rows, fine_res, k1, _p(t1), t1.stride(0), k2, _p(t2),
                                  0 if t2 is None else t2.stride(0), k, _p(s), _p(pm), _p(tt), _p(cpool), _p(ws), wsb,
                                  _p(tt_amax), _p(t_amax), _p(c_amax), _stream()), "gx_fineq_bwd")
    return (s, pm, tt, cpool, tt_amax) if want_amax else (s, pm, tt, cpool)


# ------------------------------------------------------------------------------------- generator pieces
def pixel_norm(x):
    _chk(x)
    y = torch.empty_like(x)
    _lib.check(lib().gx_pixel_norm(_p(x), x.shape[0], x.shape[1], _p(y), _stream()), "gx_pixel_norm")
    return y


def bias_act_(y, alpha=1.0, bias=None, bscale=1.0, act=True):
    _chk(y, bias)
    _lib.check(lib().gx_bias_act(_p(y), y.shape[0], y.shape[1], alpha, _p(bias), bscale, int(act), _stream()),
               "gx_bias_act")
    return y


def equal_linear_skinny(x, wt, alpha, bias, bscale, act=True):
    """act(alpha * x @ wt + bias * bscale) for a few rows x (rows, K); wt (K, N) = the EqualLinear weight transposed"""
    _chk(x, wt, bias)
    rows, k = x.shape
    out = torch.empty(rows, wt.shape[1], dtype=torch.float32, device=x.device)
    _lib.check(lib().gx_equal_linear_skinny(_p(x), rows, k, _p(wt), wt.shape[1], alpha, _p(bias), bscale, int(act), _p(out),
                                            _stream()), "gx_equal_linear_skinny")
    return out


def lerp_rows(w, base, psi):
    """base + psi * (w - base); base (1, D) or (D,) broadcast over the rows of w (..., D)."""
    _chk(w, base)
    out = torch.empty_like(w)
    _lib.check(lib().gx_lerp_rows(_p(w), _p(base), w.numel(), w.shape[-1], psi, _p(out), _stream()), "gx_lerp_rows")
    return out


def view_latents(w, mean, psi, nw, mix):
    """lat (B, n_latent, S): truncated w (B,S) with the perturbed latents mixed towards nw (B,2,S) by mix (B,n_latent,2)"""
    _chk(w, mean, nw, mix)
    b, s = w.shape
    nl = mix.shape[1]
    lat = torch.empty(b, nl, s, dtype=torch.float32, device=w.device)
    _lib.check(lib().gx_view_latents(_p(w), _p(mean), psi, _p(nw), _p(mix), b, nl, s, _p(lat), _stream()), "gx_view_latents")
    return lat


def style_pick(s_all, b, n_latent, table, nlayer):
    """per-layer styles (packed: layer j at [b*off_j, b*(off_j+cin_j)) as (b, cin_j)) from the
    all-latents x all-layers affine output s_all (b*n_latent, total)"""
    _chk(s_all)
    total = s_all.shape[1]
    out = torch.empty(b * total, dtype=torch.float32, device=s_all.device)
    _lib.check(lib().gx_style_pick(_p(s_all), b, n_latent, total, _p(table), nlayer, _p(out), _stream()), "gx_style_pick")
    return out


def demod_batch(s_packed, wsq_all, b, table, scale, nconv, total_rows):
    """demod' of every styled conv of a forward in one launch (packed like style_pick's output)"""
    _chk(s_packed, wsq_all, scale)
    out = torch.empty(b * total_rows, dtype=torch.float32, device=s_packed.device)
    _lib.check(lib().gx_demod_batch(_p(s_packed), _p(wsq_all), b, _p(table), _p(scale), nconv, total_rows, _p(out),
                                    _stream()), "gx_demod_batch")
    return out


def style_norm_(s_packed, demod_packed, b, table, nconv):
    """per (styled conv, sample), in place: s /= c, demod *= c with c the power of two that brings the sample's largest |s| into
    [2, 4) — exact, demod * conv(x * s) is unchanged; what the convs see of a style is then bounded whatever the checkpoint"""
    _chk(s_packed, demod_packed)
    _lib.check(lib().gx_style_norm(_p(s_packed), _p(demod_packed), b, _p(table), nconv, _stream()), "gx_style_norm")


def demod(s, wsq, scale):
    _chk(s, wsq)
    b, cin = s.shape
    cout = wsq.shape[0]
    d = torch.empty(b, cout, dtype=torch.float32, device=s.device)
    _lib.check(lib().gx_demod(_p(s), _p(wsq), b, cin, cout, scale, _p(d), _stream()), "gx_demod")
    return d


def _noise_args(noise, b, h, w):
    if noise is None:
        return None, 0
    _chk(noise)
    if noise.numel() == h * w:
        return noise, 0
    if noise.numel() == b * h * w:
        return noise, h * w
    raise RuntimeError("noise must be (1,1,H,W) or (B,1,H,W)")


def _conv_slab(L, b, oh, ow, cout, device):
    """split-K scratch, only for layers too small to fill the chip (< 256 output tiles)"""
    if ((b * oh * ow + 127) // 128) * ((cout + 127) // 128) >= 256:
        return None, 0
    nbytes = L.gx_modconv_slab_bytes(b, oh, ow, cout)
    return workspace("convslab", nbytes, device), nbytes


def split_conv_weights(wt):
    """bf16x6 planes of frozen conv weights wt (taps, Cout, Cin), tiled for the conv kernel (int16 buffer)"""
    _chk(wt)
    taps, cout, cin = wt.shape
    L = lib()
    planes = torch.empty(L.gx_conv_weight_planes_bytes(taps, cout, cin) // 2, dtype=torch.int16, device=wt.device)
    _lib.check(L.gx_split_conv_weights(_p(wt), taps, cout, cin, _p(planes), _stream()), "gx_split_conv_weights")
    return planes


_XPLANES = {}


def gemm_finish_ok(n, k1, res, k):
    return bool(lib().gx_gemm_finish_ok(n, k1, res, k))


def gemm_nt_planes_finish(a, b_planes, coeffs, z, wc, gwc, beta, res, want_pool, want_amax=False, a_range=None, a_amax=None):
    """G = a b^T with the low-rank path's dense finish in the GEMM's epilogue (gx_gemm_f32_nt_planes_finish):
    G += coeffs[:, :K] wc + coeffs[:, K] .* z, gwc = beta gwc + coeffs[:, :K]^T z, and the 2x2 sums of G.
    a (rows, K1) = pixels of res x res images, z (B, res, res, D); -> (G (B, res, res, D), Gpool (B, res/2, res/2, D) or None)
    a_range / a_amax with f16x3 planes (split_gemm_weights(h2=True)): as in gemm_nt_planes"""
    _chk(coeffs, z, wc, gwc)
    _chk2d(a)
    bsz, r, _, d = z.shape
    rows, k = bsz * r * r, wc.shape[0]
    L = lib()
    if r != res or z.shape[2] != r or a.shape[0] != rows or tuple(wc.shape) != (k, d) or tuple(gwc.shape) != (k, d) or \
            tuple(coeffs.shape) != (rows, L.gx_lowrank_stride(k)):
        raise RuntimeError("gemm_nt_planes_finish: shape mismatch")
    if b_planes.numel() * 2 != L.gx_gemm_weight_planes_bytes(d, a.shape[1]):
        raise RuntimeError("gemm_nt_planes_finish: planes do not match (N, K)")
    if not L.gx_gemm_finish_ok(d, a.shape[1], r, k):
        raise RuntimeError("gemm_nt_planes_finish: not this shape / engine (gemm_finish_ok)")
    g = torch.empty_like(z)
    gp = torch.empty(bsz, r // 2, r // 2, d, dtype=torch.float32, device=z.device) if want_pool else None
    amax = torch.zeros(1, dtype=torch.float32, device=z.device) if want_amax else None     # max |G|, for the f16x3 GEMMs below
    wsb = L.gx_gemm_finish_workspace_bytes(rows, d, k)
    ws = workspace("gemm_finish", wsb, z.device)
    with record("gemm_f32_kernel", 2.0 * rows * d * a.shape[1], 0.0):
        _lib.check(L.gx_gemm_f32_nt_planes_finish(rows, d, a.shape[1], _p(a), a.stride(0), _p(b_planes.buf), _p(g), r, _p(coeffs),
                                                  _p(z), _p(wc), k, _p(gp), _p(gwc), beta, _p(ws), wsb, _p(amax),
                                                  _a_range(b_planes, a_range, "gemm_nt_planes_finish", a_amax),
                                                  _p(a_amax) if b_planes.h2 else None, _stream()),
                   "gx_gemm_f32_nt_planes_finish")
    return (g, gp, amax) if want_amax else (g, gp)


def xplanes_buffer(b, h, w, cin, device, slot="", cache=None):
    """A persistent x-planes buffer for (b, h, w, cin) (gx_xplanes_bytes; int16 view).  Its zero border is written
    here, once; kernels only ever write the interior, so the buffer is reused step after step.
    cache: the owner's dict (a Generator keeps its own: two generators, or one generator on two streams, must not
    share planes) holding ONE buffer per slot — a request with another shape (a different batch size) replaces it, so a
    process that alternates between batch sizes does not pin a full set of planes per size.  Without `cache` a
    module-wide dict keyed by the full shape is used (lab scripts).  Users of one cache are ordered on one stream."""
    shape = (b, h, w, cin, str(device))
    store, key = (_XPLANES, shape + (slot,)) if cache is None else (cache, slot)
    hit = store.get(key)
    if hit is not None and hit[0] == shape:
        return hit[1]
    buf = torch.zeros(lib().gx_xplanes_bytes(b, h, w, cin) // 2, dtype=torch.int16, device=device)
    store[key] = (shape, buf)
    return buf


def conv_tiles(b, h, w, cout, up):
    """128 x 128 output tiles of a styled conv on a (b, h, w, .) input: a plain 3x3 conv, or the four parity
    classes of the stride-2 transposed conv (one launch)"""
    ntn = (cout + 127) // 128
    if not up:
        return ((b * h * w + 127) // 128) * ntn
    return sum((b * oh * ow + 127) // 128 for oh in (h + 1, h) for ow in (w + 1, w)) * ntn


def conv_xp_ok(b, h, w, cout, cin, wt_planes, up=False):
    """the x-planes conv path applies: bf16x6 engine, pre-split weights, enough output tiles to fill the chip
    without split-K"""
    return wt_planes is not None and get_gemm_mode() == 1 and cin % 4 == 0 and conv_tiles(b, h, w, cout, up) >= 256


def modsplit(x, s, out=None):
    """x (B,H,W,Cin) NHWC * s (B,Cin) -> the three bf16 planes the _xp convs read (csrc/gx_conv.hip)."""
    _chk(x, s)
    b, h, w, cin = x.shape
    if out is None:
        out = xplanes_buffer(b, h, w, cin, x.device)
    with record("modsplit_kernel", 0.0, 10.0 * b * h * w * cin):
        _lib.check(lib().gx_modsplit(_p(x), b, h, w, cin, _p(s), _p(out), _stream()), "gx_modsplit")
    return out


def modconv(x, wt, ksize, s, dmod, noise, noise_w, bias, act=True, wt_planes=None, x_planes=None, next_style=None,
            next_planes=None):
    """x (B,H,W,Cin) NHWC -> (B,H,W,Cout) NHWC.  wt (k*k, Cout, Cin); wt_planes = split_conv_weights(wt) (optional);
    x_planes = modsplit(x, s) (optional: the pre-split modulated input, then x and s are not read).
    next_style (B,Cout) + next_planes (xplanes_buffer of the output shape): also leave out * next_style there as
    x planes for the next conv (from the conv's own epilogue on the x-planes path, else by one modsplit pass)."""
    _chk(x, wt, s, dmod, noise_w, bias, next_style)
    b, h, w, cin = x.shape
    cout = wt.shape[1]
    nz, nbs = _noise_args(noise, b, h, w)
    out = torch.empty(b, h, w, cout, dtype=torch.float32, device=x.device)
    L = lib()
    if x_planes is not None:
        with record("modconv_kernel", 2.0 * b * h * w * cin * cout * ksize * ksize):
            fuse = next_planes is not None and cout % 4 == 0
            _lib.check(L.gx_modconv_fwd_xp(_p(x_planes), b, h, w, cin, _p(wt), _p(wt_planes), ksize, cout, _p(dmod), _p(nz),
                                           nbs, _p(noise_w), _p(bias), int(act), _p(out), _p(next_style) if fuse else None,
                                           _p(next_planes) if fuse else None, _stream()), "gx_modconv_fwd_xp")
        if next_planes is not None and not fuse:
            modsplit(out, next_style, out=next_planes)
        return out
    slab, sbytes = _conv_slab(L, b, h, w, cout, x.device)
    with record("modconv_kernel", 2.0 * b * h * w * cin * cout * ksize * ksize):
        _lib.check(L.gx_modconv_fwd_p(_p(x), b, h, w, cin, _p(wt), _p(wt_planes), ksize, cout, _p(s), _p(dmod), _p(nz), nbs,
                                      _p(noise_w), _p(bias), int(act), _p(out), _p(slab), sbytes, _stream()),
                   "gx_modconv_fwd_p")
    if next_planes is not None:
        modsplit(out, next_style, out=next_planes)
    return out


def modconv_up(x, wt, s, dmod, kf, noise, noise_w, bias, act=True, wt_planes=None, x_planes=None, next_style=None,
               next_planes=None):
    """Upsampling StyledConv: (B,H,W,Cin) -> (B,2H,2W,Cout) NHWC.  x_planes / next_style / next_planes: as in
    modconv (the planes of the output come from the blur + activation pass that writes it)."""
    _chk(x, wt, s, dmod, kf, noise_w, bias, next_style)
    b, h, w, cin = x.shape
    cout = wt.shape[1]
    nz, nbs = _noise_args(noise, b, 2 * h, 2 * w)
    L = lib()
    scratch = workspace("upconv", L.gx_modconv_up_scratch_bytes(b, h, w, cout), x.device)
    out = torch.empty(b, 2 * h, 2 * w, cout, dtype=torch.float32, device=x.device)
    slab, sbytes = _conv_slab(L, b, 2 * h + 1, 2 * w + 1, cout, x.device)
    with record("modconv_kernel", 2.0 * b * h * w * cin * cout * 9):
        if x_planes is not None:
            _lib.check(L.gx_modconv_up_convt_xp(_p(x_planes), b, h, w, cin, _p(wt), _p(wt_planes), cout, _p(scratch),
                                                _stream()), "gx_modconv_up_convt_xp")
        else:
            _lib.check(L.gx_modconv_up_convt_p(_p(x), b, h, w, cin, _p(wt), _p(wt_planes), cout, _p(s), _p(scratch),
                                               _p(slab), sbytes, _stream()), "gx_modconv_up_convt_p")
    # T read once, the fp32 tap written once, and (when the next conv reads planes) the three bf16 planes of out * s_next
    tbytes = 4.0 * b * cout * ((2 * h + 1) * (2 * w + 1) + 4 * h * w) + (6.0 * b * cout * 4 * h * w if next_planes is not None else 0.0)
    with record("blur_act_kernel", 0.0, tbytes):
        _lib.check(L.gx_blur_act_fwd_xp(_p(scratch), b, 2 * h + 1, 2 * w + 1, cout, _p(kf), _p(dmod), _p(nz), nbs,
                                        _p(noise_w), _p(bias), int(act), _p(out), _p(next_style), _p(next_planes),
                                        _stream()), "gx_blur_act_fwd_xp")
    return out


def torgb(x, w3, s, scale, bias3, prev_img, kf):
    """x (B,H,W,Cin) NHWC -> image (B,3,H,W) planar."""
    _chk(x, w3, s, bias3, prev_img, kf)
    b, h, w, cin = x.shape
    img = torch.empty(b, 3, h, w, dtype=torch.float32, device=x.device)
    _lib.check(lib().gx_torgb_fwd(_p(x), b, h, w, cin, _p(w3), _p(s), scale, _p(bias3), _p(prev_img), _p(kf),
                                  _p(img), _stream()), "gx_torgb_fwd")
    return img


def nchw_to_nhwc(x):
    _chk(x)
    b, c, h, w = x.shape
    out = torch.empty(b, h, w, c, dtype=torch.float32, device=x.device)
    _lib.check(lib().gx_nchw_to_nhwc(_p(x), b, c, h * w, _p(out), _stream()), "gx_nchw_to_nhwc")
    return out


def upfirdn2d_native(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """The reference's native op with its own argument list (lib/gan/optim/upfirdn2d.cpp:18-35):
    x (major, in_h, in_w, minor), kernel (kh, kw) -> (major, out_h, out_w, minor).  Like the reference's
    CHECK_CUDA / CHECK_CONTIGUOUS (:9-15) a host or strided tensor raises RuntimeError."""
    _chk(x, kernel)
    if x.dim() != 4 or kernel.dim() != 2:
        raise RuntimeError("upfirdn2d: input must be (major, in_h, in_w, minor) and kernel (kh, kw)")
    major, in_h, in_w, minor = x.shape
    oh, ow = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib().gx_upfirdn2d_out_shape(in_h, in_w, kernel.shape[0], kernel.shape[1], up_x, up_y, down_x, down_y,
                                            pad_x0, pad_x1, pad_y0, pad_y1, ctypes.byref(oh), ctypes.byref(ow)),
               "gx_upfirdn2d_out_shape")
    out = torch.empty(major, oh.value, ow.value, minor, dtype=torch.float32, device=x.device)
    _lib.check(lib().gx_upfirdn2d(_p(x), major, in_h, in_w, minor, _p(kernel), kernel.shape[0], kernel.shape[1],
                                  up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, _p(out), _stream()),
               "gx_upfirdn2d")
    return out


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d(input (N, C, H, W), kernel, up, down, pad) of lib/gan/optim/upfirdn2d.py:146-173 (and of
    models/stylegan2/model.py:46-58): channels folded into `major`, minor = 1."""
    n, c, h, w = x.shape
    out = upfirdn2d_native(x.reshape(n * c, h, w, 1).contiguous(), kernel.contiguous(), up, up, down, down, pad[0],
                           pad[1], pad[0], pad[1])
    return out.reshape(n, c, out.shape[1], out.shape[2])


def fused_bias_act(x, bias=None, refer=None, act=3, grad=0, alpha=0.2, scale=1.0):
    """The reference's native op with its own argument list (lib/gan/optim/fused_bias_act.cpp:18-31): the bias
    is indexed by dim 1 of x (step_b = product of the later dims); an empty tensor / None = no bias / no refer."""
    bias = None if (bias is None or bias.numel() == 0) else bias
    refer = None if (refer is None or refer.numel() == 0) else refer
    _chk(x, bias, refer)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= int(d)
    out = torch.empty_like(x)
    _lib.check(lib().gx_fused_bias_act(_p(x), x.numel(), _p(bias), 0 if bias is None else bias.numel(), step_b,
                                       _p(refer), act, grad, alpha, scale, _p(out), _stream()), "gx_fused_bias_act")
    return out


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """fused_leaky_relu(input, bias, negative_slope, scale) of lib/gan/optim/fused_act.py:232-254 (forward)."""
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)


# ------------------------------------------------------------------------------------- gather
def rotflip_map(angle_deg, flip, h, w, device, out=None):
    if out is None:
        out = torch.empty(h * w, dtype=torch.int32, device=device)
    _lib.check(lib().gx_rotflip_map(float(angle_deg), int(bool(flip)), h, w, _p(out), _stream()), "gx_rotflip_map")
    return out


def random_subset_picks(keys, n):
    """keys (rows, HW) uniform fp32 in [0,1) -> (rows, n) int32: indices of the n smallest keys of every row
    (a uniform random n-subset in ascending pixel order) — radix select, no sort"""
    _chk(keys)
    rows, hw = keys.shape
    out = torch.empty(rows, n, dtype=torch.int32, device=keys.device)
    _lib.check(lib().gx_random_subset_picks(_p(keys), rows, hw, n, _p(out), _stream()), "gx_random_subset_picks")
    return out


def rotflip_map_batch(angles, flips, h, w, out):
    """out (n, h*w) int32: one launch for the maps of a whole view batch (None angle = 0 degrees)"""
    n = len(angles)
    a = (ctypes.c_float * n)(*[float(x or 0.0) for x in angles])
    f = (ctypes.c_int * n)(*[int(bool(x)) for x in flips])
    _lib.check(lib().gx_rotflip_map_batch(a, f, n, h, w, _p(out), _stream()), "gx_rotflip_map_batch")
    return out


class TapSet:
    """Host-side descriptor of the NHWC taps of a batch of views: device pointers + shapes,
    marshalled once per view batch for gx_gather_taps / gx_norm_map."""

    def __init__(self, taps):
        for t in taps:
            _chk(t)
            if t.dim() != 4 or t.shape[1] != t.shape[2]:
                raise RuntimeError("taps must be (B, r, r, C) NHWC")
        self.taps = taps
        n = len(taps)
        self.n = n
        self.B = taps[0].shape[0]
        self.ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in taps])
        self.chans = (ctypes.c_int * n)(*[t.shape[3] for t in taps])
        self.res = (ctypes.c_int * n)(*[t.shape[1] for t in taps])
        self.total_c = sum(t.shape[3] for t in taps)
        self.size = max(t.shape[1] for t in taps)


INTERP = {"nearest": 0, "bilinear": 1}      # swav_args['hf_interp'] -> gx_gather_taps' interp


def gather_taps(tapset, picks, hlen, maps=None, out=None, interp="nearest"):
    """picks (B, N) int32 pixel indices; maps (B, H*W) int32 or None -> X (B*N, hlen).  interp: F.interpolate's mode over the
    taps ('nearest' | 'bilinear')."""
    b, n = picks.shape
    if b != tapset.B or picks.dtype != torch.int32 or not picks.is_cuda or not picks.is_contiguous():
        raise RuntimeError("picks must be a contiguous (B, N) int32 HIP tensor")
    if maps is not None and (maps.dtype != torch.int32 or not maps.is_contiguous()):
        raise RuntimeError("maps must be contiguous int32")
    hw = tapset.size
    if out is None:
        out = torch.empty(b * n, hlen, dtype=torch.float32, device=picks.device)
    _lib.check(lib().gx_gather_taps(tapset.n, tapset.ptrs, tapset.chans, tapset.res, b, hw, hw, hlen, _p(picks), n,
                                    _p(maps), INTERP[interp], _p(out), out.stride(0), _stream()), "gx_gather_taps")
    return out


def row_sumsq(x2d, cused=None):
    _chk(x2d)
    rows, ld = x2d.shape
    out = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    _lib.check(lib().gx_row_sumsq(_p(x2d), rows, cused or ld, ld, _p(out), _stream()), "gx_row_sumsq")
    return out


def norm_map(tapset, hlen, maps=None, interp="nearest"):
    """(B, H, W) per-pixel L2 norm of the (rotated/flipped) first-hlen-channel stack."""
    if interp == "bilinear":
        hw = tapset.size
        img = torch.empty(tapset.B, hw, hw, dtype=torch.float32, device=tapset.taps[0].device)
        _lib.check(lib().gx_stack_norm_map_bilinear(tapset.n, tapset.ptrs, tapset.chans, tapset.res, hlen, _p(maps), tapset.B, hw,
                                                    hw, _p(img), _stream()), "gx_stack_norm_map_bilinear")
        return img
    ss, off = [], 0
    for t in tapset.taps:
        c = t.shape[3]
        if off >= hlen:
            break
        ss.append(row_sumsq(t.reshape(-1, c), min(c, hlen - off)))
        off += c
    n = len(ss)
    ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in ss])
    res = (ctypes.c_int * n)(*[t.shape[1] for t in tapset.taps[:n]])
    hw = tapset.size
    img = torch.empty(tapset.B, hw, hw, dtype=torch.float32, device=ss[0].device)
    _lib.check(lib().gx_norm_map(n, ptrs, res, _p(maps), tapset.B, hw, hw, _p(img), _stream()), "gx_norm_map")
    return img


# ------------------------------------------------------------------------------------- row helpers
def row_l2_normalize_(z, want_rnorm=True):
    _chk(z)
    rn = torch.empty(z.shape[0], dtype=torch.float32, device=z.device) if want_rnorm else None
    _lib.check(lib().gx_row_l2_normalize(_p(z), z.shape[0], z.shape[1], z.stride(0), _p(rn), _stream()),
               "gx_row_l2_normalize")
    return rn


def l2_normalize_bwd_(g, zhat, rnorm):
    _chk(g, zhat, rnorm)
    _lib.check(lib().gx_l2_normalize_bwd(_p(g), _p(zhat), _p(rnorm), g.shape[0], g.shape[1], g.stride(0), _stream()),
               "gx_l2_normalize_bwd")
    return g


def row_argmax(z):
    _chk(z)
    out = torch.empty(z.shape[0], dtype=torch.int64, device=z.device)
    _lib.check(lib().gx_row_argmax(_p(z), z.shape[0], z.shape[1], z.stride(0), _p(out), _stream()), "gx_row_argmax")
    return out


def colsum(x, out, alpha=1.0, beta=0.0):
    _chk2d(x)
    _chk(out)
    L = lib()
    k = x.shape[1]
    wsb = L.gx_colsum_workspace_bytes(k)
    ws = workspace("colsum", wsb, x.device)
    _lib.check(L.gx_colsum(_p(x), x.shape[0], k, x.stride(0), alpha, beta, _p(out), _p(ws), wsb, _stream()),
               "gx_colsum")
    return out


# ------------------------------------------------------------------------------------- Sinkhorn
class Sinkhorn:
    """Log-domain scaling-vector Sinkhorn-Knopp state for one score matrix (rows, K).

    After run(): `la` (K,) with codes q = softmax_k(S/eps + la).  Single device: `run()`.
    Sharded: pass `reduce(tensor, op)` that all-reduces in place over the ranks ('max' for the K
    column maxima, 'sum' for the K column sums) and c_uniform = 1/rows_global."""

    def __init__(self, scores, eps, niters, r=None, c=None, c_uniform=None, tag="sk", la=None, rowmap=None):
        """rowmap (int32 (rows,), many prototypes only): the problem's rows are rows rowmap[n] of `scores`, a table with more
        rows than the problem (the per-pixel score table of a view) — no gathered copy is made"""
        _chk(scores, r, c)
        self.S = scores
        self.rows, self.K = scores.shape
        self.rowmap = rowmap
        if rowmap is not None:
            if rowmap.dtype != torch.int32 or not rowmap.is_cuda or not rowmap.is_contiguous() or rowmap.dim() != 1:
                raise RuntimeError("Sinkhorn: rowmap must be a contiguous 1-D int32 HIP tensor")
            self.rows = rowmap.numel()
        self.eps, self.niters = float(eps), int(niters)
        self.r, self.c = r, c
        self.c_uniform = 1.0 / self.rows if c_uniform is None else float(c_uniform)
        self.tag = tag
        self.wsb = lib().gx_sinkhorn_workspace_bytes(self.K)
        self._ws = None
        self.la = torch.empty(self.K, dtype=torch.float32, device=scores.device) if la is None else la
        self.cmax_preset = None       # (K,) column maxima of `scores` when the producer already has them (gemm_nt_planes)
        self.u_first_preset = None    # (K,) exp((scores - cmax_preset) / eps).sum(0): the first pass, from the same producer

    @property
    def ws(self):
        if self._ws is None:
            self._ws = workspace(self.tag, self.wsb, self.S.device)
        return self._ws

    def run(self, reduce=None):
        L, S, K = lib(), self.S, self.K
        if self.rowmap is not None and reduce is None and self.cmax_preset is None:
            _lib.check(L.gx_sinkhorn_rows(_p(S), _p(self.rowmap), self.rows, K, S.stride(0), self.eps, self.niters, _p(self.r),
                                          _p(self.c), self.c_uniform, _p(self.la), _p(self.ws), self.wsb, _stream()),
                       "gx_sinkhorn_rows")
            return self
        if self.rowmap is None and reduce is None and self.c_uniform == 1.0 / self.rows and self.cmax_preset is None:
            if K <= 64 and self.rows >= 4096:
                _lib.check(L.gx_sinkhorn_small(_p(S), self.rows, K, S.stride(0), self.eps, self.niters, _p(self.r),
                                               _p(self.c), _p(self.la), _p(self.ws), self.wsb, _stream()),
                           "gx_sinkhorn_small")
                return self
            _lib.check(L.gx_sinkhorn(_p(S), self.rows, K, S.stride(0), self.eps, self.niters, _p(self.r), _p(self.c),
                                     _p(self.la), _p(self.ws), self.wsb, _stream()), "gx_sinkhorn")
            return self
        from .sharded import run_sinkhorn
        return run_sinkhorn(self, self.niters, reduce)

    # --- kernel object interface of sharded.run_sinkhorn ---
    def colmax(self):
        S = self.S
        if self.cmax_preset is not None:          # the pass over S is already paid for
            return self.cmax_set(self.cmax_preset)
        if self.rowmap is not None:
            _lib.check(lib().gx_sinkhorn_colmax_rows(_p(S), _p(self.rowmap), self.rows, self.K, S.stride(0), _p(self.ws),
                                                     self.wsb, _stream()), "gx_sinkhorn_colmax_rows")
            return
        _lib.check(lib().gx_sinkhorn_colmax(_p(S), self.rows, self.K, S.stride(0), _p(self.ws), self.wsb, _stream()),
                   "gx_sinkhorn_colmax")

    def cmax_get(self, out=None):
        cm = torch.empty(self.K, dtype=torch.float32, device=self.S.device) if out is None else out
        _lib.check(lib().gx_sinkhorn_cmax_get(_p(self.ws), self.K, _p(cm), _stream()), "gx_sinkhorn_cmax_get")
        return cm

    def cmax_set(self, cm):
        _lib.check(lib().gx_sinkhorn_cmax_set(_p(self.ws), self.K, _p(cm), _stream()), "gx_sinkhorn_cmax_set")

    def sweep(self, first):
        S = self.S
        if first and self.u_first_preset is not None and self.cmax_preset is not None:
            # the score GEMM left this pass behind, relative to ITS maxima: rescale to the maxima now in the workspace
            # (equal on one rank; after a cross-rank MAX all-reduce possibly larger)
            _lib.check(lib().gx_sinkhorn_first_preset(_p(self.u_first_preset), _p(self.cmax_preset), self.K, self.eps,
                                                      _p(self.ws), _stream()), "gx_sinkhorn_first_preset")
            return
        if self.rowmap is not None:
            _lib.check(lib().gx_sinkhorn_pass_rows(_p(S), _p(self.rowmap), self.rows, self.K, S.stride(0), self.eps, int(first),
                                                   _p(self.la), _p(self.c), self.c_uniform, None, _p(self.ws), self.wsb,
                                                   _stream()), "gx_sinkhorn_pass_rows")
            return
        _lib.check(lib().gx_sinkhorn_pass(_p(S), self.rows, self.K, S.stride(0), self.eps, int(first), _p(self.la),
                                          _p(self.c), self.c_uniform, None, _p(self.ws), self.wsb, _stream()),
                   "gx_sinkhorn_pass")

    def colsums(self):
        return self.ws[self.K:2 * self.K]

    def update(self, first, u=None):
        u = self.colsums() if u is None else u
        _lib.check(lib().gx_sinkhorn_update(_p(u), _p(self.r), self.K, int(first), _p(self.ws), self.eps,
                                            _p(self.la), _stream()), "gx_sinkhorn_update")

    def scores(self):
        """the problem's (rows, K) score matrix (a gathered copy when the problem lives on a table)"""
        return self.S if self.rowmap is None else self.S[self.rowmap.long()]

    def codes(self):
        S = self.scores()
        q = torch.empty(self.rows, self.K, dtype=torch.float32, device=S.device)
        _lib.check(lib().gx_sinkhorn_codes(_p(S), self.rows, self.K, S.stride(0), self.eps, _p(self.la),
                                           _p(q), q.stride(0), _stream()), "gx_sinkhorn_codes")
        return q


def sinkhorn_first_rows(scores, eps, rowmap=None):
    """(cmax (K,), u (K,)) of the Sinkhorn problem on `scores` (rows rowmap[n] of it when given): the column maxima and the
    first pass u = exp((S - cmax) / eps).sum(0) in ONE sweep (online log-sum-exp per block, merged exactly) — what
    Sinkhorn.cmax_preset / u_first_preset take"""
    _chk(scores)
    k = scores.shape[1]
    rows = scores.shape[0] if rowmap is None else rowmap.numel()
    L = lib()
    cm = torch.empty(k, dtype=torch.float32, device=scores.device)
    u = torch.empty(k, dtype=torch.float32, device=scores.device)
    wsb = L.gx_sinkhorn_first_rows_workspace_bytes(k)
    ws = workspace("sk_first_rows", wsb, scores.device)
    with record("sinkhorn_pass_kernel", 0.0, 4.0 * rows * k):
        _lib.check(L.gx_sinkhorn_first_rows(_p(scores), _p(rowmap), rows, k, scores.stride(0), eps, _p(cm), _p(u), _p(ws), wsb,
                                            _stream()), "gx_sinkhorn_first_rows")
    return cm, u


class SinkhornBatch:
    """`nprob` independent, equally shaped Sinkhorn problems — the 2 views x patches (x images for
    the per-image scope) score matrices of one optimizer step — solved together.

    scores (nprob, rows, K) contiguous; r (nprob, K) / c (nprob, rows) or None (uniform).
    One rank, K <= 64: ONE launch per Sinkhorn iteration for the whole batch.  Sharded: the same
    per-problem kernels as `Sinkhorn`, but ONE all-reduce of nprob*K values per exchange
    (sharded.run_sinkhorn_group) instead of one per problem.  self[i] is problem i's state."""

    def __init__(self, scores, eps, niters, r=None, c=None, c_uniform=None, tag="skb"):
        _chk(scores, r, c)
        if scores.dim() != 3:
            raise RuntimeError("SinkhornBatch: scores must be (nprob, rows, K)")
        self.S = scores
        self.nprob, self.rows, self.K = scores.shape
        self.eps, self.niters = float(eps), int(niters)
        self.r, self.c = r, c
        self.c_uniform = 1.0 / self.rows if c_uniform is None else float(c_uniform)
        self.tag = tag
        self.la = torch.empty(self.nprob, self.K, dtype=torch.float32, device=scores.device)
        self.probs = [Sinkhorn(scores[i], eps, niters, r=None if r is None else r[i], c=None if c is None else c[i],
                               c_uniform=self.c_uniform, tag=f"{tag}.{i}", la=self.la[i])
                      for i in range(self.nprob)]

    @classmethod
    def from_rows(cls, tables, rowmaps, eps, niters, c_uniform, tag="skb", sorted_rows=False, r=None, c=None):
        """problems given as (score table, row map) pairs: problem i = rows rowmaps[i] of tables[i] (many prototypes).
        sorted_rows: every row map ascends — the problems of one table are then solved side by side (gx_sinkhorn_rows_batch:
        one launch per pass, a table row several problems hold is fetched from HBM once).  r / c: lists of the problems'
        prototype (K,) and row (rows,) marginals, c in the order of rowmaps[i]; None = uniform"""
        self = cls.__new__(cls)
        self.S = None
        self.sorted_rows = bool(sorted_rows)
        if (r is None) != (c is None):
            raise RuntimeError("SinkhornBatch.from_rows: r and c come together")
        self.nprob, self.rows, self.K = len(tables), rowmaps[0].numel(), tables[0].shape[1]
        self.eps, self.niters = float(eps), int(niters)
        self.r = self.c = None
        self.c_uniform, self.tag = float(c_uniform), tag
        self.la = torch.empty(self.nprob, self.K, dtype=torch.float32, device=tables[0].device)
        self.probs = [Sinkhorn(tables[i], eps, niters, r=None if r is None else r[i], c=None if c is None else c[i],
                               c_uniform=self.c_uniform, tag=f"{tag}.{i}", la=self.la[i], rowmap=rowmaps[i])
                      for i in range(self.nprob)]
        return self

    def __getitem__(self, i):
        return self.probs[i]

    def run(self, reduce=None):
        if reduce is not None:
            return self._run(reduce)
        # one read of every score matrix per half-iteration (+ the column maxima unless the score GEMM left them behind)
        npass = self.niters + (0 if all(p.cmax_preset is not None for p in self.probs) else 1) - \
            (1 if all(p.u_first_preset is not None for p in self.probs) else 0)
        with record("sinkhorn_pass_kernel", 0.0, 4.0 * self.nprob * self.rows * self.K * npass, launches=npass):
            return self._run(None)

    def _run(self, reduce):
        if self.S is None:                    # row-mapped problems on score tables: the per-problem many-prototype kernels
            if reduce is None:
                by_table = {}
                for i, p in enumerate(self.probs):
                    by_table.setdefault((p.S.data_ptr(), p.S.stride(0)), []).append(i)
                chunks = [ids[i:i + 96] for ids in by_table.values() for i in range(0, len(ids), 96)]     # <= 96 problems a launch
                for idx in chunks:
                    grp = [self.probs[i] for i in idx]
                    npre = sum(p.cmax_preset is not None for p in grp)
                    if not (self.sorted_rows and len(grp) > 1 and npre in (0, len(grp))):
                        for p in grp:
                            p.run()
                        continue
                    L, S = lib(), grp[0].S
                    wsb = L.gx_sinkhorn_rows_batch_workspace_bytes(len(grp), self.rows, self.K)
                    ws = workspace(self.tag + ".rows", wsb, S.device)
                    maps = torch.stack([p.rowmap for p in grp])
                    # (no presets: the batch makes its own column maxima and first pass, one launch each over all problems)
                    cm = torch.stack([p.cmax_preset for p in grp]) if npre else None
                    uf = torch.stack([p.u_first_preset for p in grp]) if npre and all(p.u_first_preset is not None for p in grp) else None
                    la = torch.empty(len(grp), self.K, dtype=torch.float32, device=S.device)
                    r_all = c_all = None
                    if grp[0].r is not None:
                        r_all, c_all = torch.stack([p.r for p in grp]), torch.stack([p.c for p in grp])
                    _lib.check(L.gx_sinkhorn_rows_batch(_p(S), _p(maps), len(grp), self.rows, self.K, S.stride(0), self.eps,
                                                        self.niters, _p(r_all), _p(c_all), self.c_uniform, _p(cm), _p(uf),
                                                        _p(la), _p(ws), wsb, _stream()), "gx_sinkhorn_rows_batch")
                    self.la[idx] = la
                return self
            from .sharded import run_sinkhorn_group
            run_sinkhorn_group(self.probs, self.niters, reduce)
            return self
        if reduce is None and self.c_uniform == 1.0 / self.rows and self.K <= 64 and self.rows >= 4096:
            L = lib()
            wsb = self.nprob * L.gx_sinkhorn_workspace_bytes(self.K)
            ws = workspace(self.tag, wsb, self.S.device)
            _lib.check(L.gx_sinkhorn_small_batch(
                _p(self.S), self.nprob, self.S.stride(0), self.rows, self.K, self.S.stride(1), self.eps, self.niters,
                _p(self.r), 0 if self.r is None else self.r.stride(0), _p(self.c),
                0 if self.c is None else self.c.stride(0), _p(self.la), _p(ws), wsb, _stream()),
                "gx_sinkhorn_small_batch")
            return self
        if reduce is None:
            S = self.S
            npre = sum(p.cmax_preset is not None for p in self.probs)
            if (self.nprob >= 16 and 64 < self.K <= 8192 and self.K % 4 == 0 and self.c_uniform == 1.0 / self.rows and
                    npre in (0, self.nprob) and S.stride(1) % 4 == 0 and S.stride(0) % 4 == 0 and S.data_ptr() % 16 == 0):
                # many many-prototype problems (the per-image scope at K = 5000): one launch per pass for all of them
                L = lib()
                wsb = L.gx_sinkhorn_bigk_batch_workspace_bytes(self.nprob, self.rows, self.K)
                ws = workspace(self.tag + ".bigk", wsb, S.device)
                cm = torch.stack([p.cmax_preset for p in self.probs]).contiguous() if npre else None
                uf = None
                if npre and all(p.u_first_preset is not None for p in self.probs):
                    uf = torch.stack([p.u_first_preset for p in self.probs]).contiguous()
                _lib.check(L.gx_sinkhorn_bigk_batch(
                    _p(S), self.nprob, S.stride(0), self.rows, self.K, S.stride(1), self.eps, self.niters, _p(self.r),
                    0 if self.r is None else self.r.stride(0), _p(self.c), 0 if self.c is None else self.c.stride(0),
                    _p(cm), _p(uf), _p(self.la), _p(ws), wsb, _stream()), "gx_sinkhorn_bigk_batch")
                return self
            for p in self.probs:
                p.run()
            return self
        if self.K <= 64 and self.rows >= 4096:
            return self._run_sharded_small(reduce)
        from .sharded import run_sinkhorn_group
        run_sinkhorn_group(self.probs, self.niters, reduce)
        return self

    def _run_sharded_small(self, reduce):
        """Row-sharded batch, K <= 64: batched kernels, 1 MAX + niters SUM all-reduces of nprob*K floats."""
        L = lib()
        wsb = self.nprob * L.gx_sinkhorn_workspace_bytes(self.K)
        ws = workspace(self.tag, wsb, self.S.device)
        S, n, K = self.S, self.nprob, self.K
        rs = 0 if self.r is None else self.r.stride(0)
        cs = 0 if self.c is None else self.c.stride(0)
        cm = torch.empty(n, K, dtype=torch.float32, device=S.device)
        u = [torch.empty(n, K, dtype=torch.float32, device=S.device) for _ in range(2)]
        _lib.check(L.gx_sinkhorn_small_batch_colmax(_p(S), n, S.stride(0), self.rows, K, S.stride(1), _p(ws), wsb, _p(cm),
                                                    _stream()), "gx_sinkhorn_small_batch_colmax")
        reduce(cm, "max")
        for it in range(self.niters):
            _lib.check(L.gx_sinkhorn_small_batch_pass(
                _p(S), n, S.stride(0), self.rows, K, S.stride(1), self.eps, it, _p(self.r), rs, _p(self.c), cs,
                self.c_uniform, _p(cm), _p(u[(it + 1) & 1]) if it > 0 else None, _p(ws), wsb, _p(u[it & 1]), _stream()),
                "gx_sinkhorn_small_batch_pass")
            reduce(u[it & 1], "sum")
        _lib.check(L.gx_sinkhorn_small_batch_finish(n, K, self.eps, self.niters, _p(self.r), rs,
                                                    _p(u[(self.niters - 1) & 1]), _p(ws), _p(self.la), _stream()),
                   "gx_sinkhorn_small_batch_finish")
        return self


def sinkhorn_table_ok(npatch, k):
    """the table solver takes this many patches and prototypes (their scaling vectors must fit one CU's LDS)"""
    return bool(lib().gx_sinkhorn_table_ok(int(npatch), int(k)))


class SinkhornTable:
    """All Sinkhorn problems of a many-prototype step solved ON the two per-pixel score tables (csrc/gx_sktab.hip): problem
    (p, v[, b]) = the rows of view v's table that patch p drew [in image b]; every pass reads a table row ONCE for all the
    patches that drew it.

    s_tab / t_tab (npix, K); inv (P, npix) int32: row of table row q in patch p's problem(s), -1 = not drawn; live: int32 HIP
    scalar (the drawn pixels are table rows [0, live)) or None.  ngroups = 1: one problem per patch and view; ngroups = B with
    seg (B + 1,) int32 on the device: one per image too — image b's live rows are table rows [seg[b], seg[b + 1]) and inv
    numbers a patch's samples image by image (rows / ngroups each).  Problem index = (p * 2 + view) * ngroups + image: `la`
    (2P ngroups, K) is what swapped_ce_pixels_tab reads; r (2P ngroups, K) / c (2P, rows) = the 'image' source pdf's marginals
    (c in the order inv numbers the rows) or None = uniform with c_uniform the row marginal.  rowmaps (optional, list of P int32
    vectors): lets self[i] hand out problem i's scores / codes (tests)."""

    def __init__(self, s_tab, t_tab, inv, live, eps, niters, c_uniform, rows, r=None, c=None, rowmaps=None, tag="sktab",
                 ngroups=1, seg=None):
        _chk(s_tab, t_tab, r, c)
        npix, k = s_tab.shape
        if tuple(t_tab.shape) != (npix, k) or t_tab.stride(0) != s_tab.stride(0) or inv.dtype != torch.int32 or \
                inv.dim() != 2 or inv.shape[1] != npix or not inv.is_contiguous() or not inv.is_cuda:
            raise RuntimeError("SinkhornTable: shape mismatch")
        if live is not None and (live.dtype != torch.int32 or not live.is_cuda or live.numel() != 1):
            raise RuntimeError("SinkhornTable: `live` is an int32 HIP scalar")
        if (r is None) != (c is None):
            raise RuntimeError("SinkhornTable: r and c come together")
        self.ngroups = int(ngroups)
        if self.ngroups > 1 and (seg is None or seg.dtype != torch.int32 or not seg.is_cuda or seg.numel() != self.ngroups + 1
                                 or not seg.is_contiguous() or rows % self.ngroups):
            raise RuntimeError("SinkhornTable: per-image problems need seg = (ngroups + 1) int32 row offsets on the device")
        self.P = inv.shape[0]
        self.nprob = 2 * self.P * self.ngroups
        if not sinkhorn_table_ok(self.P, k):
            raise RuntimeError(f"SinkhornTable: {self.P} patches x {k} prototypes do not fit (ops.sinkhorn_table_ok)")
        if r is not None and (tuple(r.shape) != (self.nprob, k) or tuple(c.shape) != (2 * self.P, rows)):
            raise RuntimeError("SinkhornTable: r is (2P ngroups, K), c is (2P, rows)")
        self.S = (s_tab, t_tab)
        self.inv, self.live, self.seg, self.K, self.npix, self.rows = inv, live, seg, k, npix, int(rows)
        self.eps, self.niters, self.c_uniform = float(eps), int(niters), float(c_uniform)
        self.r, self.c, self.rowmaps, self.tag = r, c, rowmaps, tag
        dev = s_tab.device
        self.la = torch.empty(self.nprob, k, dtype=torch.float32, device=dev)
        self.cmax = torch.empty(self.nprob, k, dtype=torch.float32, device=dev)
        self.u = torch.empty(self.nprob, k, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(self.nprob, 2, dtype=torch.float32, device=dev)      # what an update tells the next sweep
        self.wsb = lib().gx_sinkhorn_table_workspace_bytes(self.P, npix, k, self.ngroups)
        self.ws = workspace(tag, self.wsb, dev)

    def _args(self):
        s, t = self.S
        return (_p(s), _p(t), s.stride(0), self.K, _p(self.inv), self.npix, self.P, _p(self.live), _p(self.seg), self.ngroups,
                self.eps)

    def first(self):
        _lib.check(lib().gx_sinkhorn_table_first(*self._args(), _p(self.cmax), _p(self.u), _p(self.ws), self.wsb, _stream()),
                   "gx_sinkhorn_table_first")

    def sweep(self, shifted=False):
        """shifted: the sweep before this one ran on this workspace and `update` has been called since — the rows are
        exponentiated relative to a bound known beforehand where the update vouched for it (no maxima pass)"""
        _lib.check(lib().gx_sinkhorn_table_sweep(*self._args(), _p(self.la), _p(self.stats) if shifted else None, _p(self.c),
                                                 self.rows, self.c_uniform, _p(self.u), _p(self.ws), self.wsb, _stream()),
                   "gx_sinkhorn_table_sweep")

    def update(self, first, cmax=None):
        _lib.check(lib().gx_sinkhorn_table_update(_p(self.u), _p(self.r), self.nprob, self.K, int(first),
                                                  _p(self.cmax if cmax is None else cmax), self.eps, _p(self.la),
                                                  None if first else _p(self.stats), _stream()),
                   "gx_sinkhorn_table_update")

    def run(self, reduce=None):
        """reduce(tensor, 'max' | 'sum') all-reduces in place over the ranks (rows sharded, c_uniform = 1 / global rows):
        one MAX exchange of the problems' K column maxima and `niters` SUM exchanges of their K column sums"""
        if reduce is None:
            _lib.check(lib().gx_sinkhorn_table(*self._args(), self.niters, _p(self.r), _p(self.c), self.rows, self.c_uniform,
                                               _p(self.la), _p(self.cmax), _p(self.u), _p(self.stats), _p(self.ws), self.wsb,
                                               _stream()), "gx_sinkhorn_table")
            return self
        from .sharded import run_sinkhorn_table
        return run_sinkhorn_table(self, self.niters, reduce)

    def rebase(self, cm):
        """self.u (first-pass sums relative to this rank's maxima self.cmax) re-based on the maxima cm over all ranks"""
        _lib.check(lib().gx_sinkhorn_table_rebase(_p(self.u), _p(self.cmax), _p(cm), self.u.numel(), self.eps, _stream()),
                   "gx_sinkhorn_table_rebase")

    def __getitem__(self, i):
        """problem i = (p * 2 + view) * ngroups + image as a Sinkhorn state (scores(), codes()); needs `rowmaps`"""
        if self.rowmaps is None:
            raise RuntimeError("SinkhornTable: built without rowmaps")
        pv, g = divmod(i, self.ngroups)
        rm = self.rowmaps[pv // 2]
        if self.ngroups > 1:
            rm = rm.view(self.ngroups, -1)[g].contiguous()
        return Sinkhorn(self.S[pv % 2], self.eps, self.niters, c_uniform=self.c_uniform, tag=f"{self.tag}.{i}", la=self.la[i],
                        rowmap=rm)


def hist_marginal(x, bins, reduce=None):
    """normalised torch.histc(x, bins)+1e-9 with h[0]=h[1]; with `reduce` the data range and the
    bin counts are all-reduced over the ranks (x = this rank's part of the norm maps)."""
    _chk(x)
    L = lib()
    hist = torch.empty(bins, dtype=torch.float32, device=x.device)
    scratch = torch.empty(2 + 256, dtype=torch.int32, device=x.device)      # (min, max) as ordered ints + 256 block partials
    if reduce is None:
        _lib.check(L.gx_hist_marginal(_p(x), x.numel(), bins, _p(hist), _p(scratch), scratch.numel(), _stream()), "gx_hist_marginal")
        return hist
    lohi = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.check(L.gx_hist_range(_p(x), x.numel(), _p(scratch), _p(lohi), _stream()), "gx_hist_range")
    lohi[0].neg_()                      # one MAX all-reduce for both ends: (-min, max)
    reduce(lohi, "max")
    lohi[0].neg_()
    _lib.check(L.gx_hist_counts(_p(x), x.numel(), _p(lohi), bins, _p(hist), _p(scratch), _stream()), "gx_hist_counts")
    reduce(hist, "sum")
    _lib.check(L.gx_hist_finish(_p(hist), bins, _stream()), "gx_hist_finish")
    return hist


def swapped_ce(sk_s, sk_t, temperature, grad_scale, loss_out, loss_scale, loss_beta, d_s=None, d_t=None):
    """Fused Sinkhorn-codes + swapped-prediction CE forward (+ d scores).  sk_* are run()
    Sinkhorn states of the two views; loss_out is a 1-element HIP tensor (accumulated)."""
    S, T = sk_s.S, sk_t.S
    _chk(loss_out)
    _chk2d(d_s, d_t)                       # row-strided dS (rows padded to whole 16-column chunks): the kernel takes the stride
    if d_s is not None and d_t is not None and d_s.stride(0) != d_t.stride(0):
        raise RuntimeError("swapped_ce: d_s and d_t must share their row stride")
    L = lib()
    wsb = L.gx_swapped_ce_workspace_bytes(sk_s.rows, sk_s.K)
    ws = workspace("ce", wsb, S.device)
    _lib.check(L.gx_swapped_ce(_p(S), _p(T), sk_s.rows, sk_s.K, S.stride(0), sk_s.eps, temperature, _p(sk_s.la),
                               _p(sk_t.la), grad_scale, _p(d_s), _p(d_t),
                               (d_s.stride(0) if d_s is not None else 0), loss_scale, loss_beta, _p(loss_out), _p(ws),
                               wsb, _stream()), "gx_swapped_ce")
    return loss_out


def swapped_ce_fused_ok(sk_s, d_s):
    """many prototypes, aligned rows: the one-pass CE that also returns the bias gradient (gx_swapped_ce_fused)"""
    k, S = sk_s.K, sk_s.S
    return (64 < k <= 6144 and k % 4 == 0 and d_s is not None and S.stride(0) % 4 == 0 and d_s.stride(0) % 4 == 0 and
            S.data_ptr() % 16 == 0 and d_s.data_ptr() % 16 == 0)


def swapped_ce_fused(sk_s, sk_t, temperature, grad_scale, loss_out, loss_scale, loss_beta, d_s, d_t, gb, gb_beta):
    """swapped_ce + gb = gb_beta * gb + colsum(d_s) + colsum(d_t), one pass over the two score matrices"""
    S, T = sk_s.S, sk_t.S
    _chk(loss_out, gb)
    _chk2d(d_s, d_t)                       # row-strided dS (padded rows) is fine: the kernel takes the stride
    L = lib()
    wsb = L.gx_swapped_ce_fused_workspace_bytes(sk_s.K)
    ws = workspace("ce_fused", wsb, S.device)
    with record("ce_fused_bigk_kernel", 0.0, 16.0 * sk_s.rows * sk_s.K):        # two score matrices read, two dS written
            _lib.check(L.gx_swapped_ce_fused(_p(S), _p(T), sk_s.rows, sk_s.K, S.stride(0), sk_s.eps, temperature, _p(sk_s.la),
                                         _p(sk_t.la), grad_scale, _p(d_s), _p(d_t), d_s.stride(0), loss_scale, loss_beta,
                                         _p(loss_out), _p(gb), gb_beta, _p(ws), wsb, _stream()), "gx_swapped_ce_fused")
    return loss_out


def swapped_ce_pixels(scores_all, skb, ngroups, inv, temperature, grad_scale, loss_out, loss_scale, loss_beta, d_s, d_t, kpad,
                      gb, gb_beta):
    """The swapped CE of every patch of a step, pixel-major (gx_swapped_ce_pixels): scores_all (P, 2, rows, K) with skb its
    solved SinkhornBatch ((p * 2 + view) * ngroups + group order), inv (P, npix) int32 = the pixel's row in patch p or -1;
    d_s / d_t (npix, >= kpad) are written with the pixel's d(scores) summed over the patches that drew it."""
    _chk(scores_all, loss_out, gb)
    _chk2d(d_s, d_t)
    npatch, two, rows, k = scores_all.shape
    npix = d_s.shape[0]
    if two != 2 or inv.dtype != torch.int32 or tuple(inv.shape) != (npatch, npix) or not inv.is_contiguous() or not inv.is_cuda or \
            d_t.shape[0] != npix or d_s.stride(0) != d_t.stride(0) or skb.la.shape[0] != npatch * 2 * ngroups:
        raise RuntimeError("swapped_ce_pixels: shape mismatch")
    L = lib()
    wsb = L.gx_swapped_ce_fused_workspace_bytes(k)
    ws = workspace("ce_fused", wsb, scores_all.device)
    with record("ce_fused_bigk_kernel", 0.0, 8.0 * npatch * rows * k + 8.0 * npix * kpad):     # scores once, tables once
        _lib.check(L.gx_swapped_ce_pixels(_p(scores_all), npatch, rows, k, scores_all.stride(2), _p(inv), npix, ngroups, skb.eps,
                                          temperature, _p(skb.la), grad_scale, _p(d_s), _p(d_t), d_s.stride(0), kpad, loss_scale,
                                          loss_beta, _p(loss_out), _p(gb), gb_beta, _p(ws), wsb, _stream()),
                   "gx_swapped_ce_pixels")
    return loss_out


def swapped_ce_pixels_tab(s_tab, t_tab, skb, ngroups, inv, temperature, grad_scale, loss_out, loss_scale, loss_beta, d_s, d_t, kpad,
                          gb, gb_beta, live=None, live_frac=1.0, row_group=None):
    """swapped_ce_pixels with the scores as per-pixel tables too (s_tab / t_tab (npix, K), one per view): a pixel's score rows
    are read once for all the patches that drew it; inv (P, npix) int32 only says which did (>= 0).  live (int32 HIP scalar):
    the drawn pixels are the first `live` rows of the tables — the d(scores) rows past them are left as they are (live_frac: its expected share, for the byte accounting).  row_group (int32
    (npix,), with ngroups > 1): the Sinkhorn problem (image) of every table row — the rows of an image are not npix / ngroups apart
    once the drawn pixels come first"""
    if live is not None and (live.dtype != torch.int32 or not live.is_cuda or live.numel() != 1):
        raise RuntimeError("swapped_ce_pixels_tab: `live` is an int32 HIP scalar")
    if row_group is not None and (row_group.dtype != torch.int32 or not row_group.is_cuda or row_group.numel() != s_tab.shape[0]
                                  or not row_group.is_contiguous()):
        raise RuntimeError("swapped_ce_pixels_tab: `row_group` is an int32 HIP vector with one entry per table row")
    _chk(s_tab, t_tab, loss_out, gb)
    _chk2d(d_s, d_t)
    npix, k = s_tab.shape
    npatch = inv.shape[0]
    if inv.dtype != torch.int32 or tuple(inv.shape) != (npatch, npix) or not inv.is_contiguous() or not inv.is_cuda or \
            tuple(t_tab.shape) != (npix, k) or d_s.shape[0] != npix or d_t.shape[0] != npix or d_s.stride(0) != d_t.stride(0) or \
            skb.la.shape[0] != npatch * 2 * ngroups:
        raise RuntimeError("swapped_ce_pixels_tab: shape mismatch")
    L = lib()
    wsb = L.gx_swapped_ce_fused_workspace_bytes(k)
    ws = workspace("ce_fused", wsb, s_tab.device)
    with record("ce_fused_bigk_kernel", 0.0, (8.0 * npix * k + 8.0 * npix * kpad) * live_frac):    # score tables once, d(scores) tables once
        _lib.check(L.gx_swapped_ce_pixels_tab(_p(s_tab), _p(t_tab), npatch, k, s_tab.stride(0), _p(inv), npix, ngroups, skb.eps,
                                              temperature, _p(skb.la), grad_scale, _p(d_s), _p(d_t), d_s.stride(0), kpad,
                                              loss_scale, loss_beta, _p(loss_out), _p(gb), gb_beta, _p(live), _p(row_group), _p(ws),
                                              wsb, _stream()),
                   "gx_swapped_ce_pixels_tab")
    return loss_out


# ------------------------------------------------------------------------------------- non-linear projection heads
def gather_rows(z, picks, maps=None):
    """z (B, H, W, D) table; picks (B, N) int32 -> the sampled rows (B*N, D), zero where the rotation zero-filled"""
    _chk(z)
    b, h, w, d = z.shape
    n = picks.shape[1]
    out = torch.empty(b * n, d, dtype=torch.float32, device=z.device)
    _lib.check(lib().gx_gather_rows(_p(z), b, h * w, d, _p(picks), n, _p(maps), _p(out), _stream()), "gx_gather_rows")
    return out


def scatter_rows_add_(grows, picks, maps, g):
    """g (B, H, W, D) += the adjoint of gather_rows on grows (B*N, D) (float atomics)"""
    _chk(grows, g)
    b, h, w, d = g.shape
    _lib.check(lib().gx_scatter_rows_add(_p(grows), b, h * w, d, _p(picks), picks.shape[1], _p(maps), _p(g), _stream()),
               "gx_scatter_rows_add")
    return g


def leaky_relu_(x, slope=0.01):
    _chk(x)
    _lib.check(lib().gx_leaky_relu(_p(x), x.numel(), slope, _stream()), "gx_leaky_relu")
    return x


def leaky_relu_bwd_(dy, y, slope=0.01):
    """dy *= LeakyReLU'(.) read off the OUTPUT y (slope > 0 keeps the sign of the input)"""
    _chk(dy, y)
    _lib.check(lib().gx_leaky_relu_bwd(_p(dy), _p(y), dy.numel(), slope, _stream()), "gx_leaky_relu_bwd")
    return dy


def col_moment(x, center=None, square=False, scale=1.0):
    """scale * sum over rows of (x - center) or its square, per column of x (rows, D)"""
    _chk(x, center)
    rows, d = x.shape
    L = lib()
    out = torch.empty(d, dtype=torch.float32, device=x.device)
    wsb = L.gx_col_moment_workspace_bytes(d)
    ws = workspace("colmoment", wsb, x.device)
    _lib.check(L.gx_col_moment(_p(x), rows, d, _p(center), int(square), scale, _p(out), _p(ws), wsb, _stream()),
               "gx_col_moment")
    return out


def batch_stats(x, reduce=None, world=1):
    """BatchNorm1d's batch mean and biased variance over the rows of x — over the rows of every rank when `reduce`
    (t, op) is given (equal row counts per rank)"""
    n = x.shape[0] * world
    mean = col_moment(x, None, False, 1.0 / n)
    if reduce is not None:
        reduce(mean, "sum")
    var = col_moment(x, mean, True, 1.0 / n)
    if reduce is not None:
        reduce(var, "sum")
    return mean, var


BN_LRELU, BN_TANH = 0, 1


def bn_act_fwd(x, mean, var, gamma, beta, eps, act, slope=0.01, want_xhat=True, out=None, xhat_out=None):
    _chk(x, mean, var, gamma, beta, out, xhat_out)
    a = torch.empty_like(x) if out is None else out
    xhat = xhat_out if xhat_out is not None else (torch.empty_like(x) if want_xhat else None)
    _lib.check(lib().gx_bn_act_fwd(_p(x), x.shape[0], x.shape[1], _p(mean), _p(var), _p(gamma), _p(beta), eps, act, slope,
                                   _p(a), _p(xhat), _stream()), "gx_bn_act_fwd")
    return a, xhat


def bn_act_bwd(da, xhat, var, gamma, beta, eps, act, slope=0.01, reduce=None, world=1, out=None):
    """-> (dx, dgamma, dbeta) with dgamma / dbeta THIS rank's sums (they join the step's gradient all-reduce); the
    copies that enter dx are all-reduced first when `reduce` is given"""
    _chk(da, xhat, var, gamma, beta, out)
    rows, d = da.shape
    L = lib()
    dg = torch.empty(d, dtype=torch.float32, device=da.device)
    db = torch.empty(d, dtype=torch.float32, device=da.device)
    wsb = L.gx_col_moment_workspace_bytes(d)
    ws = workspace("colmoment", wsb, da.device)
    _lib.check(L.gx_bn_act_bwd_sums(_p(da), _p(xhat), rows, d, _p(gamma), _p(beta), act, slope, _p(dg), _p(db), _p(ws), wsb,
                                    _stream()), "gx_bn_act_bwd_sums")
    dg_all, db_all = dg, db
    if reduce is not None:
        dg_all, db_all = dg.clone(), db.clone()
        reduce(dg_all, "sum")
        reduce(db_all, "sum")
    dx = torch.empty_like(da) if out is None else out
    _lib.check(L.gx_bn_act_bwd_apply(_p(da), _p(xhat), rows, d, _p(var), _p(gamma), _p(beta), eps, act, slope, _p(dg_all),
                                     _p(db_all), float(rows * world), _p(dx), _stream()), "gx_bn_act_bwd_apply")
    return dx, dg, db


# ------------------------------------------------------------------------------------- SimCLR head
def col_stats(x):
    """(mean, biased var) over the rows of x (rows, D)"""
    _chk(x)
    rows, d = x.shape
    L = lib()
    mean = torch.empty(d, dtype=torch.float32, device=x.device)
    var = torch.empty(d, dtype=torch.float32, device=x.device)
    wsb = L.gx_col_stats_workspace_bytes(d)
    ws = workspace("colstats", wsb, x.device)
    _lib.check(L.gx_col_stats(_p(x), rows, d, _p(mean), _p(var), _p(ws), wsb, _stream()), "gx_col_stats")
    return mean, var


def bn_lrelu_fwd(x, mean, var, gamma, beta, eps=1e-5, slope=0.01, want_xhat=True):
    _chk(x, mean, var, gamma, beta)
    a = torch.empty_like(x)
    xhat = torch.empty_like(x) if want_xhat else None
    _lib.check(lib().gx_bn_lrelu_fwd(_p(x), x.shape[0], x.shape[1], _p(mean), _p(var), _p(gamma), _p(beta), eps, slope,
                                     _p(a), _p(xhat), _stream()), "gx_bn_lrelu_fwd")
    return a, xhat


def bn_lrelu_bwd(da, xhat, var, gamma, beta, eps=1e-5, slope=0.01):
    _chk(da, xhat, var, gamma, beta)
    n, d = da.shape
    dx = torch.empty_like(da)
    dg = torch.empty(d, dtype=torch.float32, device=da.device)
    db = torch.empty(d, dtype=torch.float32, device=da.device)
    _lib.check(lib().gx_bn_lrelu_bwd(_p(da), _p(xhat), n, d, _p(var), _p(gamma), _p(beta), eps, slope, _p(dx), _p(dg),
                                     _p(db), _stream()), "gx_bn_lrelu_bwd")
    return dx, dg, db


def simclr_ntxent(out, temperature, want_grad=True):
    """loss (1,) and d loss / d out of the reference's inline SimCLR loss on out (2b, D)"""
    _chk(out)
    n, d = out.shape
    loss = torch.empty(1, dtype=torch.float32, device=out.device)
    dout = torch.empty_like(out) if want_grad else None
    _lib.check(lib().gx_simclr_ntxent(_p(out), n, d, float(temperature), _p(loss), _p(dout), _stream()),
               "gx_simclr_ntxent")
    return loss, dout


def row_div_(z, nrm):
    _chk(z, nrm)
    _lib.check(lib().gx_row_div(_p(z), z.shape[0], z.shape[1], _p(nrm), _stream()), "gx_row_div")
    return z


# ------------------------------------------------------------------------------------- optimizer
def sumsq(x, out):
    """out[0] = sum(x^2) (deterministic)."""
    _chk(x, out)
    L = lib()
    wsb = L.gx_sumsq_workspace_bytes()
    ws = workspace("sumsq", wsb, x.device)
    _lib.check(L.gx_sumsq(_p(x), x.numel(), _p(out), _p(ws), wsb, _stream()), "gx_sumsq")
    return out


def larc_sgd_(p, g, buf, lr, momentum, trust, first_step, use_larc=True):
    _chk(p, g, buf)
    norms = None
    if use_larc:
        norms = torch.empty(2, dtype=torch.float32, device=p.device)
        sumsq(p, norms[0:1])
        sumsq(g, norms[1:2])
    _lib.check(lib().gx_larc_sgd(_p(p), _p(g), _p(buf), p.numel(), _p(norms), lr, momentum, trust, int(first_step),
                                 int(use_larc), _stream()), "gx_larc_sgd")
    return p


def sqrt_fan(cin, k):
    return 1.0 / math.sqrt(cin * k * k)
