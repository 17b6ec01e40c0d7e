"""Thin torch-tensor wrappers over the C ABI (include/idiff.h).  torch supplies device memory and the
stream; every arithmetic operation below runs in the hand-written HIP kernels.  No fallbacks."""
import contextlib
import ctypes as C
import math
import os
import threading

import torch

from . import _lib
from ._lib import ConvDesc, check

CONV_NORMAL, CONV_UPSAMPLE2, CONV_UNSHUFFLE2 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
SDE_STEP, SDE_MEAN, SDE_ODE = 0, 1, 2

# bench.py sets this to a list to time every conv launch with events on the launch stream (roofline leg)
PROFILE = None
# tests set this to a collections.Counter: (algo, ks, Cin, Cout, Hout, Wout) -> calls, to assert which kernel served a layer
ALGO_TRACE = None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, name="tensor", dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.IdiffError(f"{name} must live on the GPU (the hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.IdiffError(f"{name} must be {dtype}, got {t.dtype}")


def _bs(t, name="tensor"):
    """batch stride of an NCHW tensor whose samples are contiguous (channel slices of a bigger buffer allowed)."""
    _chk(t, name)
    exp = 1
    for d in range(t.dim() - 1, 0, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise _lib.IdiffError(f"{name}: samples must be contiguous (shape {tuple(t.shape)}, strides {t.stride()})")
        exp *= t.shape[d]
    return t.stride(0) if t.shape[0] > 1 else max(exp, t.stride(0))


def _c(t, name="tensor", dtype=torch.float32):
    _chk(t, name, dtype)
    if t is not None and not t.is_contiguous():
        raise _lib.IdiffError(f"{name} must be contiguous")
    return t


# IDIFF_WINOGRAD=0 keeps every 3x3 conv on the direct implicit-GEMM kernel (A/B runs, parity bisection)
WINOGRAD = bool(int(os.environ.get("IDIFF_WINOGRAD", "1")))
# IDIFF_X3=0: no three-plane bf16 images of 1x1 weights are built, so every 1x1 conv stays on the f32 matrix cores
X3 = bool(int(os.environ.get("IDIFF_X3", "1")))
# IDIFF_WINOGRAD4=0 keeps the forward 3x3 convs off the F(4x4,3x3) kernel (they run F(2x2,3x3) or direct instead)
WINOGRAD4 = WINOGRAD and bool(int(os.environ.get("IDIFF_WINOGRAD4", "1")))
# IDIFF_WINOGRAD4_DGRAD=0 keeps the data-gradient convs of the backward pass on F(2x2,3x3)
WINOGRAD4_DGRAD = bool(int(os.environ.get("IDIFF_WINOGRAD4_DGRAD", "1")))


CONV_ALGO_DIRECT, CONV_ALGO_WINOGRAD, CONV_ALGO_STREAM1X1, CONV_ALGO_WINOGRAD4, CONV_ALGO_WINOGRAD4H, CONV_ALGO_X3 = 0, 1, 2, 3, 4, 5
_ALGO_REQUEST = threading.local()


@contextlib.contextmanager
def request_conv3x3_algo(algo):
    """Diagnostic / test scope (this thread only): every 3x3 conv2d() issued inside asks the library for `algo` through the per-call
    idiff_conv_desc.algo_request where the layer's shape tiles for it, e.g. the F(4x4,3x3) kernel on the small levels of a 64x64
    input (which the library's own choice leaves on F(2x2,3x3)).  The product path never enters this scope."""
    old = getattr(_ALGO_REQUEST, "algo", None)
    _ALGO_REQUEST.algo = algo
    try:
        yield
    finally:
        _ALGO_REQUEST.algo = old


# ---------------------------------------------------------------------------------------------------
def _alloc_conv_images(w, transpose):
    """The direct image and, as attributes, whichever other images the weight's shape admits -- allocated, not filled."""
    lib = _lib.load()
    _c(w, "weight")
    co, ci, k, _ = w.shape
    out = torch.empty((k * k, co, ci) if transpose else (k * k, ci, co), device=w.device, dtype=torch.float32)
    if k == 3 and WINOGRAD and co % 8 == 0 and ci % 8 == 0 and (ci if transpose else co) % 16 == 0:
        # Winograd-domain copy rides along as an attribute; conv2d hands it to the C ABI (idiff_conv_desc.wwino)
        cconv, kconv = (ci, co) if transpose else (co, ci)  # the conv's (Cout, Cin); Cout is padded to whole 64-blocks
        out.wino = torch.empty((16 * kconv * ((cconv + 63) // 64) * 64,), device=w.device, dtype=torch.float32)
        if WINOGRAD4 and (WINOGRAD4_DGRAD or not transpose):
            out.wino4 = torch.empty((36 * kconv * ((cconv + 63) // 64) * 64,), device=w.device, dtype=torch.float32)
    cconv1, kconv1 = (ci, co) if transpose else (co, ci)  # the conv's (Cout, Cin)
    if k == 1 and X3 and cconv1 % 64 == 0 and kconv1 >= 32 and kconv1 % 8 == 0:
        # three-plane bf16 image of a 1x1 weight (idiff_conv_desc.wx3): flattened 1x1 layers then run on the bf16 matrix cores; the
        # data-gradient pack is the image of the transposed matrix
        out.x3 = torch.empty((lib.idiff_conv1x1_x3_image_bytes(cconv1, kconv1) // 2,), device=w.device, dtype=torch.int16)
    return out


def _fill_conv_images(out, w, transpose, algo=None):
    """algo None: every image `out` carries; CONV_ALGO_x: the one image that kernel reads."""
    lib = _lib.load()
    co, ci, k, _ = w.shape
    tr = 1 if transpose else 0
    if algo is None or algo == CONV_ALGO_DIRECT:
        fn = lib.idiff_pack_conv_weight_T if transpose else lib.idiff_pack_conv_weight
        check(fn(_p(w), _p(out), co, ci, k, _stream()), "pack_conv_weight")
    if hasattr(out, "wino") and (algo is None or algo == CONV_ALGO_WINOGRAD):
        check(lib.idiff_pack_conv_weight_wino(_p(w), _p(out.wino), co, ci, tr, _stream()), "pack_conv_weight_wino")
    if hasattr(out, "wino4") and (algo is None or algo in (CONV_ALGO_WINOGRAD4, CONV_ALGO_WINOGRAD4H)):
        check(lib.idiff_pack_conv_weight_wino4(_p(w), _p(out.wino4), co, ci, tr, _stream()), "pack_conv_weight_wino4")
    if hasattr(out, "x3") and (algo is None or algo == CONV_ALGO_X3):
        cconv1, kconv1 = (ci, co) if transpose else (co, ci)
        wm = w.reshape(co, ci).t().contiguous() if transpose else w
        check(lib.idiff_pack_conv1x1_x3(_p(wm), out.x3.data_ptr(), cconv1, kconv1, _stream()), "pack_conv1x1_x3")


def pack_conv_weight(w, transpose=False):
    """[Cout,Cin,k,k] -> packed [k*k][Cin][Cout] (or the flipped/transposed pack for the data gradient), with the Winograd-domain /
    split-bf16 images the shape admits as attributes (.wino, .wino4, .x3)."""
    out = _alloc_conv_images(w, transpose)
    _fill_conv_images(out, w, transpose)
    return out


class LazyConvWeight:
    """A conv weight for ONE conv2d call whose images are packed at that call: conv2d asks the library which kernel it will launch
    (idiff_conv2d_plan) and fills only the image that kernel reads.  For weights that change every step (training: one pack launch
    per use instead of three); the sampling path packs once per weight version and keeps every image (pack_conv_weight)."""

    def __init__(self, w, transpose=False):
        self.w, self.transpose = _c(w, "weight"), transpose


def conv2d(src0, wpk, bias, ks, Cout, src1=None, mode=CONV_NORMAL, pro=None, out=None, res=None, vec=None, aux=None,
           want_stats=False, algo=None, gn=None):
    """Implicit-GEMM conv.  pro=(a,b): per-(b,c) affine+SiLU applied to src0 while it is gathered;
    aux=(tensor,a,b): adds silu(a*tensor+b) in the epilogue.  Returns out or (out, stats).
    algo: None = the library picks; CONV_ALGO_x = that kernel or an error (idiff_conv_desc.algo_request).
    gn = dict(groups, gamma, beta, film=None, eps=1e-5, want_mean_rstd=False): the GroupNorm(+FiLM) finalize of this conv's
    statistics rides on the call (a finalize launch enqueued by the library behind the conv): returns (out, (a, b)) or
    (out, (a, b, mean_rstd))."""
    lib = _lib.load()
    B, C0, Hin, Win = src0.shape
    lazy = wpk if isinstance(wpk, LazyConvWeight) else None
    if lazy is not None:
        wpk = _alloc_conv_images(lazy.w, lazy.transpose)
    d = ConvDesc()
    if algo is not None:
        d.algo_request = 1 + algo
    elif ks == 3 and getattr(_ALGO_REQUEST, "algo", None) is not None:
        d.algo_request = -(1 + _ALGO_REQUEST.algo)  # scope request: taken where the shape tiles, the library's choice elsewhere
    d.src0, d.src0_bstride, d.C0 = src0.data_ptr(), _bs(src0, "src0"), C0
    if src1 is not None:
        assert src1.shape[0] == B and src1.shape[2:] == src0.shape[2:]
        d.src1, d.src1_bstride, d.C1 = src1.data_ptr(), _bs(src1, "src1"), src1.shape[1]
    d.B, d.Hin, d.Win, d.mode, d.ks, d.Cout = B, Hin, Win, mode, ks, Cout
    _c(wpk, "wpk")
    d.wpk = wpk.data_ptr()
    wino = getattr(wpk, "wino", None)
    if wino is not None and ks == 3:
        d.wwino = wino.data_ptr()
        wino4 = getattr(wpk, "wino4", None)
        if wino4 is not None:
            d.wwino4 = wino4.data_ptr()
    x3 = getattr(wpk, "x3", None)
    if x3 is not None and ks == 1 and mode in (CONV_NORMAL, CONV_UNSHUFFLE2):
        d.wx3 = x3.data_ptr()
    if bias is not None:
        d.bias = _c(bias, "bias").data_ptr()
    if pro is not None:
        d.pro_a, d.pro_b = _c(pro[0], "pro_a").data_ptr(), _c(pro[1], "pro_b").data_ptr()
    if mode == CONV_UPSAMPLE2:
        Hout, Wout = Hin * 2, Win * 2
    elif mode == CONV_UNSHUFFLE2:
        Hout, Wout = Hin // 2, Win // 2
    else:
        Hout, Wout = Hin, Win
    if out is None:
        out = torch.empty((B, Cout, Hout, Wout), device=src0.device, dtype=torch.float32)
    else:
        assert tuple(out.shape) == (B, Cout, Hout, Wout), (out.shape, (B, Cout, Hout, Wout))
    d.out, d.out_bstride = out.data_ptr(), _bs(out, "out")
    if res is not None:
        assert res.shape == out.shape
        d.res, d.res_bstride = res.data_ptr(), _bs(res, "res")
    if vec is not None:
        assert tuple(vec.shape) == (B, Cout)
        d.vec = _c(vec, "vec").data_ptr()
    if aux is not None:
        t, a, b = aux
        assert t.shape == out.shape
        d.aux, d.aux_bstride = t.data_ptr(), _bs(t, "aux")
        d.aux_a, d.aux_b = _c(a, "aux_a").data_ptr(), _c(b, "aux_b").data_ptr()
    stats = None
    bufs = gn.get("bufs") if gn is not None else None  # caller-placed (stats, out_a, out_b[, mean_rstd]): the guard-band tests
    if want_stats or gn is not None:
        nt = lib.idiff_conv2d_num_tiles(Hout, Wout)
        stats = bufs[0] if bufs is not None else torch.empty((B, nt, Cout, 2), device=src0.device, dtype=torch.float32)
        assert tuple(stats.shape) == (B, nt, Cout, 2) and stats.is_contiguous() and stats.dtype == torch.float32
        d.stats = stats.data_ptr()
    gn_out = None
    if gn is not None:
        if bufs is not None:
            ga, gb = bufs[1], bufs[2]
            assert tuple(ga.shape) == (B, Cout) == tuple(gb.shape) and ga.is_contiguous() and gb.is_contiguous()
        else:
            ga, gb = torch.empty((B, Cout), device=src0.device, dtype=torch.float32), torch.empty((B, Cout), device=src0.device, dtype=torch.float32)
        d.gn_groups, d.gn_eps = int(gn["groups"]), float(gn.get("eps", 1e-5))
        d.gn_gamma, d.gn_beta = _c(gn["gamma"], "gn gamma").data_ptr(), _c(gn["beta"], "gn beta").data_ptr()
        film = gn.get("film")
        if film is not None:
            _chk(film, "film")
            assert film.shape[0] == B and film.shape[1] == 2 * Cout and film.stride(1) == 1
            d.gn_film, d.gn_film_ld = film.data_ptr(), film.stride(0)
        d.gn_out_a, d.gn_out_b = ga.data_ptr(), gb.data_ptr()
        gn_out = (ga, gb)
        if gn.get("want_mean_rstd"):
            mr = bufs[3] if bufs is not None else torch.empty((B, d.gn_groups, 2), device=src0.device, dtype=torch.float32)
            assert tuple(mr.shape) == (B, d.gn_groups, 2) and mr.is_contiguous()
            d.gn_mean_rstd = mr.data_ptr()
            gn_out = (ga, gb, mr)
    planned = None
    if lazy is not None:
        planned = lib.idiff_conv2d_plan(C.byref(d))
        check(min(planned, 0), "conv2d_plan")
        _fill_conv_images(wpk, lazy.w, lazy.transpose, planned)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.idiff_conv2d_fwd(C.byref(d), _stream()), "conv2d_fwd")
        e1.record()
        Cin = (C0 * 4 if mode == CONV_UNSHUFFLE2 else C0) + (src1.shape[1] if src1 is not None else 0)
        PROFILE.append(dict(ks=ks, mode=mode, Cin=Cin, Cout=Cout, B=B, Hout=Hout, Wout=Wout, e0=e0, e1=e1, algo=lib.idiff_conv2d_last_algo(),
                            flops=2.0 * Cin * Cout * ks * ks * Hout * Wout * B))
    else:
        check(lib.idiff_conv2d_fwd(C.byref(d), _stream()), "conv2d_fwd")
    if planned is not None and lib.idiff_conv2d_last_algo() != planned:  # the one image that was filled is not the one that was read
        raise RuntimeError("conv2d: planned kernel %d, launched %d" % (planned, lib.idiff_conv2d_last_algo()))
    if ALGO_TRACE is not None:
        Cin = (C0 * 4 if mode == CONV_UNSHUFFLE2 else C0) + (src1.shape[1] if src1 is not None else 0)
        ALGO_TRACE[(lib.idiff_conv2d_last_algo(), ks, Cin, Cout, Hout, Wout)] += 1
    if gn is not None:
        return (out, gn_out, stats) if want_stats else (out, gn_out)
    return (out, stats) if want_stats else out


def gn_finalize(stats, groups, HW, gamma, beta, film=None, eps=1e-5, want_mean_rstd=False):
    lib = _lib.load()
    B, nt, Cc, _ = stats.shape
    a = torch.empty((B, Cc), device=stats.device, dtype=torch.float32)
    b = torch.empty_like(a)
    mr = torch.empty((B, groups, 2), device=stats.device, dtype=torch.float32) if want_mean_rstd else None
    film_ld = 0
    if film is not None:
        _chk(film, "film")
        assert film.shape[0] == B and film.shape[1] == 2 * Cc and film.stride(1) == 1
        film_ld = film.stride(0)
    check(lib.idiff_gn_finalize(_p(_c(stats)), nt, B, Cc, groups, HW, _p(_c(gamma)), _p(_c(beta)), _p(film), film_ld, eps,
                                _p(a), _p(b), _p(mr), _stream()), "gn_finalize")
    return (a, b, mr) if want_mean_rstd else (a, b)


def affine_silu_add(h, ab=None, res=None, vec=None, out=None):
    lib = _lib.load()
    B, Cc = h.shape[:2]
    HW = h.shape[2] * h.shape[3]
    if out is None:
        out = torch.empty(h.shape, device=h.device, dtype=torch.float32)
    a, b = ab if ab is not None else (None, None)
    check(lib.idiff_affine_silu_add(_p(h), _bs(h, "h"), _p(_c(a)), _p(_c(b)), _p(res), _bs(res, "res") if res is not None else 0,
                                    _p(_c(vec)), _p(out), _bs(out, "out"), B, Cc, HW, _stream()), "affine_silu_add")
    return out


def linear(x, w, bias=None, res=None, gscale=None, act_in=ACT_NONE, act_out=ACT_NONE, out=None):
    """x [R,K] (row-strided ok), w [N,K] (row-strided ok) -> [R,N]."""
    lib = _lib.load()
    _chk(x, "x"), _chk(w, "w")
    assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] == w.shape[1]
    R, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((R, N), device=x.device, dtype=torch.float32)
    assert out.stride(1) == 1 and tuple(out.shape) == (R, N)
    if res is not None:
        assert res.stride(1) == 1 and tuple(res.shape) == (R, N)
    check(lib.idiff_linear_fwd(_p(x), x.stride(0), _p(w), w.stride(0), _p(_c(bias)), _p(res), res.stride(0) if res is not None else 0,
                               _p(_c(gscale)), _p(out), out.stride(0), R, K, N, act_in, act_out, _stream()), "linear_fwd")
    return out


def linear_t(x, wT, bias=None, res=None, gscale=None, act_in=ACT_NONE, act_out=ACT_NONE, out=None, ln=None):
    """x [R,K] (row-strided ok), wT [K,N] PRE-TRANSPOSED weight (row-strided ok) -> [R,N].
    ln = (gamma, beta, eps): LayerNorm(x) over K fused in front of the product (no act_in then)."""
    lib = _lib.load()
    _chk(x, "x"), _chk(wT, "wT")
    assert x.dim() == 2 and wT.dim() == 2 and x.stride(1) == 1 and wT.stride(1) == 1 and x.shape[1] == wT.shape[0]
    R, K = x.shape
    N = wT.shape[1]
    if out is None:
        out = torch.empty((R, N), device=x.device, dtype=torch.float32)
    assert out.stride(1) == 1 and tuple(out.shape) == (R, N)
    if res is not None:
        assert res.stride(1) == 1 and tuple(res.shape) == (R, N)
    if ln is not None:
        assert act_in == ACT_NONE
        check(lib.idiff_linear_t_ln_fwd(_p(x), x.stride(0), _p(_c(ln[0])), _p(_c(ln[1])), float(ln[2]), _p(wT), wT.stride(0), _p(_c(bias)), _p(res),
                                        res.stride(0) if res is not None else 0, _p(_c(gscale)), _p(out), out.stride(0), R, K, N, act_out,
                                        _stream()), "linear_t_ln_fwd")
        return out
    check(lib.idiff_linear_t_fwd(_p(x), x.stride(0), _p(wT), wT.stride(0), _p(_c(bias)), _p(res), res.stride(0) if res is not None else 0,
                                 _p(_c(gscale)), _p(out), out.stride(0), R, K, N, act_in, act_out, _stream()), "linear_t_fwd")
    return out


def linear_t_grouped(groups):
    """ONE launch for up to 16 independent linear_t problems (idiff_linear_t_grouped_fwd).  Each group is a dict with the keyword
    arguments of linear_t: x, wT, bias=None, res=None, gscale=None, act_in, act_out, out=None, ln=None.  Returns the outputs."""
    lib = _lib.load()
    n = len(groups)
    assert 1 <= n <= _lib.LINEAR_MAX_GROUPS, n
    arr = (_lib.LinearGroup * n)()
    outs = []
    for d, g in zip(arr, groups):
        x, wT = g["x"], g["wT"]
        _chk(x, "x"), _chk(wT, "wT")
        assert x.dim() == 2 and wT.dim() == 2 and x.stride(1) == 1 and wT.stride(1) == 1 and x.shape[1] == wT.shape[0]
        R, K = x.shape
        N = wT.shape[1]
        out = g.get("out")
        if out is None:
            out = torch.empty((R, N), device=x.device, dtype=torch.float32)
        assert out.stride(1) == 1 and tuple(out.shape) == (R, N)
        res, bias, gs, ln = g.get("res"), g.get("bias"), g.get("gscale"), g.get("ln")
        d.x, d.ldx, d.wT, d.ldw = x.data_ptr(), x.stride(0), wT.data_ptr(), wT.stride(0)
        d.bias = _c(bias, "bias").data_ptr() if bias is not None else None
        if res is not None:
            assert res.stride(1) == 1 and tuple(res.shape) == (R, N)
            d.res, d.ldr = res.data_ptr(), res.stride(0)
        d.gscale = _c(gs, "gscale").data_ptr() if gs is not None else None
        d.out, d.ldo = out.data_ptr(), out.stride(0)
        if ln is not None:
            assert g.get("act_in", ACT_NONE) == ACT_NONE
            d.ln_g, d.ln_b, d.ln_eps = _c(ln[0], "ln_g").data_ptr(), _c(ln[1], "ln_b").data_ptr(), float(ln[2])
        d.R, d.K, d.N, d.act_in, d.act_out = R, K, N, g.get("act_in", ACT_NONE), g.get("act_out", ACT_NONE)
        outs.append(out)
    check(lib.idiff_linear_t_grouped_fwd(arr, n, _stream()), "linear_t_grouped_fwd")
    return outs


def attn_tokens_packed_grouped(qkvs, heads, scale):
    """attn_tokens_packed for several packed projections of one shape [B,N,3C] in ONE launch -> list of [B,N,C]"""
    lib = _lib.load()
    n = len(qkvs)
    assert 1 <= n <= _lib.LINEAR_MAX_GROUPS
    B, Nq, C3 = qkvs[0].shape
    Cc = C3 // 3
    PA = C.c_void_p * n
    q, k, v, o = PA(), PA(), PA(), PA()
    outs = []
    for i, t in enumerate(qkvs):
        _c(t, "qkv")
        assert tuple(t.shape) == (B, Nq, C3)
        out = torch.empty((B, Nq, Cc), device=t.device, dtype=torch.float32)
        base = t.data_ptr()
        q[i], k[i], v[i], o[i] = base, base + 4 * Cc, base + 8 * Cc, out.data_ptr()
        outs.append(out)
    check(lib.idiff_attn_tokens_grouped_fwd(q, k, v, o, n, B, Nq, Nq, Cc, heads, scale, C3, C3, _stream()), "attn_tokens_grouped_fwd")
    return outs


def linear_t_heads(x, wT, bias, out, heads, K, N, x_hs, w_hs, b_hs, o_hs):
    """heads independent [R,K] x [K,N] products in one launch: head h reads x[:, h*x_hs : h*x_hs+K], the [K,N] block of wT
    starting w_hs elements further per head, bias[h*b_hs : h*b_hs+N], and writes out[:, h*o_hs : h*o_hs+N]."""
    lib = _lib.load()
    _chk(x, "x"), _chk(wT, "wT"), _chk(out, "out")
    assert x.dim() == 2 and wT.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and wT.stride(1) == 1 and out.stride(1) == 1
    R = x.shape[0]
    assert out.shape[0] == R and (heads - 1) * x_hs + K <= x.shape[1] and (heads - 1) * o_hs + N <= out.shape[1]
    assert (heads - 1) * w_hs + (K - 1) * wT.stride(0) + N <= wT.numel()
    check(lib.idiff_linear_t_heads_fwd(_p(x), x.stride(0), x_hs, _p(wT), wT.stride(0), w_hs, _p(_c(bias)), b_hs, _p(out), out.stride(0), o_hs,
                                       R, K, N, heads, _stream()), "linear_t_heads_fwd")
    return out


def smm_memproj(feat, ln1_g, ln1_b, wpk, bias, ln2_g, ln2_b, eps=1e-5):
    """feat [B,C,H,W] -> mem [B,256,H*W] = LN(Linear(LN(tokens)))"""
    lib = _lib.load()
    B, Cc, H, W = feat.shape
    out = torch.empty((B, 256, H * W), device=feat.device, dtype=torch.float32)
    check(lib.idiff_smm_memproj_fwd(_p(feat), _bs(feat, "feat"), _p(_c(ln1_g)), _p(_c(ln1_b)), _p(_c(wpk)), _p(_c(bias)), _p(_c(ln2_g)),
                                    _p(_c(ln2_b)), _p(out), B, Cc, H * W, eps, _stream()), "smm_memproj_fwd")
    return out


def smm_memproj_compact(feat, ln1_g, ln1_b, gram, hvec, evar, Cm, eps1=1e-5, eps2=1e-5):
    """feat [B,C,H,W] -> [B,Cm,H*W] rows [xhat*rstd2 ; rstd2 ; 0]: the (C+1)-dim affine pre-image of the 256-wide memory.
    gram [C,C], hvec [C], evar: the quadratic form of the 256-wide variance (memory_variance_form)."""
    lib = _lib.load()
    B, Cc, H, W = feat.shape
    out = torch.empty((B, Cm, H * W), device=feat.device, dtype=torch.float32)
    check(lib.idiff_smm_memproj_compact_fwd(_p(feat), _bs(feat, "feat"), _p(_c(ln1_g)), _p(_c(ln1_b)), _p(_c(gram)), _p(_c(hvec)), float(evar),
                                            _p(out), B, Cc, H * W, Cm, eps1, eps2, _stream()), "smm_memproj_compact_fwd")
    return out


def memory_variance_form(lin_weight, lin_bias):
    """(gram [C,C], hvec [C], evar) with var_256(W x + b) = x^T gram x + 2 hvec.x + evar  (host-side weight preparation, fp64)."""
    W = lin_weight.detach().double()
    b = lin_bias.detach().double()
    Wc = W - W.mean(dim=0, keepdim=True)
    bc = b - b.mean()
    n = W.shape[0]
    return (Wc.t() @ Wc / n).float().contiguous(), (Wc.t() @ bc / n).float().contiguous(), float(bc.dot(bc) / n)


def layernorm_rows(x, gamma, beta, eps=1e-5, want_mean_rstd=False):
    lib = _lib.load()
    _chk(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    out = torch.empty((R, Cc), device=x.device, dtype=torch.float32)
    mr = torch.empty((R, 2), device=x.device, dtype=torch.float32) if want_mean_rstd else None
    check(lib.idiff_layernorm_rows_fwd(_p(x), x.stride(0), _p(_c(gamma)), _p(_c(beta)), _p(out), Cc, R, Cc, eps, _p(mr), _stream()),
          "layernorm_rows")
    return (out, mr) if want_mean_rstd else out


def time_embed(t, dim, freqs=None):
    lib = _lib.load()
    _c(t, "t"), _c(freqs, "freqs")
    B = t.numel()
    out = torch.empty((B, dim), device=t.device, dtype=torch.float32)
    check(lib.idiff_time_embed_fwd(_p(t), _p(freqs), B, dim, _p(out), _stream()), "time_embed")
    return out


def chan_layernorm(x, gamma, beta, eps=1e-5, want_mean_rstd=False):
    lib = _lib.load()
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    mr = torch.empty((B, H * W, 2), device=x.device, dtype=torch.float32) if want_mean_rstd else None
    check(lib.idiff_chan_layernorm_fwd(_p(x), _bs(x, "x"), _p(_c(gamma)), _p(_c(beta)), _p(out), Cc * H * W, B, Cc, H * W, eps, _p(mr),
                                       _stream()), "chan_layernorm")
    return (out, mr) if want_mean_rstd else out


# IDIFF_ATTN_DTYPE=bf16 | f16: the self-attention contractions on the bf16 / fp16 matrix cores (reduced-precision VARIANTS, never the
# default; f16 = the reference's own form, operands clamped to +-255; bench.py --attn bf16|f16 reports them as their own lines with
# the PSNR delta against the fp32 path)
ATTN_DTYPE = os.environ.get("IDIFF_ATTN_DTYPE", "f32").lower()


def attn_self(qkv, heads, scale, want_lse=False):
    """qkv [B,3C,H,W] -> [B,C,H,W]"""
    lib = _lib.load()
    _c(qkv, "qkv")
    B, C3, H, W = qkv.shape
    Cc, N = C3 // 3, H * W
    out = torch.empty((B, Cc, H, W), device=qkv.device, dtype=torch.float32)
    if ATTN_DTYPE in ("bf16", "f16") and not want_lse and Cc // heads == 64 and N % 4 == 0:
        fn = lib.idiff_attn_self_bf16_fwd if ATTN_DTYPE == "bf16" else lib.idiff_attn_self_f16_fwd
        check(fn(_p(qkv), _p(out), B, Cc, N, heads, scale, _stream()), "attn_self_%s_fwd" % ATTN_DTYPE)
        return out
    lse = torch.empty((B, heads, N), device=qkv.device, dtype=torch.float32) if want_lse else None
    check(lib.idiff_attn_self_fwd(_p(qkv), _p(out), _p(lse), B, Cc, N, heads, scale, _stream()), "attn_self_fwd")
    return (out, lse) if want_lse else out


def attn_ctx(q, k, v, heads, scale):
    """q [B,C,H,W]; k,v [B,M,C] -> [B,C,H,W]"""
    lib = _lib.load()
    _c(q, "q"), _c(k, "k"), _c(v, "v")
    B, Cc, H, W = q.shape
    M = k.shape[1]
    out = torch.empty_like(q)
    check(lib.idiff_attn_ctx_fwd(_p(q), _p(k), _p(v), _p(out), B, Cc, H * W, M, heads, scale, _stream()), "attn_ctx_fwd")
    return out


def attn_tokens(q, k, v, heads, scale):
    """q [B,Nq,C]; k,v [B,M,C] -> [B,Nq,C]"""
    lib = _lib.load()
    _c(q, "q"), _c(k, "k"), _c(v, "v")
    B, Nq, Cc = q.shape
    M = k.shape[1]
    out = torch.empty_like(q)
    check(lib.idiff_attn_tokens_fwd(_p(q), _p(k), _p(v), _p(out), B, Nq, M, Cc, heads, scale, Cc, Cc, _stream()), "attn_tokens_fwd")
    return out


def attn_tokens_packed(qkv, heads, scale):
    """self-attention on a packed projection qkv [B,N,3C] (q | k | v along the last dim) -> [B,N,C]; no copies."""
    lib = _lib.load()
    _c(qkv, "qkv")
    B, Nq, C3 = qkv.shape
    Cc = C3 // 3
    out = torch.empty((B, Nq, Cc), device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    check(lib.idiff_attn_tokens_fwd(C.c_void_p(base), C.c_void_p(base + 4 * Cc), C.c_void_p(base + 8 * Cc), _p(out), B, Nq, Nq, Cc, heads, scale,
                                    C3, C3, _stream()), "attn_tokens_fwd")
    return out


def attn_tokens_packed_f16(qkv, heads, scale):
    """attn_tokens_packed in the reference's half-precision form (Attention_flash: +-255 clamp, fp16 operands and result, fp32
    statistics; idiff_attn_tokens_f16_fwd) -- the `if_flash` variant of the ScoreMapModule decoder"""
    lib = _lib.load()
    _c(qkv, "qkv")
    B, Nq, C3 = qkv.shape
    Cc = C3 // 3
    out = torch.empty((B, Nq, Cc), device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    check(lib.idiff_attn_tokens_f16_fwd(C.c_void_p(base), C.c_void_p(base + 4 * Cc), C.c_void_p(base + 8 * Cc), _p(out), B, Nq, Nq, Cc, heads, scale,
                                        C3, C3, _stream()), "attn_tokens_f16_fwd")
    return out


def smm_xattn_kv_f16(q, k, v, heads, scale):
    """cross-attention of q [B,Nq,C] over UNFOLDED keys / values k, v [B,C,N] (channel-major) in the reference's half-precision form
    (Attention_flash) -> [B,Nq,C]; C = heads * 64, heads = 4, Nq <= 8"""
    lib = _lib.load()
    _c(q, "q"), _c(k, "k"), _c(v, "v")
    B, Nq, Cc = q.shape
    N = k.shape[2]
    assert tuple(k.shape) == (B, Cc, N) and tuple(v.shape) == (B, Cc, N)
    ws = torch.empty((lib.idiff_smm_xattn_kv_f16_ws_floats(B, N),), device=q.device, dtype=torch.float32)
    out = torch.empty_like(q)
    check(lib.idiff_smm_xattn_kv_f16_fwd(_p(q), _p(k), _p(v), _p(out), _p(ws), B, Nq, heads, Cc, N, scale, _stream()), "smm_xattn_kv_f16_fwd")
    return out


def smm_xattn(qf, mem, scale):
    """qf [B,Nq,heads,Cm]; mem [B,Cm,N] -> o [B,Nq,heads,Cm] (attention-weighted mem rows)."""
    lib = _lib.load()
    _c(qf, "qf"), _c(mem, "mem")
    B, Nq, heads, Cm = qf.shape
    N = mem.shape[2]
    nws = lib.idiff_smm_xattn_ws_floats(B, Nq, heads, Cm, N)
    ws = torch.empty((nws,), device=qf.device, dtype=torch.float32)
    o = torch.empty_like(qf)
    check(lib.idiff_smm_xattn_fwd(_p(qf), _p(mem), _p(o), _p(ws), B, Nq, heads, Cm, N, scale, _stream()), "smm_xattn_fwd")
    return o


def smm_xattn_grouped(qfs, mems, scale):
    """smm_xattn for several (qf [B,Nq,heads,Cm_i], mem [B,Cm_i,N_i]) pairs of one (B, Nq, heads) in ONE attention launch + ONE merge
    launch (idiff_smm_xattn_grouped_fwd) -> list of o; the same bits as the single calls."""
    lib = _lib.load()
    n = len(qfs)
    assert 1 <= n <= _lib.XATTN_MAX_GROUPS and len(mems) == n
    B, Nq, heads, _ = qfs[0].shape
    arr = (_lib.XattnGroup * n)()
    outs, keep = [], []
    for d, qf, mem in zip(arr, qfs, mems):
        _c(qf, "qf"), _c(mem, "mem")
        Cm, N = qf.shape[3], mem.shape[2]
        assert tuple(qf.shape[:3]) == (B, Nq, heads) and tuple(mem.shape[:2]) == (B, Cm)
        ws = torch.empty((lib.idiff_smm_xattn_ws_floats(B, Nq, heads, Cm, N),), device=qf.device, dtype=torch.float32)
        o = torch.empty_like(qf)
        d.qf, d.mem, d.o, d.ws, d.Cm, d.N = qf.data_ptr(), mem.data_ptr(), o.data_ptr(), ws.data_ptr(), Cm, N
        outs.append(o)
        keep.append(ws)
    check(lib.idiff_smm_xattn_grouped_fwd(arr, n, B, Nq, heads, scale, _stream()), "smm_xattn_grouped_fwd")
    return outs


def smm_memproj_compact_grouped(items, eps1=1e-5, eps2=1e-5):
    """smm_memproj_compact for several levels in ONE launch; items: dicts feat, ln1_g, ln1_b, gram, hvec, evar, Cm -> list of [B,Cm,H*W]"""
    lib = _lib.load()
    n = len(items)
    assert 1 <= n <= _lib.MEMPROJ_MAX_GROUPS
    B = items[0]["feat"].shape[0]
    arr = (_lib.MemprojGroup * n)()
    outs = []
    for d, it in zip(arr, items):
        feat = it["feat"]
        Bf, Cc, H, W = feat.shape
        assert Bf == B
        out = torch.empty((B, it["Cm"], H * W), device=feat.device, dtype=torch.float32)
        d.feat, d.feat_bstride = feat.data_ptr(), _bs(feat, "feat")
        d.ln1_g, d.ln1_b = _c(it["ln1_g"]).data_ptr(), _c(it["ln1_b"]).data_ptr()
        d.gram, d.hvec, d.evar = _c(it["gram"]).data_ptr(), _c(it["hvec"]).data_ptr(), float(it["evar"])
        d.out, d.C, d.N, d.Cm = out.data_ptr(), Cc, H * W, it["Cm"]
        outs.append(out)
    check(lib.idiff_smm_memproj_compact_grouped_fwd(arr, n, B, eps1, eps2, _stream()), "smm_memproj_compact_grouped_fwd")
    return outs


def scoremap_grouped(feats, tvs, idx):
    """scoremap for several levels in ONE launch -> list of (score [B,K,H,W], sel [B,1,H,W] or None)"""
    lib = _lib.load()
    n = len(feats)
    assert 1 <= n <= _lib.SCOREMAP_MAX_GROUPS and len(tvs) == n
    B = feats[0].shape[0]
    K = tvs[0].shape[1]
    if idx is not None:
        _c(idx, "idx", torch.int32)
    arr = (_lib.ScoremapGroup * n)()
    outs = []
    for d, feat, tv in zip(arr, feats, tvs):
        Bf, Cc, H, W = feat.shape
        _c(tv, "tv")
        assert Bf == B and tuple(tv.shape) == (B, K, Cc)
        out = torch.empty((B, K, H, W), device=feat.device, dtype=torch.float32)
        sel = torch.empty((B, 1, H, W), device=feat.device, dtype=torch.float32) if idx is not None else None
        d.feat, d.feat_bstride, d.tv, d.out, d.sel, d.C, d.HW = feat.data_ptr(), _bs(feat, "feat"), tv.data_ptr(), out.data_ptr(), \
            (sel.data_ptr() if sel is not None else None), Cc, H * W
        outs.append((out, sel))
    check(lib.idiff_scoremap_grouped_fwd(arr, n, _p(idx), B, K, _stream()), "scoremap_grouped_fwd")
    return outs


def time_mlp(t, freqs, w0, b0, w2, b2):
    """temb [B, nout] = w2 . GELU(w0 . sinusoidal(t) + b0) + b2 in one launch (idiff_time_mlp_fwd)"""
    lib = _lib.load()
    _c(t, "t"), _c(freqs, "freqs"), _c(w0, "w0"), _c(b0, "b0"), _c(w2, "w2"), _c(b2, "b2")
    B = t.numel()
    hid, dim = w0.shape
    nout = w2.shape[0]
    assert w2.shape[1] == hid
    out = torch.empty((B, nout), device=t.device, dtype=torch.float32)
    check(lib.idiff_time_mlp_fwd(_p(t), _p(freqs), _p(w0), _p(b0), _p(w2), _p(b2), _p(out), B, dim, hid, nout, _stream()), "time_mlp")
    return out


def conv3x3_select(x, weight, bias, idx):
    """final 3x3 conv + class gather in one pass: x [B,C,H,W], weight [K,C,3,3] (torch layout), idx int32 [B] -> [B,1,H,W]."""
    lib = _lib.load()
    B, Cc, H, W = x.shape
    K = weight.shape[0]
    assert tuple(weight.shape) == (K, Cc, 3, 3) and idx.dtype == torch.int32 and idx.numel() == B
    out = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
    check(lib.idiff_conv3x3_select_fwd(_p(x), _bs(x, "x"), _p(_c(weight)), _p(_c(bias)), C.c_void_p(idx.contiguous().data_ptr()), _p(out), B, Cc, K, H, W,
                                       _stream()), "conv3x3_select_fwd")
    return out


def scoremap(feat, tv, idx=None):
    """feat [B,C,H,W], tv [B,K,C] -> score [B,K,H,W] (+ sel [B,1,H,W] = score[b, idx[b]])."""
    lib = _lib.load()
    B, Cc, H, W = feat.shape
    K = tv.shape[1]
    _c(tv, "tv")
    out = torch.empty((B, K, H, W), device=feat.device, dtype=torch.float32)
    sel = None
    if idx is not None:
        _c(idx, "idx", torch.int32)
        sel = torch.empty((B, 1, H, W), device=feat.device, dtype=torch.float32)
    check(lib.idiff_scoremap_fwd(_p(feat), _bs(feat, "feat"), _p(tv), _p(out), _p(idx), _p(sel), B, Cc, H * W, K, _stream()), "scoremap")
    return out, sel


def image_metrics(pred, target):
    """pred/target [B,H,W] in [-1,1] -> [B,3] = (RMSE, PSNR dB, SSIM) on x/2+0.5, data_range 1 (device tensor)."""
    lib = _lib.load()
    pred, target = pred.contiguous(), target.contiguous()
    _c(pred, "pred"), _c(target, "target")
    B, H, W = pred.shape
    out = torch.empty((B, 3), device=pred.device, dtype=torch.float32)
    ws = torch.empty((B * 128,), device=pred.device, dtype=torch.float32)
    check(lib.idiff_image_metrics(_p(pred), _p(target), _p(out), _p(ws), B, H, W, _stream()), "image_metrics")
    return out


def gather_channel(x, idx):
    lib = _lib.load()
    _c(x, "x"), _c(idx, "idx", torch.int32)
    B, Cc, H, W = x.shape
    out = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
    check(lib.idiff_gather_channel(_p(x), _p(idx), _p(out), B, Cc, H * W, _stream()), "gather_channel")
    return out


# ---------------------------------------------------------------------------------------------------
def irsde_reverse_step(x, mu, noise_pred, z, theta, sigma, sigma_bar, dt, sqrt_dt, mode=SDE_STEP, seed=0, offset=0, out=None):
    lib = _lib.load()
    _c(x, "x"), _c(mu, "mu"), _c(noise_pred, "noise_pred"), _c(z, "z")
    if out is None:
        out = torch.empty_like(x)
    check(lib.idiff_irsde_reverse_step(_p(x), _p(mu), _p(noise_pred), _p(z), _p(out), x.numel(), theta, sigma, sigma_bar, dt, sqrt_dt, mode,
                                       seed, offset, _stream()), "irsde_reverse_step")
    return out


# op codes of idiff_irsde_map (include/idiff.h)
(IRSDE_SCORE_FROM_NOISE, IRSDE_MU_BAR, IRSDE_DRIFT, IRSDE_REV_DRIFT, IRSDE_DISPERSION, IRSDE_STEP_MEAN, IRSDE_STEP_SDE, IRSDE_FORWARD_STEP,
 IRSDE_OPT_STEP, IRSDE_REAL_NOISE, IRSDE_REAL_SCORE, IRSDE_INIT_FROM_NOISE, IRSDE_RANDOM_STATES) = range(13)


def irsde_map(op, like, a=None, b=None, z=None, mu=None, k=None, coef_dev=None, seed=0, offset=0, out=None):
    """One piece of the reference's IRSDE arithmetic (include/idiff.h: IDIFF_IRSDE_*) over tensors shaped like `like` [B,...].
    mu: tensor of that shape or a python number; k: up to 6 python floats, or coef_dev: device [B,6] per-sample rows."""
    lib = _lib.load()
    _c(like, "like")
    for name, t in (("a", a), ("b", b), ("z", z)):
        _c(t, name)
        if t is not None and t.shape != like.shape:
            raise _lib.IdiffError(f"irsde_map: operand {name} has shape {tuple(t.shape)}, expected {tuple(like.shape)}")
    mu_t, mu_s = (mu, 0.0) if torch.is_tensor(mu) else (None, float(mu if mu is not None else 0.0))
    if mu_t is not None:
        if mu_t.shape != like.shape:
            mu_t = mu_t.expand(like.shape).contiguous()
        _c(mu_t, "mu")
    B = like.shape[0] if (coef_dev is not None and like.dim() > 0) else 1
    karr = None
    if coef_dev is not None:
        _c(coef_dev, "coef_dev")
        assert tuple(coef_dev.shape) == (B, 6), (coef_dev.shape, B)
    else:
        kk = [float(v) for v in (k or [])]
        karr = (C.c_float * 6)(*(kk + [0.0] * (6 - len(kk))))
    if out is None:
        out = torch.empty_like(like)
    check(lib.idiff_irsde_map(op, _p(a), _p(b), _p(z), _p(mu_t), mu_s, _p(out), B, like.numel() // B, _p(coef_dev), karr, seed, offset,
                              _stream()), "irsde_map")
    return out


def drift_reverse_step(x, r_hat, e_hat, z, a, b, c, cond=None, seed=0, offset=0, out=None, xa_out=None):
    lib = _lib.load()
    _c(x, "x"), _c(r_hat, "r_hat"), _c(e_hat, "e_hat"), _c(z, "z"), _c(cond, "cond")
    if out is None:
        out = torch.empty_like(x)
    if cond is not None and xa_out is None:
        xa_out = torch.empty_like(x)
    check(lib.idiff_drift_reverse_step(_p(x), _p(r_hat), _p(e_hat), _p(z), _p(cond), _p(out), _p(xa_out), x.numel(), a, b, c, seed, offset,
                                       _stream()), "drift_reverse_step")
    return (out, xa_out) if cond is not None else out


def drift_reverse_step_dev(x, r_hat, e_hat, z_base, cond, xa, coef, state, seed, nper, offset_base=0):
    """in-place, graph-replayable drift step: per-step scalars from `coef` [3, T+1] and `state` int32 [3] on the device"""
    lib = _lib.load()
    _c(x, "x"), _c(r_hat, "r_hat"), _c(e_hat, "e_hat"), _c(z_base, "z"), _c(cond, "cond"), _c(xa, "xa"), _c(coef, "coef")
    assert state.dtype == torch.int32 and state.is_cuda and state.numel() == 3 and coef.dim() == 2 and coef.shape[0] == 3
    check(lib.idiff_drift_reverse_step_dev(_p(x), _p(r_hat), _p(e_hat), _p(z_base), _p(cond), _p(xa), x.numel(), _p(coef), coef.shape[1],
                                           C.c_void_p(state.data_ptr()), seed, nper, offset_base, _stream()), "drift_reverse_step_dev")


def step_state_advance(state, tdev, T, t_stop=0):
    lib = _lib.load()
    _c(tdev, "tdev")
    check(lib.idiff_step_state_advance(C.c_void_p(state.data_ptr()), _p(tdev), tdev.numel(), T, t_stop, _stream()), "step_state_advance")


def randn(shape, device, seed, offset=0):
    lib = _lib.load()
    out = torch.empty(shape, device=device, dtype=torch.float32)
    check(lib.idiff_randn(_p(out), out.numel(), seed, offset, _stream()), "randn")
    return out


def dropout(x, p, seed, offset, out=None):
    """inverted dropout with the Philox-keyed mask of (seed, offset): x / (1-p) where kept, else 0 (the backward is the same call)"""
    lib = _lib.load()
    _c(x, "x")
    if out is None:
        out = torch.empty_like(x)
    check(lib.idiff_dropout(_p(x), _p(out), x.numel(), float(p), int(seed), int(offset), _stream()), "dropout")
    return out


def philox_raw(ncounters, device, seed, offset=0):
    lib = _lib.load()
    out = torch.empty((ncounters, 4), device=device, dtype=torch.int32)
    check(lib.idiff_philox_raw(_p(out), ncounters, seed, offset, _stream()), "philox_raw")
    return out


def axpby(x, y, alpha, beta, out=None):
    lib = _lib.load()
    _c(x, "x"), _c(y, "y")
    if out is None:
        out = torch.empty_like(x)
    check(lib.idiff_axpby(_p(x), _p(y), _p(out), x.numel(), alpha, beta, _stream()), "axpby")
    return out


def mix3_per_sample(x0, cond, eps, c0, c1, c2):
    lib = _lib.load()
    for t in (x0, cond, eps, c0, c1, c2):
        _c(t)
    B = x0.shape[0]
    out = torch.empty_like(x0)
    check(lib.idiff_mix3_per_sample(_p(x0), _p(cond), _p(eps), _p(c0), _p(c1), _p(c2), _p(out), B, x0.numel() // B, _stream()),
          "mix3_per_sample")
    return out


def f32_to_bf16(x, out=None):
    """flat fp32 -> bf16 (round to nearest even), the wire format of the gradient all-reduce"""
    lib = _lib.load()
    _c(x, "x")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _c(out, "out", torch.bfloat16)
    check(lib.idiff_f32_to_bf16(_p(x), C.c_void_p(out.data_ptr()), x.numel(), _stream()), "f32_to_bf16")
    return out


def bf16_to_f32(x, out=None):
    lib = _lib.load()
    _c(x, "x", torch.bfloat16)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _c(out, "out")
    check(lib.idiff_bf16_to_f32(C.c_void_p(x.data_ptr()), _p(out), x.numel(), _stream()), "bf16_to_f32")
    return out

