"""Training path: autograd Functions whose forward AND backward are the hand-written HIP kernels of
include/idiff.h, the fused Adam optimizer and the `optimize_parameters_inputRes` train step
(models/drift_noise_model.py:242-312).  torch.autograd only routes gradients between these Functions; no
gradient arithmetic runs in ATen except autograd's own accumulation of fan-in gradients."""
import ctypes as C

import os

import torch

from . import _lib, ops
from ._lib import ConvDesc, check
from .ops import _bs, _c, _chk, _p, _stream

# bumped whenever weights are modified behind torch's back (fused Adam writes parameters in place through raw
# pointers, which does not bump tensor._version); prepared-weight caches include it in their signature
WEIGHT_EPOCH = [0]


class _prof:
    """bench.py's training roofline pass (ops.PROFILE set to a list): HIP events on the launch stream around one library call of a
    matrix-core kernel class, with the FLOP that call executes.  A no-op otherwise."""

    def __init__(self, kind, flops, **extra):
        self.on = ops.PROFILE is not None
        if self.on:
            self.rec = dict(kind=kind, flops=float(flops), e0=torch.cuda.Event(enable_timing=True), e1=torch.cuda.Event(enable_timing=True), **extra)

    def __enter__(self):
        if self.on:
            self.rec["e0"].record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.rec["e1"].record()
            ops.PROFILE.append(self.rec)
        return False


# =====================================================================================================
# raw wrappers of the training kernels
# =====================================================================================================
def _conv_desc(src0, src1, mode, ks, Cout, pro=None):
    B, C0, Hin, Win = src0.shape
    d = ConvDesc()
    d.src0, d.src0_bstride, d.C0 = src0.data_ptr(), _bs(src0, "src0"), C0
    if src1 is not None:
        d.src1, d.src1_bstride, d.C1 = src1.data_ptr(), _bs(src1, "src1"), src1.shape[1]
    d.B, d.Hin, d.Win, d.mode, d.ks, d.Cout = B, Hin, Win, mode, ks, Cout
    if pro is not None:
        d.pro_a, d.pro_b = _c(pro[0]).data_ptr(), _c(pro[1]).data_ptr()
    return d


def conv2d_wgrad(src0, src1, mode, ks, dy, Cin, pro=None, dw=None, accumulate=False):
    """dW [Cout, Cin, ks, ks] of conv2d(src0 (+src1), mode, ks, pro) given dy."""
    lib = _lib.load()
    Cout = dy.shape[1]
    d = _conv_desc(src0, src1, mode, ks, Cout, pro)
    nws = lib.idiff_conv2d_wgrad_ws_floats(C.byref(d))
    ws = torch.empty((nws,), device=dy.device, dtype=torch.float32)
    if dw is None:
        dw = torch.empty((Cout, Cin, ks, ks), device=dy.device, dtype=torch.float32)
        accumulate = False
    B, _, Hout, Wout = dy.shape
    pr = _prof("wgrad", 2.0 * Cin * Cout * ks * ks * Hout * Wout * B, ks=ks)
    with pr:
        check(lib.idiff_conv2d_wgrad(C.byref(d), _p(dy), _bs(dy, "dy"), _p(_c(dw)), 1 if accumulate else 0, _p(ws), _stream()), "conv2d_wgrad")
    if pr.on:
        pr.rec["algo"] = lib.idiff_conv2d_wgrad_last_algo()
    return dw


def conv2d_dgrad(dy, weight, ks, mode, Cin_virtual, res=None):
    """data gradient w.r.t. the (virtual) conv input: [B, Cin_v, Hout, Wout]; the caller undoes upsample/unshuffle."""
    w = weight.detach().contiguous()
    wT = ops.LazyConvWeight(w, transpose=True) if LAZY_PACK else ops.pack_conv_weight(w, transpose=True)  # packed at the call
    return ops.conv2d(dy, wT, None, ks, Cin_virtual, res=res)


def channel_sums(x, per_sample=False):
    """sum over pixels (and batch unless per_sample): [C] or [B,C]"""
    lib = _lib.load()
    B, Cc, H, W = x.shape
    bc = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    check(lib.idiff_plane_sum(_p(x), _bs(x, "x"), _p(bc), B, Cc, H * W, _stream()), "plane_sum")
    if per_sample:
        return bc
    out = torch.empty((Cc,), device=x.device, dtype=torch.float32)
    check(lib.idiff_batch_sum(_p(bc), _p(out), B, Cc, 0, _stream()), "batch_sum")
    return out


def sumpool2x2(x):
    lib = _lib.load()
    _c(x)
    B, Cc, H2, W2 = x.shape
    out = torch.empty((B, Cc, H2 // 2, W2 // 2), device=x.device, dtype=torch.float32)
    check(lib.idiff_sumpool2x2(_p(x), _p(out), B * Cc, H2 // 2, W2 // 2, _stream()), "sumpool2x2")
    return out


def pixel_shuffle2(x):
    lib = _lib.load()
    _c(x)
    B, C4, h, w = x.shape
    out = torch.empty((B, C4 // 4, 2 * h, 2 * w), device=x.device, dtype=torch.float32)
    check(lib.idiff_pixel_shuffle2(_p(x), _p(out), B, C4 // 4, h, w, _stream()), "pixel_shuffle2")
    return out


def gn_silu_bwd(dy, h, a, b, mean_rstd, gamma, beta, film, groups, want_sums=False):
    """-> dh, dgamma, dbeta, dfilm (or None) [, dh_sum [C], dy_sum [B,C]]: with want_sums the plane sums of dh (bias gradient of the conv
    that produced h) and of dy come out of the same passes (no plane_sum launch over either tensor)"""
    lib = _lib.load()
    B, Cc, H, W = h.shape
    dh = torch.empty((B, Cc, H, W), device=h.device, dtype=torch.float32)
    dgamma = torch.empty((Cc,), device=h.device, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    dfilm = torch.empty((B, 2 * Cc), device=h.device, dtype=torch.float32) if film is not None else None
    dh_sum = torch.empty((Cc,), device=h.device, dtype=torch.float32) if want_sums else None
    dy_sum = torch.empty((B, Cc), device=h.device, dtype=torch.float32) if want_sums else None
    ws = torch.empty((lib.idiff_gn_silu_bwd_ws_floats(B, Cc, groups),), device=h.device, dtype=torch.float32)
    check(lib.idiff_gn_silu_bwd(_p(dy), _bs(dy, "dy"), _p(h), _bs(h, "h"), _p(_c(a)), _p(_c(b)), _p(_c(mean_rstd)), _p(_c(gamma)), _p(_c(beta)),
                                _p(film), film.stride(0) if film is not None else 0, _p(dh), _bs(dh), _p(dgamma), _p(dbeta), _p(dfilm),
                                2 * Cc, _p(ws), B, Cc, groups, H * W, 0, _p(dh_sum), _p(dy_sum), _stream()), "gn_silu_bwd")
    if want_sums:
        return dh, dgamma, dbeta, dfilm, dh_sum, dy_sum
    return dh, dgamma, dbeta, dfilm


def batch_sum(bc):
    """[B,C] -> [C] (fixed order over b)"""
    lib = _lib.load()
    B, Cc = bc.shape
    out = torch.empty((Cc,), device=bc.device, dtype=torch.float32)
    check(lib.idiff_batch_sum(_p(_c(bc)), _p(out), B, Cc, 0, _stream()), "batch_sum")
    return out


def bgemm(A, B, M, N, K, lda, ldb, transA, transB, sA, sB, batch, out=None, alpha=1.0, beta=0.0, ldc=None, sC=None):
    """ldc / sC: row and batch stride of C inside `out` (default: a dense [batch, M, N]); A, B, out are base pointers (tensors whose
    data_ptr() is the first element of batch 0)"""
    lib = _lib.load()
    _chk(A), _chk(B)
    if out is None:
        out = torch.empty((batch, M, N), device=A.device, dtype=torch.float32)
    nws = lib.idiff_bgemm_ws_floats(M, N, K, batch)
    ws = torch.empty((nws,), device=A.device, dtype=torch.float32) if nws else None
    with _prof("bgemm", 2.0 * M * N * K * batch):
        check(lib.idiff_bgemm(_p(A), _p(B), _p(out), M, N, K, lda, ldb, N if ldc is None else ldc, 1 if transA else 0, 1 if transB else 0, sA, sB,
                              M * N if sC is None else sC, batch, alpha, beta, _p(ws), _stream()), "bgemm")
    return out


# =====================================================================================================
# autograd Functions
# =====================================================================================================
def _samples_contiguous(t):
    """t itself when every sample is contiguous (a channel slice of a bigger NCHW tensor: the gradient of one source of a virtual
    concat, of torch.cat) -- the kernels take a batch stride; a copy otherwise."""
    exp = 1
    for d in range(t.dim() - 1, 0, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            return t.contiguous()
        exp *= t.shape[d]
    return t if (t.shape[0] == 1 or t.stride(0) >= exp) and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 else t.contiguous()


LAZY_PACK = os.environ.get("IDIFF_LAZY_PACK", "1") != "0"  # A/B runs: 0 = every image of a weight packed at every use (r03)


def _packed(w):
    if not LAZY_PACK:
        return ops.pack_conv_weight(w.detach().contiguous())
    return ops.LazyConvWeight(w.detach().contiguous())  # packed at the conv call: only the image the chosen kernel reads


class ConvFn(torch.autograd.Function):
    """plain conv (virtual concat / upsample / unshuffle gather modes), bias; no fused prologue/epilogue."""

    @staticmethod
    def forward(ctx, src0, src1, weight, bias, ks, mode, slot=None):
        Cout = weight.shape[0]
        out = ops.conv2d(src0, _packed(weight), bias, ks, Cout, src1=src1, mode=mode, out=None if slot is None else slot.t)
        ctx.save_for_backward(src0, src1, weight)
        ctx.ks, ctx.mode, ctx.has_bias = ks, mode, bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        src0, src1, weight = ctx.saved_tensors
        dout = _samples_contiguous(dout)
        ks, mode = ctx.ks, ctx.mode
        Cout, Cin = weight.shape[0], weight.shape[1]
        C0 = src0.shape[1]
        d0 = d1 = None
        if ctx.needs_input_grad[0] or (src1 is not None and ctx.needs_input_grad[1]):
            dxv = conv2d_dgrad(dout, weight, ks, mode, Cin)
            if mode == ops.CONV_UPSAMPLE2:
                dxv = sumpool2x2(dxv)
            elif mode == ops.CONV_UNSHUFFLE2:
                dxv = pixel_shuffle2(dxv)
            d0 = dxv[:, :C0] if src1 is not None else dxv
            d1 = dxv[:, C0:] if src1 is not None else None
        dw = conv2d_wgrad(src0, src1, mode, ks, dout, Cin) if ctx.needs_input_grad[2] else None
        db = channel_sums(dout) if ctx.has_bias and ctx.needs_input_grad[3] else None
        return d0, d1, dw, db, None, None, None


class ResBlockFn(torch.autograd.Function):
    """conv3x3 -> GN -> FiLM -> SiLU -> conv3x3 -> GN -> SiLU, + res(cat(src0,src1)) + vec, fused exactly like the
    inference path (GN statistics in the conv epilogue, normalise+SiLU in the next conv's gather)."""

    @staticmethod
    def forward(ctx, src0, src1, film, vec, w1, b1, g1, be1, w2, b2, g2, be2, wr, br, groups, eps, slot=None):
        B, _, H, W = src0.shape
        dst = None if slot is None else slot.t
        Co, HW = w1.shape[0], H * W
        h1, st1 = ops.conv2d(src0, _packed(w1), b1, 3, Co, src1=src1, want_stats=True)
        a1, c1, mr1 = ops.gn_finalize(st1, groups, HW, g1, be1, film=film, eps=eps, want_mean_rstd=True)
        h2, st2 = ops.conv2d(h1, _packed(w2), b2, 3, Co, pro=(a1, c1), want_stats=True)
        a2, c2, mr2 = ops.gn_finalize(st2, groups, HW, g2, be2, eps=eps, want_mean_rstd=True)
        if wr is None:
            out = ops.affine_silu_add(h2, (a2, c2), res=src0, vec=vec, out=dst)
        else:
            out = ops.conv2d(src0, _packed(wr), br, 1, Co, src1=src1, aux=(h2, a2, c2), vec=vec, out=dst)
        ctx.save_for_backward(src0, src1, film, w1, g1, be1, w2, g2, be2, wr, h1, h2, a1, c1, mr1, a2, c2, mr2)
        ctx.groups, ctx.has_vec = groups, vec is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        src0, src1, film, w1, g1, be1, w2, g2, be2, wr, h1, h2, a1, c1, mr1, a2, c2, mr2 = ctx.saved_tensors
        G = ctx.groups
        dout = _samples_contiguous(dout)
        Co = w1.shape[0]
        C0 = src0.shape[1]
        Cin = w1.shape[1]
        # tail: out = silu(a2*h2+c2) + res(x) + vec
        # the bias gradients (plane sums of dh2 / dh1) and the plane sums of dout (gradient of `vec`, bias gradient of the residual
        # 1x1 conv) come out of the GroupNorm backward's own passes: no plane_sum launch over those tensors
        dh2, dg2, dbe2, _, db2, dout_sums = gn_silu_bwd(dout, h2, a2, c2, mr2, g2, be2, None, G, want_sums=True)
        dvec = dout_sums if ctx.has_vec else None
        dw2 = conv2d_wgrad(h1, None, ops.CONV_NORMAL, 3, dh2, Co, pro=(a1, c1))
        dact1 = conv2d_dgrad(dh2, w2, 3, ops.CONV_NORMAL, Co)
        dh1, dg1, dbe1, dfilm, db1, _ = gn_silu_bwd(dact1, h1, a1, c1, mr1, g1, be1, film, G, want_sums=True)
        dw1 = conv2d_wgrad(src0, src1, ops.CONV_NORMAL, 3, dh1, Cin)
        if wr is None:
            dres = dout  # identity residual (single source)
            dwr = dbr = None
        else:
            dres = conv2d_dgrad(dout, wr, 1, ops.CONV_NORMAL, Cin)
            dwr = conv2d_wgrad(src0, src1, ops.CONV_NORMAL, 1, dout, Cin)
            dbr = batch_sum(dout_sums)
        dx = conv2d_dgrad(dh1, w1, 3, ops.CONV_NORMAL, Cin, res=dres)
        d0 = dx[:, :C0] if src1 is not None else dx
        d1 = dx[:, C0:] if src1 is not None else None
        return d0, d1, dfilm, dvec, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, dwr, dbr, None, None, None


class BgemmFn(torch.autograd.Function):
    """C[b] = op(A[b]) . op(B[b]); A, B are 3-D [batch, rows, cols] row-major (possibly strided views with unit
    inner stride); transX -> the stored matrix is the transpose of the operand."""

    @staticmethod
    def forward(ctx, A, B, transA, transB, alpha=1.0):
        batch = A.shape[0]
        M, K = (A.shape[2], A.shape[1]) if transA else (A.shape[1], A.shape[2])
        N = B.shape[1] if transB else B.shape[2]
        assert A.stride(2) == 1 and B.stride(2) == 1 and B.shape[0] == batch
        out = bgemm(A, B, M, N, K, A.stride(1), B.stride(1), transA, transB, A.stride(0), B.stride(0), batch, alpha=alpha)
        ctx.save_for_backward(A, B)
        ctx.tA, ctx.tB, ctx.dims, ctx.alpha = transA, transB, (M, N, K, batch), alpha
        return out

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        tA, tB, al = ctx.tA, ctx.tB, ctx.alpha
        M, N, K, batch = ctx.dims
        dC = dC.contiguous()
        dA = dB = None
        if ctx.needs_input_grad[0]:
            if not tA:  # dA [M,K] = dC [M,N] . op(B)^T [N,K]
                dA = bgemm(dC, B, M, K, N, N, B.stride(1), False, not tB, M * N, B.stride(0), batch, alpha=al)
            else:       # dA stored [K,M] = op(B) [K,N] . dC^T [N,M]
                dA = bgemm(B, dC, K, M, N, B.stride(1), N, tB, True, B.stride(0), M * N, batch, alpha=al)
        if ctx.needs_input_grad[1]:
            if not tB:  # dB [K,N] = op(A)^T [K,M] . dC [M,N]
                dB = bgemm(A, dC, K, N, M, A.stride(1), N, not tA, False, A.stride(0), M * N, batch, alpha=al)
            else:       # dB stored [N,K] = dC^T [N,M] . op(A) [M,K]
                dB = bgemm(dC, A, N, K, M, N, A.stride(1), True, tA, M * N, A.stride(0), batch, alpha=al)
        return dA, dB, None, None, None


# ---- stacked token chains (r05): the same layer of a net's four ScoreMapModule decoders as ONE batch-L launch ----------------------
class StackParamsFn(torch.autograd.Function):
    """nk kinds x L levels of parameter tensors (kind-major: kind k of level l at index k*L + l; the L tensors of a kind share their
    shape) -> nk stacked [L, *shape] tensors, gathered by ONE launch (idiff_gather_segments; the segment table is built once per set
    of addresses and stays on the device).  Backward hands every parameter the view dstack_k[l] -- no copy."""
    _tables = {}

    @staticmethod
    def forward(ctx, L, nk, *params):
        import numpy as np
        assert len(params) == L * nk
        dev = params[0].device
        shapes, sizes = [], []
        for k in range(nk):
            sh = tuple(params[k * L].shape)
            for l in range(L):
                p = params[k * L + l]
                assert tuple(p.shape) == sh and p.is_contiguous() and p.dtype == torch.float32, (k, l, tuple(p.shape), sh)
            shapes.append(sh)
            sizes.append(params[k * L].numel())
        total = L * sum(sizes)
        flat = torch.empty((total,), device=dev, dtype=torch.float32)
        key = tuple(p.data_ptr() for p in params)
        hit = StackParamsFn._tables.get(key)
        if hit is None:
            tab = np.zeros((L * nk, 4), dtype=np.int64)
            o = 0
            for k in range(nk):
                for l in range(L):
                    tab[k * L + l, :3] = (params[k * L + l].data_ptr(), o, sizes[k])
                    o += sizes[k]
            nb = (tab[:, 2] + 4095) // 4096
            tab[:, 3] = np.cumsum(nb) - nb
            if len(StackParamsFn._tables) > 64:
                StackParamsFn._tables.clear()
            hit = StackParamsFn._tables[key] = (torch.from_numpy(tab).to(dev), int(nb.sum()))
        check(_lib.load().idiff_gather_segments(hit[0].data_ptr(), L * nk, hit[1], _p(flat), _stream()), "gather_segments")
        outs, o = [], 0
        for k in range(nk):
            outs.append(flat[o:o + L * sizes[k]].view((L,) + shapes[k]))
            o += L * sizes[k]
        ctx.L, ctx.nk = L, nk
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        L, nk = ctx.L, ctx.nk
        out = [None, None]
        for k in range(nk):
            for l in range(L):
                out.append(None if gs[k] is None else gs[k][l])
        return tuple(out)


class StackFn(torch.autograd.Function):
    """L same-shaped activations -> [L, *shape] (L copy launches); backward = views"""

    @staticmethod
    def forward(ctx, *xs):
        L = len(xs)
        out = torch.empty((L,) + tuple(xs[0].shape), device=xs[0].device, dtype=torch.float32)
        for l, x in enumerate(xs):
            x = x.contiguous()
            ops.axpby(x, x, 1.0, 0.0, out=out[l])
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g[l] for l in range(g.shape[0]))


class UnstackFn(torch.autograd.Function):
    """[L, ...] -> L views; backward gathers the L gradients into one stacked tensor (L copy launches)"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return tuple(x[l].detach() for l in range(x.shape[0]))

    @staticmethod
    def backward(ctx, *gs):
        live = next(g for g in gs if g is not None)
        out = torch.empty(ctx.shape, device=live.device, dtype=torch.float32)
        for l, g in enumerate(gs):
            if g is None:
                ops.axpby(live.contiguous(), live.contiguous(), 0.0, 0.0, out=out[l])
            else:
                g = g.contiguous()
                ops.axpby(g, g, 1.0, 0.0, out=out[l])
        return out


class JoinFn(torch.autograd.Function):
    """L parts that their producers wrote straight into the slices buf[l] of one stacked buffer -> the buffer (no copy); backward = views"""

    @staticmethod
    def forward(ctx, slot, *parts):
        buf = slot.t
        for l, p_ in enumerate(parts):
            assert p_.data_ptr() == buf[l].data_ptr() and p_.numel() == buf[l].numel()
        return buf.detach()

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g[l] for l in range(g.shape[0]))


def _bgemm_bias(A, B, out, M, N, K, lda, ldb, ldc, tA, tB, sA, sB, sC, batch, bias=None, sbias=0):
    with _prof("bgemm", 2.0 * M * N * K * batch):
        check(_lib.load().idiff_bgemm_bias(_p(A), _p(B), _p(out), M, N, K, lda, ldb, ldc, 1 if tA else 0, 1 if tB else 0, sA, sB, sC, batch, 1.0,
                                           _p(bias), sbias, _stream()), "bgemm_bias")
    return out


class BLinearFn(torch.autograd.Function):
    """y[l] = x[l] w[l]^T + b[l] for the L stacked decoders in ONE launch each way: x [L,R,K], w [L,N,K], b [L,N] or None"""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = x.contiguous(), w.contiguous()
        L, R, K = x.shape
        N = w.shape[1]
        assert tuple(w.shape) == (L, N, K) and (b is None or tuple(b.shape) == (L, N))
        y = torch.empty((L, R, N), device=x.device, dtype=torch.float32)
        _bgemm_bias(x, w, y, R, N, K, K, K, N, False, True, R * K, N * K, R * N, L, bias=None if b is None else b.contiguous(), sbias=N)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        L, R, K = x.shape
        N = w.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = bgemm(dy, w, R, K, N, N, K, False, False, R * N, N * K, L)          # dy [R,N] w [N,K]
        if ctx.needs_input_grad[1]:
            dw = bgemm(dy, x, N, K, R, N, K, True, False, R * N, R * K, L)           # dy^T [N,R] x [R,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((L, N), device=dy.device, dtype=torch.float32)
            check(_lib.load().idiff_colsum_g(_p(dy), N, _p(db), L * R, N, L, _stream()), "colsum_g")
        return dx, dw, db


class BLinear3Fn(torch.autograd.Function):
    """[x wq^T | x wk^T | x wv^T] -> packed [L, R, 3N] (bias-free), three batch-L launches; dx accumulates through beta = 1"""

    @staticmethod
    def forward(ctx, x, wq, wk, wv):
        x = x.contiguous()
        L, R, K = x.shape
        N = wq.shape[1]
        y = torch.empty((L, R, 3 * N), device=x.device, dtype=torch.float32)
        for i, w in enumerate((wq, wk, wv)):
            w = w.contiguous()
            assert tuple(w.shape) == (L, N, K)
            _bgemm_bias(x, w, y[:, :, i * N:], R, N, K, K, K, 3 * N, False, True, R * K, N * K, R * 3 * N, L)
        ctx.save_for_backward(x, wq, wk, wv)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wq, wk, wv = ctx.saved_tensors
        dy = dy.contiguous()
        L, R, K = x.shape
        N = wq.shape[1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dws = [None, None, None]
        for i, w in enumerate((wq, wk, wv)):
            dyi = dy[:, :, i * N:]   # base pointer of column block i; row stride 3N, batch stride R*3N
            if dx is not None:
                bgemm(dyi, w.contiguous(), R, K, N, 3 * N, K, False, False, R * 3 * N, N * K, L, out=dx, beta=0.0 if i == 0 else 1.0, ldc=K, sC=R * K)
            if ctx.needs_input_grad[1 + i]:
                dws[i] = bgemm(dyi, x, N, K, R, 3 * N, K, True, False, R * 3 * N, R * K, L)
        return dx, dws[0], dws[1], dws[2]


class BLayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dim of x [L, R, C] with a parameter row per level: gamma, beta [L, C]"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        lib = _lib.load()
        x, gamma, beta = x.contiguous(), gamma.contiguous(), beta.contiguous()
        L, R, Cc = x.shape
        y = torch.empty_like(x)
        mr = torch.empty((L * R, 2), device=x.device, dtype=torch.float32)
        check(lib.idiff_layernorm_rows_g_fwd(_p(x), Cc, _p(gamma), _p(beta), _p(y), Cc, L * R, Cc, eps, _p(mr), L, _stream()), "layernorm_rows_g_fwd")
        ctx.save_for_backward(x, gamma, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mr = ctx.saved_tensors
        dy = dy.contiguous()
        L, R, Cc = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty((L, Cc), device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        check(lib.idiff_layernorm_rows_g_bwd(_p(dy), Cc, _p(x), Cc, _p(gamma), _p(mr), _p(dx), Cc, _p(dg), _p(db), L * R, Cc, L, _stream()),
              "layernorm_rows_g_bwd")
        return dx, dg, db, None


class HeadFoldInLFn(torch.autograd.Function):
    """HeadFoldFn "in" for level l of a STACKED query matrix xq [L, R, heads*dh] (read in place); the L calls of a layer share the
    gradient buffer dxq: each writes its slice, the last one to run hands it to autograd (the others return None)."""

    @staticmethod
    def forward(ctx, xq, w, heads, l, shared):
        w = w.contiguous()
        L, R, HD = xq.shape
        Wd = w.shape[1]
        dh = HD // heads
        y = torch.empty((R, heads, Wd), device=xq.device, dtype=torch.float32)
        bgemm(xq[l], w, R, Wd, dh, HD, Wd, False, False, dh, dh * Wd, heads, out=y, ldc=heads * Wd, sC=Wd)
        ctx.save_for_backward(xq, w)
        ctx.geom, ctx.shared = (R, Wd, dh, heads, l), shared
        shared["pending"] = shared.get("pending", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dy):
        xq, w = ctx.saved_tensors
        R, Wd, dh, heads, l = ctx.geom
        sh = ctx.shared
        dy = dy.contiguous()
        if sh.get("buf") is None:
            sh["buf"] = torch.empty_like(xq)
        bgemm(dy, w, R, dh, Wd, heads * Wd, Wd, False, True, Wd, dh * Wd, heads, out=sh["buf"][l], ldc=heads * dh, sC=dh)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            bgemm(xq[l], dy, dh, Wd, R, heads * dh, heads * Wd, True, False, dh, Wd, heads, out=dw, ldc=Wd, sC=dh * Wd)
        sh["pending"] -= 1
        if sh["pending"] > 0:
            return None, dw, None, None, None
        buf, sh["buf"] = sh["buf"], None
        return buf, dw, None, None, None


class HeadFoldOutLFn(torch.autograd.Function):
    """HeadFoldFn "out" writing into slot.t (= slice l of a stacked buffer): y[r, h*dh:(h+1)*dh] = x[r, h, :] @ w[h*dh:(h+1)*dh, :]^T"""

    @staticmethod
    def forward(ctx, x, w, heads, slot):
        x, w = x.contiguous(), w.contiguous()
        R = x.shape[0]
        Wd = w.shape[1]
        dh = w.shape[0] // heads
        y = slot.t
        assert tuple(y.shape) == (R, heads * dh) and y.is_contiguous()
        bgemm(x, w, R, dh, Wd, heads * Wd, Wd, False, True, Wd, dh * Wd, heads, out=y, ldc=heads * dh, sC=dh)
        ctx.save_for_backward(x, w)
        ctx.geom = (R, Wd, dh, heads)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        R, Wd, dh, heads = ctx.geom
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            bgemm(dy, w, R, Wd, dh, heads * dh, Wd, False, False, dh, dh * Wd, heads, out=dx, ldc=heads * Wd, sC=Wd)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            bgemm(dy, x, dh, Wd, R, heads * dh, heads * Wd, True, False, dh, Wd, heads, out=dw, ldc=Wd, sC=dh * Wd)
        return dx, dw, None, None


class StackRowsFn(torch.autograd.Function):
    """[A ; brow ; 0] -> [Cm, n]: the (C + 1) live rows of the compact memory's weight image above zero padding; backward = row slices"""

    @staticmethod
    def forward(ctx, A, brow, Cm):
        A, brow = A.contiguous(), brow.contiguous()
        Cr, n = A.shape
        assert tuple(brow.shape) == (1, n) and Cm > Cr
        out = torch.empty((Cm, n), device=A.device, dtype=torch.float32)
        ops.axpby(A, A, 1.0, 0.0, out=out[:Cr])
        ops.axpby(brow, brow, 1.0, 0.0, out=out[Cr:Cr + 1])
        if Cm > Cr + 1:
            z = out[Cr + 1:]
            src = A.reshape(-1)[:z.numel()].reshape(z.shape) if A.numel() >= z.numel() else None
            assert src is not None
            ops.axpby(src, src, 0.0, 0.0, out=z)   # 0 * finite weights: the padding rows
        ctx.Cr = Cr
        return out

    @staticmethod
    def backward(ctx, d):
        return d[:ctx.Cr], d[ctx.Cr:ctx.Cr + 1], None


class CompactMemFn(torch.autograd.Function):
    """The compact (C + 1)-row memory of a 64-channel ScoreMapModule level in the TRAINING step (r05; the sampling path has used it
    since r01): m = [xh r ; r ; 0] with xh = LayerNorm_C(feat), r = (xh^T G xh + 2 h.xh + e + eps)^-1/2 -- the 256-wide projection and
    its LayerNorm are never materialised ([B, 72, N] instead of two [B, 256, N] tensors per level and pass).  gram / hvec / evar are
    device-side functions of the memory Linear's weights (unet_autograd._memory_fold), so their gradients flow back to the weights.
    Forward and backward are one fused launch each (idiff_smm_memproj_compact_train_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, feat, g1, b1, gram, hvec, evar, Cm, eps1, eps2):
        lib = _lib.load()
        B, Cc, H, W = feat.shape
        N = H * W
        gram, hvec, evar = gram.contiguous(), hvec.contiguous(), evar.contiguous()
        out = torch.empty((B, Cm, N), device=feat.device, dtype=torch.float32)
        check(lib.idiff_smm_memproj_compact_train_fwd(_p(feat), _bs(feat, "feat"), _p(_c(g1)), _p(_c(b1)), _p(_c(gram)), _p(_c(hvec)), _p(_c(evar)),
                                                      _p(out), B, Cc, N, Cm, eps1, eps2, _stream()), "smm_memproj_compact_train_fwd")
        ctx.save_for_backward(feat, g1, b1, gram, hvec, evar)
        ctx.geom = (Cm, eps1, eps2)
        return out

    @staticmethod
    def backward(ctx, dm):
        lib = _lib.load()
        feat, g1, b1, gram, hvec, evar = ctx.saved_tensors
        Cm, eps1, eps2 = ctx.geom
        dm = dm.contiguous()
        B, Cc, H, W = feat.shape
        N = H * W
        dfeat = torch.empty((B, Cc, H, W), device=feat.device, dtype=torch.float32)
        dpar = torch.empty((Cc * Cc + 3 * Cc + 1,), device=feat.device, dtype=torch.float32)
        ws = torch.empty((lib.idiff_smm_memproj_compact_bwd_ws_floats(B, Cc, N),), device=feat.device, dtype=torch.float32)
        check(lib.idiff_smm_memproj_compact_bwd(_p(feat), _bs(feat, "feat"), _p(g1), _p(b1), _p(gram), _p(hvec), _p(evar), _p(dm), _p(dfeat),
                                                Cc * N, _p(dpar), _p(ws), B, Cc, N, Cm, eps1, eps2, _stream()), "smm_memproj_compact_bwd")
        o = Cc * Cc
        return (dfeat, dpar[o:o + Cc], dpar[o + Cc:o + 2 * Cc], dpar[:o].reshape(Cc, Cc), dpar[o + 2 * Cc:o + 3 * Cc],
                dpar[o + 3 * Cc:o + 3 * Cc + 1].reshape(evar.shape), None, None, None)


class HeadFoldFn(torch.autograd.Function):
    """Per-head product between token rows and a weight's head blocks, with the head dimension living in the STRIDES of the batched GEMM
    (no permute / reshape copies, whose autograd backward is an ATen copy or a zero-fill + add per head):
      fold = "in":   y[r, h, :] = x[r, h*dh:(h+1)*dh] @ w[h*dh:(h+1)*dh, :]          x [R, heads*dh], w [heads*dh, Wd] -> y [R, heads, Wd]
      fold = "out":  y[r, h*dh:(h+1)*dh] = x[r, h, :] @ w[h*dh:(h+1)*dh, :]^T        x [R, heads, Wd], w [heads*dh, Wd] -> y [R, heads*dh]
    (the cross-attention's k / v projections folded onto the class-token queries / outputs; row r = (sample, class token), so
    y.reshape(B, K*heads, Wd) is the per-sample row block the fused attention kernel takes, rows ordered (token, head))."""

    @staticmethod
    def forward(ctx, x, w, heads, fold):
        x, w = x.contiguous(), w.contiguous()
        R = x.shape[0]
        Wd = w.shape[1]
        dh = w.shape[0] // heads
        ctx.save_for_backward(x, w)
        ctx.geom = (R, Wd, dh, heads, fold)
        if fold == "in":
            y = torch.empty((R, heads, Wd), device=x.device, dtype=torch.float32)
            bgemm(x, w, R, Wd, dh, heads * dh, Wd, False, False, dh, dh * Wd, heads, out=y, ldc=heads * Wd, sC=Wd)
        else:
            y = torch.empty((R, heads * dh), device=x.device, dtype=torch.float32)
            bgemm(x, w, R, dh, Wd, heads * Wd, Wd, False, True, Wd, dh * Wd, heads, out=y, ldc=heads * dh, sC=dh)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        R, Wd, dh, heads, fold = ctx.geom
        dy = dy.contiguous()
        dx = dw = None
        if fold == "in":  # y_h = x_h w_h
            if ctx.needs_input_grad[0]:   # dx_h [R, dh] = dy_h [R, Wd] w_h^T
                dx = torch.empty_like(x)
                bgemm(dy, w, R, dh, Wd, heads * Wd, Wd, False, True, Wd, dh * Wd, heads, out=dx, ldc=heads * dh, sC=dh)
            if ctx.needs_input_grad[1]:   # dw_h [dh, Wd] = x_h^T [dh, R] dy_h [R, Wd]
                dw = torch.empty_like(w)
                bgemm(x, dy, dh, Wd, R, heads * dh, heads * Wd, True, False, dh, Wd, heads, out=dw, ldc=Wd, sC=dh * Wd)
        else:             # y_h = x_h w_h^T
            if ctx.needs_input_grad[0]:   # dx_h [R, Wd] = dy_h [R, dh] w_h [dh, Wd]
                dx = torch.empty_like(x)
                bgemm(dy, w, R, Wd, dh, heads * dh, Wd, False, False, dh, dh * Wd, heads, out=dx, ldc=heads * Wd, sC=Wd)
            if ctx.needs_input_grad[1]:   # dw_h [dh, Wd] = dy_h^T [dh, R] x_h [R, Wd]
                dw = torch.empty_like(w)
                bgemm(dy, x, dh, Wd, R, heads * dh, heads * Wd, True, False, dh, Wd, heads, out=dw, ldc=Wd, sC=dh * Wd)
        return dx, dw, None, None


class Linear3Fn(torch.autograd.Function):
    """[x Wq^T | x Wk^T | x Wv^T] as ONE packed [R, 3N] output (bias-free projections of one input): three matrix-core launches into the
    column blocks of one buffer; the backward accumulates dx through the residual input of the transposed-weight linear instead of
    leaving two fan-in adds to autograd."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv):
        _chk(x, "x")
        lib = _lib.load()
        assert x.dim() == 2 and x.stride(1) == 1
        R, K = x.shape
        N = wq.shape[0]
        y = torch.empty((R, 3 * N), device=x.device, dtype=torch.float32)
        for i, w in enumerate((wq, wk, wv)):
            assert tuple(w.shape) == (N, K) and w.stride(1) == 1
            check(lib.idiff_linear_mfma_fwd(_p(x), x.stride(0), _p(w), w.stride(0), None, C.c_void_p(y.data_ptr() + 4 * i * N), 3 * N, R, K, N,
                                            _stream()), "linear_mfma_fwd")
        ctx.save_for_backward(x, wq, wk, wv)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wq, wk, wv = ctx.saved_tensors
        dy = dy.contiguous()
        R, K = x.shape
        N = wq.shape[0]
        dx = None
        dws = [None, None, None]
        for i, w in enumerate((wq, wk, wv)):
            dyi = dy[:, i * N:(i + 1) * N]
            if ctx.needs_input_grad[0]:
                dx = ops.linear_t(dyi, w, res=dx)   # dx += dy_i W_i (w [N,K] is the transposed-weight form of dy -> dx)
            if ctx.needs_input_grad[1 + i]:
                dws[i] = bgemm(dyi, x, N, K, R, dy.stride(0), x.stride(0), True, False, 0, 0, 1).reshape(N, K)
        return dx, dws[0], dws[1], dws[2]


class TokenAttnFn(torch.autograd.Function):
    """self-attention among the few class tokens on a packed projection qkv [B*Nt, 3C] (q | k | v): forward = the sampling path's
    attn_tokens kernel, backward = ONE launch (idiff_attn_tokens_bwd) that recomputes P -- instead of two batched GEMMs + softmax
    forward, four + softmax backward and six head-permute copies per layer."""

    @staticmethod
    def forward(ctx, qkv, B, Nt, heads, scale):
        lib = _lib.load()
        _c(qkv, "qkv")
        C3 = qkv.shape[1]
        Cc = C3 // 3
        out = torch.empty((B * Nt, Cc), device=qkv.device, dtype=torch.float32)
        base = qkv.data_ptr()
        check(lib.idiff_attn_tokens_fwd(C.c_void_p(base), C.c_void_p(base + 4 * Cc), C.c_void_p(base + 8 * Cc), _p(out), B, Nt, Nt, Cc, heads, scale,
                                        C3, C3, _stream()), "attn_tokens_fwd")
        ctx.save_for_backward(qkv)
        ctx.geom = (B, Nt, heads, scale)
        return out

    @staticmethod
    def backward(ctx, d_o):
        lib = _lib.load()
        (qkv,) = ctx.saved_tensors
        B, Nt, heads, scale = ctx.geom
        d_o = d_o.contiguous()
        C3 = qkv.shape[1]
        Cc = C3 // 3
        dqkv = torch.empty_like(qkv)
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        check(lib.idiff_attn_tokens_bwd(C.c_void_p(base), C.c_void_p(base + 4 * Cc), C.c_void_p(base + 8 * Cc), _p(d_o), C.c_void_p(dbase),
                                        C.c_void_p(dbase + 4 * Cc), C.c_void_p(dbase + 8 * Cc), B, Nt, Nt, Cc, heads, scale, C3, C3, Cc, C3, C3,
                                        _stream()), "attn_tokens_bwd")
        return dqkv, None, None, None, None


class SmmXattnFn(torch.autograd.Function):
    """ScoreMapModule cross-attention with the K/V projections folded onto the queries, over the 256-row memory:
    o[b,r,:] = softmax_n(scale * qf[b,r,:] . mem[b,:,n]) . mem[b,:,n]^T  with qf, o [B, rows <= 32, 256], mem [B, 256, N].
    Forward = the sampling path's flash-decoding kernel (+ log-sum-exp), backward = ONE fused pass over the keys
    (idiff_smm_xattn_bwd) instead of the seven batched-GEMM / softmax launches autograd derived per decoder layer."""

    @staticmethod
    def forward(ctx, qf, mem, scale, shared=None):
        """shared: a dict common to the calls that attend to the SAME `mem` (the decoder layers of one ScoreMapModule), created per
        forward pass; it counts the calls ("pending").  Their memory gradients are then summed inside the backward kernel, in backward order, into
        one buffer that the last of them to run hands to autograd (the others return None): no [B,256,N] adds between them."""
        lib = _lib.load()
        qf, mem = qf.contiguous(), mem.contiguous()
        _c(qf), _c(mem)
        ctx.shared = shared
        if shared is not None:
            if shared.get("pending", 0) == 0 and shared.get("buf") is not None:
                raise RuntimeError("SmmXattnFn: a shared memory-gradient buffer is still held from an unfinished backward")
            shared["pending"] = shared.get("pending", 0) + 1
        B, R, Cm = qf.shape
        N = mem.shape[2]
        assert Cm in (72, 256) and tuple(mem.shape[:2]) == (B, Cm) and R <= 32   # 72: the compact (C + 1)-row memory of a 64-channel level
        o = torch.empty_like(qf)
        lse = torch.empty((B, R), device=qf.device, dtype=torch.float32)
        ws = torch.empty((lib.idiff_smm_xattn_ws_floats(B, R, 1, Cm, N),), device=qf.device, dtype=torch.float32)
        with _prof("smm_xattn_fwd", 2 * 2.0 * 32 * Cm * N * B):  # S = qf.mem, o = P.mem^T; query rows padded to the 32-row MFMA tile
            check(lib.idiff_smm_xattn_cm_lse_fwd(_p(qf), _p(mem), _p(o), _p(lse), _p(ws), B, R, Cm, N, scale, _stream()), "smm_xattn_cm_lse_fwd")
        ctx.save_for_backward(qf, mem, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        lib = _lib.load()
        qf, mem, o, lse = ctx.saved_tensors
        d_o = d_o.contiguous()
        B, R, Cm = qf.shape
        N = mem.shape[2]
        dqf = torch.empty_like(qf)
        sh = ctx.shared
        acc = 0
        if sh is None:
            dmem = torch.empty_like(mem)
        else:
            acc = 1 if sh.get("buf") is not None else 0
            dmem = sh["buf"] if acc else torch.empty_like(mem)
            sh["buf"] = dmem
            sh["pending"] -= 1
            # single use: one backward per forward (a second backward over a retained graph would hand the summed buffer out
            # again, or never) -- enforced, not assumed
            if sh["pending"] < 0:
                sh["pending"], sh["buf"] = 0, None
                raise RuntimeError("SmmXattnFn: more backward than forward calls on a shared memory gradient (retain_graph / double "
                                   "backward is not supported by the shared-gradient form; pass shared=None)")
        ws = torch.empty((lib.idiff_smm_xattn_ws_floats(B, R, 1, Cm, N),), device=qf.device, dtype=torch.float32)
        with _prof("smm_xattn_bwd", 5 * 2.0 * 32 * Cm * N * B):  # S, dP, dqf, do^T P, qf^T G: five [32 x Cm] products per key
            check(lib.idiff_smm_xattn_cm_bwd(_p(qf), _p(mem), _p(o), _p(lse), _p(d_o), _p(dqf), _p(dmem), acc, _p(ws), B, R, Cm, N, ctx.scale,
                                             _stream()), "smm_xattn_cm_bwd")
        if sh is not None:
            if sh["pending"] > 0:
                return dqf, None, None, None   # the sum is still growing: the last call to run returns it
            sh["buf"] = None
        return dqf, dmem, None, None


class SoftmaxRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        lib = _lib.load()
        _c(x)
        N = x.shape[-1]
        R = x.numel() // N
        p = torch.empty_like(x)
        check(lib.idiff_softmax_rows_fwd(_p(x), N, _p(p), N, R, N, scale, _stream()), "softmax_rows_fwd")
        ctx.save_for_backward(p)
        ctx.scale = scale
        return p

    @staticmethod
    def backward(ctx, dp):
        lib = _lib.load()
        (p,) = ctx.saved_tensors
        dp = dp.contiguous()
        N = p.shape[-1]
        R = p.numel() // N
        ds = torch.empty_like(p)
        check(lib.idiff_softmax_rows_bwd(_p(p), N, _p(dp), N, _p(ds), N, R, N, ctx.scale, _stream()), "softmax_rows_bwd")
        return ds, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on token rows ([R,K] x [N,K]); x may be a column slice (row-strided)."""

    @staticmethod
    def forward(ctx, x, w, b):
        # matrix cores (ops.linear is the sampling path's wave-per-column form: 20 us per launch at 160 rows)
        _chk(x, "x"), _chk(w, "w")
        assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] == w.shape[1]
        y = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=torch.float32)
        check(_lib.load().idiff_linear_mfma_fwd(_p(x), x.stride(0), _p(w), w.stride(0), _p(_c(b)), _p(y), y.stride(0), x.shape[0], x.shape[1],
                                                w.shape[0], _stream()), "linear_mfma_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = dy.contiguous() if dy.stride(-1) != 1 else dy
        R, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_t(dy, w)  # w [N,K] is the transposed-weight form of the map dy -> dx
        if ctx.needs_input_grad[1]:
            dw = bgemm(dy, x, N, K, R, dy.stride(0), x.stride(0), True, False, 0, 0, 1).reshape(N, K)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((N,), device=dy.device, dtype=torch.float32)
            check(lib.idiff_colsum(_p(dy), dy.stride(0), _p(db), R, N, 0, _stream()), "colsum")
        return dx, dw, db


class LayerNormRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x if x.stride(-1) == 1 else x.contiguous()
        y, mr = ops.layernorm_rows(x, gamma, beta, eps, want_mean_rstd=True)
        ctx.save_for_backward(x, gamma, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mr = ctx.saved_tensors
        dy = dy.contiguous()
        R, Cc = x.shape
        dx = torch.empty((R, Cc), device=x.device, dtype=torch.float32)
        dg = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        check(lib.idiff_layernorm_rows_bwd(_p(dy), Cc, _p(x), x.stride(0), _p(_c(gamma)), _p(mr), _p(dx), Cc, _p(dg), _p(db), R, Cc, 0, _stream()),
              "layernorm_rows_bwd")
        return dx, dg, db, None


class ChanLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mr = ops.chan_layernorm(x, gamma, beta, eps, want_mean_rstd=True)
        ctx.save_for_backward(x, gamma, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mr = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cc, H, W = x.shape
        dx = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
        dg = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        db = torch.empty_like(dg)
        ws = torch.empty((lib.idiff_chan_layernorm_bwd_ws_floats(B, Cc, H * W),), device=x.device, dtype=torch.float32)
        check(lib.idiff_chan_layernorm_bwd(_p(dy), _bs(dy), _p(x), _bs(x, "x"), _p(_c(gamma)), _p(mr), _p(dx), _bs(dx), _p(dg), _p(db), _p(ws), B,
                                           Cc, H * W, 0, _stream()), "chan_layernorm_bwd")
        return dx, dg, db, None


class ChanNormalizeFn(torch.autograd.Function):
    """F.normalize(x, dim=1) on [B,C,*] maps (thread = pixel); also used for token rows viewed as [R,C,1]."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        B, Cc = x.shape[:2]
        HW = x.numel() // (B * Cc)
        y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        nrm = torch.empty((B, HW), device=x.device, dtype=torch.float32)
        check(lib.idiff_chan_normalize_fwd(_p(x), _bs(x, "x") if x.dim() == 4 else Cc * HW, _p(y), _p(nrm), B, Cc, HW, _stream()),
              "chan_normalize_fwd")
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        y, nrm = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cc = y.shape[:2]
        HW = y.numel() // (B * Cc)
        dx = torch.empty(y.shape, device=y.device, dtype=torch.float32)
        check(lib.idiff_chan_normalize_bwd(_p(dy), _p(y), _p(nrm), _p(dx), Cc * HW, B, Cc, HW, _stream()), "chan_normalize_bwd")
        return dx


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.idiff_act_fwd(_p(x), _p(y), x.numel(), act, _stream()), "act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(lib.idiff_act_bwd(_p(dy), _p(x), _p(dx), x.numel(), ctx.act, _stream()), "act_bwd")
        return dx, None


class DropoutState:
    """Philox stream of the training-mode dropouts (one per process: seed + a cumulative counter offset, so successive dropouts of any
    sizes use disjoint counter ranges).  `record`, when set to a list, collects (shape, seed, offset, p) of every dropout in call
    order -- tests use it to inject the very same masks into the oracle."""
    seed = 20240
    offset = 0
    record = None

    @classmethod
    def reset(cls, seed):
        cls.seed, cls.offset = int(seed), 0

    @classmethod
    def next(cls, numel):
        off = cls.offset
        cls.offset += (numel + 3) // 4
        return cls.seed, off


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) in training mode on the HIP dropout kernel: the mask is regenerated from (seed, offset) in the backward pass"""

    @staticmethod
    def forward(ctx, x, p):
        x = x.contiguous()
        ctx.p = float(p)
        ctx.seed, ctx.offset = DropoutState.next(x.numel())
        if DropoutState.record is not None:
            DropoutState.record.append((tuple(x.shape), ctx.seed, ctx.offset, ctx.p))
        return ops.dropout(x, ctx.p, ctx.seed, ctx.offset)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed, ctx.offset), None


def dropout(x, p, training):
    """identity unless training with p > 0 (so eval() and dropout: 0 are bit-identical to the path without it)"""
    return DropoutFn.apply(x, p) if (training and p > 0.0) else x


class AddFn(torch.autograd.Function):
    """a + alpha*b (same shape, contiguous)"""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.alpha = alpha
        return ops.axpby(a.contiguous(), b.contiguous(), 1.0, alpha)

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        db = d if ctx.alpha == 1.0 else ops.axpby(d, d, ctx.alpha, 0.0)
        return d, db, None


def sum_n(ts):
    """sum of 2..4 same-shaped [B, ...] tensors whose samples are contiguous (batch-strided operands allowed) in ONE launch"""
    lib = _lib.load()
    n = len(ts)
    assert 2 <= n <= 4
    ts = [_samples_contiguous(t) for t in ts]
    B = ts[0].shape[0]
    per = ts[0].numel() // B
    out = torch.empty(ts[0].shape, device=ts[0].device, dtype=torch.float32)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    bss = (C.c_int64 * n)(*[(t.stride(0) if B > 1 else per) for t in ts])
    check(lib.idiff_sum_n(ptrs, bss, n, _p(out), per, B, per, _stream()), "sum_n")
    return out


class ForkFn(torch.autograd.Function):
    """x -> n aliases of x for n consumers.  The backward receives the consumers' gradients together and sums them in ONE library
    launch (idiff_sum_n; batch-strided operands are taken as they are) -- autograd's own fan-in would add them pairwise in ATen
    (n - 1 passes over the tensor, plus a copy whenever a gradient is a channel slice of a bigger tensor)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.detach() for _ in range(n))  # plain aliases (no view bookkeeping: the kernels write through raw pointers)

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        while len(live) > 4:   # (not reached by the UNet: at most four consumers)
            live = [sum_n(live[:4])] + live[4:]
        return sum_n(live), None


def fork(x, n):
    """n handles on x for n consumers (see ForkFn); needs 16-byte rows, else the plain tensor n times (autograd sums)"""
    per = x.numel() // x.shape[0]
    if n < 2 or not x.requires_grad or per % 4 != 0:
        return (x,) * n
    return ForkFn.apply(x, n)


class _Slot:
    """a plain (non-tensor) holder: hands a preallocated output buffer into an autograd Function without autograd seeing an input"""

    def __init__(self, t):
        self.t = t


class SkipCatFn(torch.autograd.Function):
    """cat(x, emb) along channels WITHOUT a copy: x and emb were written by their producers straight into the two channel slices of
    one buffer (the inference path's virtual concat); forward hands out that buffer, backward hands each producer its slice of the
    gradient (views; the consumers take batch-strided gradients)."""

    @staticmethod
    def forward(ctx, x, emb, slot):
        buf = slot.t
        assert x.data_ptr() == buf.data_ptr() and emb.data_ptr() == buf[:, x.shape[1]:].data_ptr() and buf.shape[1] == x.shape[1] + emb.shape[1]
        ctx.c0 = x.shape[1]
        return buf.detach()

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c0], g[:, ctx.c0:], None


class AddVecFn(torch.autograd.Function):
    """x[b,c,:,:] + vec[b,c]"""

    @staticmethod
    def forward(ctx, x, vec):
        return ops.affine_silu_add(x, None, vec=vec.contiguous())

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        return d, channel_sums(d, per_sample=True)


class ScaleColsFn(torch.autograd.Function):
    """y[r, n] = g[n] * x[r, n]  (ScoreMapModule gamma)"""

    @staticmethod
    def forward(ctx, x, g):
        lib = _lib.load()
        x = x.contiguous()
        ctx.save_for_backward(x, g)
        R, N = x.shape
        y = torch.empty_like(x)
        check(lib.idiff_scale_cols(_p(x), _p(_c(g)), _p(y), R, N, _stream()), "scale_cols")
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, g = ctx.saved_tensors
        dy = dy.contiguous()
        R, N = x.shape
        dx = torch.empty_like(x)
        check(lib.idiff_scale_cols(_p(dy), _p(g), _p(dx), R, N, _stream()), "scale_cols")
        dg = torch.empty((N,), device=x.device, dtype=torch.float32)
        check(lib.idiff_colsum_prod(_p(dy), _p(x), _p(dg), R, N, _stream()), "colsum_prod")
        return dx, dg


class SelectConvFn(torch.autograd.Function):
    """The UNet's output layer fused with the class pick, as in the sampling path: pred[b] = conv3x3(x[b]; w[idx[b]]) + bias[idx[b]] -- ONE
    output channel per sample (the reference computes out_nc = 5 and keeps one, models/drift_noise_model.py:250-268).  Backward = the
    transposed conv with the sample's kernel and per-class sums of g * X (idiff_conv3x3_select_bwd): no [B,5,H,W] tensors, no scatter,
    no matrix-core tiles padded from 5 channels."""

    @staticmethod
    def forward(ctx, x, weight, bias, idx):
        ctx.save_for_backward(x, weight, idx)
        ctx.has_bias = bias is not None
        return ops.conv3x3_select(x, weight.detach().contiguous(), bias, idx)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, weight, idx = ctx.saved_tensors
        g = g.contiguous()
        B, Cc, H, W = x.shape
        K = weight.shape[0]
        dx = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty((K,), device=x.device, dtype=torch.float32) if (need_w and ctx.has_bias) else None
        ws = torch.empty((lib.idiff_conv3x3_select_bwd_ws_floats(B, Cc, H, W),), device=x.device, dtype=torch.float32) if need_w else None
        check(lib.idiff_conv3x3_select_bwd(_p(x), _bs(x, "x"), _p(_c(weight.detach())), C.c_void_p(idx.data_ptr()), _p(g), _p(dx), Cc * H * W, _p(dw),
                                           _p(db), _p(ws), B, Cc, K, H, W, _stream()), "conv3x3_select_bwd")
        return dx, dw, db, None


class GatherChannelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.C = x.shape[1]
        return ops.gather_channel(x.contiguous(), idx)

    @staticmethod
    def backward(ctx, d):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        d = d.contiguous()
        B, _, H, W = d.shape
        out = torch.empty((B, ctx.C, H, W), device=d.device, dtype=torch.float32)
        check(lib.idiff_scatter_channel(_p(d), _p(idx), _p(out), B, ctx.C, H * W, _stream()), "scatter_channel")
        return out, None


# =====================================================================================================
# losses, optimizer, train step
# =====================================================================================================
def resize_bilinear(x, oh, ow):
    lib = _lib.load()
    _c(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, oh, ow), device=x.device, dtype=torch.float32)
    check(lib.idiff_resize_bilinear(_p(x), _p(out), B * Cc, H, W, oh, ow, _stream()), "resize_bilinear")
    return out


def mse_loss_and_grad(pred, target, loss_slot, weight=1.0, want_grad=True):
    """writes mean((pred-target)^2) into loss_slot (1-element device view); returns weight * d loss / d pred."""
    lib = _lib.load()
    pred, target = pred.contiguous(), target.contiguous()
    assert pred.shape == target.shape, (pred.shape, target.shape)
    grad = torch.empty_like(pred) if want_grad else None
    ws = torch.empty((256,), device=pred.device, dtype=torch.float32)
    check(lib.idiff_mse_loss(_p(pred), _p(target), _p(loss_slot), _p(grad), _p(ws), pred.numel(), weight, _stream()), "mse_loss")
    return grad


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2 weight decay added to the gradient; config.yml:138-143) in ONE kernel launch per
    parameter group: parameters, gradients and both moments live in flat buffers (the parameters' .data / .grad are
    views into them), so the flat RCCL all-reduce buffer, the Adam update and the 1/world gradient scale fuse."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            total = sum(p.numel() for p in ps)
            fp = torch.empty(total, device=dev, dtype=torch.float32)
            fg = torch.zeros(total, device=dev, dtype=torch.float32)
            o = 0
            for p in ps:
                n = p.numel()
                fp[o:o + n].copy_(p.data.reshape(-1))
                p.data = fp[o:o + n].view_as(p)
                p.grad = fg[o:o + n].view_as(p)
                o += n
            self._flat.append(dict(p=fp, g=fg, m=torch.zeros_like(fp), v=torch.zeros_like(fp), step=0, params=ps))
        self.grad_scale = 1.0
        WEIGHT_EPOCH[0] += 1

    def flat_grads(self):
        """the flat gradient buffers, complete: gradients autograd left in tensors of their own are gathered first (one launch)"""
        self._collect()
        return [f['g'] for f in self._flat if f is not None]

    def zero_grad(self, set_to_none=True):
        """set_to_none=True (default): p.grad = None for every parameter: autograd's AccumulateGrad then TAKES the gradient tensor a
        backward produces (no `grad += g` launch per parameter, no zero-fill of the flat buffer); _collect() moves them into the flat
        buffer in ONE launch (a parameter that received no gradient gets zeros) and re-binds p.grad to its view of the flat buffer.
        set_to_none=False: torch.optim semantics -- the flat buffer is zero-filled and every p.grad is (re-)bound to its view of it, so
        a backward ACCUMULATES into it."""
        for f in self._flat:
            if f is None:
                continue
            if set_to_none:
                for p in f['params']:
                    p.grad = None
            else:
                f['g'].zero_()
                o = 0
                for p in f['params']:
                    n = p.numel()
                    p.grad = f['g'][o:o + n].view_as(p)
                    o += n

    @torch.no_grad()
    def _collect(self):
        import numpy as np
        for f in self._flat:
            if f is None:
                continue
            base, o = f['g'].data_ptr(), 0
            recs, todo = [], False
            for p in f['params']:
                n = p.numel()
                g = p.grad
                if g is None:
                    recs.append((0, o, n))
                    todo = True
                elif g.data_ptr() != base + 4 * o:
                    if not (g.is_contiguous() and g.dtype == torch.float32 and g.device == f['g'].device):
                        g = g.contiguous().to(device=f['g'].device, dtype=torch.float32)
                        p.grad = g  # keep it alive until the gather has run (stream-ordered)
                    recs.append((g.data_ptr(), o, n))
                    todo = True
                o += n
            if not todo:
                continue
            if not f['g'].is_cuda:  # CPU (tests of the host logic only): no kernel library behind it
                o = 0
                for p in f['params']:
                    n = p.numel()
                    if p.grad is None:
                        f['g'][o:o + n].zero_()
                    elif p.grad.data_ptr() != base + 4 * o:
                        f['g'][o:o + n].copy_(p.grad.reshape(-1))
                    o += n
            else:
                # The segment table goes up from a persistent PINNED staging buffer with a non-blocking copy: a pageable-memory copy
                # is synchronous for the host, and this point sits right behind the waits on both backward streams -- the host would
                # stall until the whole backward has drained, once per optimizer per iteration (r04 ADVICE).  The staging buffer may
                # be overwritten by the next collect only after this copy has run: an event guards it.
                nrec = len(recs)
                st = f.get('_stage')
                if st is None or st[0].shape[0] < nrec:
                    st = f['_stage'] = (torch.empty((nrec, 4), dtype=torch.int64, pin_memory=True),
                                        torch.empty((nrec, 4), dtype=torch.int64, device=f['g'].device), torch.cuda.Event())
                else:
                    st[2].synchronize()  # (long past: one iteration ago)
                tab = st[0].numpy()[:nrec]
                tab[:, :3] = np.asarray(recs, dtype=np.int64)
                nb = (tab[:, 2] + 4095) // 4096
                tab[:, 3] = np.cumsum(nb) - nb
                dev_tab = st[1][:nrec]
                dev_tab.copy_(st[0][:nrec], non_blocking=True)
                st[2].record()
                check(_lib.load().idiff_gather_segments(dev_tab.data_ptr(), nrec, int(nb.sum()), _p(f['g']), _stream()), "gather_segments")
                f['_keep'] = [p.grad for p in f['params']]  # the gathered tensors stay alive until the next collect: the launch is asynchronous
            o = 0
            for p in f['params']:
                n = p.numel()
                p.grad = f['g'][o:o + n].view_as(p)
                o += n

    @torch.no_grad()
    def step(self, closure=None):
        lib = _lib.load()
        self._collect()
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            f['step'] += 1
            b1, b2 = group['betas']
            check(lib.idiff_adam_step(_p(f['p']), _p(f['g']), _p(f['m']), _p(f['v']), f['p'].numel(), group['lr'], b1, b2, group['eps'],
                                      group['weight_decay'], self.grad_scale, f['step'], _stream()), "adam_step")
        WEIGHT_EPOCH[0] += 1

    def state_dict(self):
        sd = super().state_dict()
        sd['flat'] = [None if f is None else dict(m=f['m'].cpu(), v=f['v'].cpu(), step=f['step']) for f in self._flat]
        return sd

    def load_torch_adam(self, opt_or_state):
        """Adopt the moments / step count / hyper-parameters of a `torch.optim.Adam` (the object itself, as the reference pickles it
        into `{iter}.state`, models/drift_noise_model.py:694-704, or its state_dict()).  Parameters are matched by position: the
        reference builds its optimizers from `net.parameters()` in the same registration order as these nets."""
        sd = opt_or_state.state_dict() if hasattr(opt_or_state, "state_dict") else opt_or_state
        groups, state = sd["param_groups"], sd["state"]
        if len(groups) != len(self.param_groups):
            raise ValueError(f"optimizer has {len(groups)} parameter groups, expected {len(self.param_groups)}")
        for g, mine, f in zip(groups, self.param_groups, self._flat):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in g:
                    mine[k] = tuple(g[k]) if k == "betas" else g[k]
            if f is None:
                continue
            ids = list(g["params"])
            if len(ids) != len(f["params"]):
                raise ValueError(f"optimizer group holds {len(ids)} parameters, the model {len(f['params'])}")
            o, steps = 0, set()
            for pid, p in zip(ids, f["params"]):
                n = p.numel()
                st = state.get(pid)
                if st is not None:  # parameters the reference never stepped have no state yet
                    if st["exp_avg"].numel() != n:
                        raise ValueError(f"moment of parameter {pid} has {st['exp_avg'].numel()} elements, the model's has {n}")
                    f["m"][o:o + n].copy_(st["exp_avg"].reshape(-1).to(torch.float32))
                    f["v"][o:o + n].copy_(st["exp_avg_sq"].reshape(-1).to(torch.float32))
                    steps.add(int(st["step"]))
                o += n
            if len(steps) > 1:
                raise ValueError(f"parameters carry different step counts {sorted(steps)}: one fused step count cannot represent them")
            f["step"] = steps.pop() if steps else 0

    def load_state_dict(self, sd):
        flat = sd.pop('flat', None)
        super().load_state_dict(sd)
        if flat:
            for f, s in zip(self._flat, flat):
                if f is not None and s is not None:
                    f['m'].copy_(s['m'])
                    f['v'].copy_(s['v'])
                    f['step'] = s['step']


def score_map_losses(score_maps, label, loss_rec, slot0, mult=(1, 2, 4, 8), size=None, want_grad=True):
    """optimize_score_map (drift_noise_model.py:234-240): sum_i MSE(sm_i, resize(label, size//m_i)) / 2 with the
    hard-coded 224 replaced by `size` (default: the label's own size).  Bilinear, align_corners=False, no antialiasing
    (torchvision 0.14 tensor semantics, the version the reference's PyTorch 1.13.1 pins).  Returns the per-map gradients
    (already carrying the /2); the un-halved loss values go to loss_rec[slot0 + i]."""
    H, W = label.shape[-2:] if size is None else size
    grads = []
    for i, sm in enumerate(score_maps):
        oh, ow = H // mult[i], W // mult[i]
        tgt = label if (oh, ow) == tuple(label.shape[-2:]) else resize_bilinear(label, oh, ow)
        grads.append(mse_loss_and_grad(sm, tgt, loss_rec[slot0 + i:slot0 + i + 1], weight=0.5, want_grad=want_grad))
    return grads


TRAIN_TWO_STREAMS = bool(int(os.environ.get("IDIFF_TRAIN_TWO_STREAMS", "1")))


def forward_backward_inputRes(model):
    """Forward of both nets, the reference's active objective and its backward (drift_noise_model.py:242-294):
       pred_drift, dsm = drift_net(x_t - LQ, LQ, t, ...) ; pred_noise, nsm = noise_net(x_t - LQ, x_t, t, ...)
       loss = MSE(pred_drift, LQ-GT) + MSE(pred_noise, std_noise) + pyramid(dsm, LQ-GT) + pyramid(nsm, std_noise)
    Losses and their gradients come from the HIP loss kernel; autograd is entered with explicit output gradients.  Leaves the
    parameter gradients in the optimizers' flat buffers -- with the data-parallel exchange of each buffer already STARTED when
    the model has a grad_sync (finish() it before reading them); returns (loss record [10] on the device, forward time, use_dsm,
    use_nsm)."""
    import time
    st = time.time()
    m = model
    xa = ops.axpby(m.drift_noised_x, m.input, 1.0, -1.0)
    tgt_d = ops.axpby(m.input, m.target, 1.0, -1.0)
    t = m.t.reshape(-1).to(torch.float32)
    use_dsm = m.dnet_settings.get("use_dsm", True) and m.dnet_settings["text_module"] == "scoremap"
    use_nsm = m.nnet_settings.get("use_nsm", True) and m.nnet_settings["text_module"] == "scoremap"
    rec = torch.zeros(10, device=m.device, dtype=torch.float32)  # dl, nl, dsm x4, nsm x4
    m.noise_optimizer.zero_grad()
    m.drift_optimizer.zero_grad()
    sync = m.grad_sync

    def fwd(net, xb, target, slot_main, slot_sm, use_sm):
        with torch.enable_grad():
            out = net(xa, xb, t, m.names, m.text_encoder, image_context=m.A_emb)
        pred, sm = out if isinstance(out, tuple) else (out, [])
        outs, grads = [pred], [mse_loss_and_grad(pred, target, rec[slot_main:slot_main + 1])]
        if use_sm:
            outs += list(sm)
            grads += score_map_losses(sm, target, rec, slot_sm)
        return outs, grads

    # The two nets share no parameters and no activations: forward, losses and backward of each run on a stream of their own
    # (autograd replays a node's backward on the stream of its forward), so one net's latency-bound token-side launches and kernel
    # tails overlap with the other's convolutions, as in the sampling loop.  IDIFF_TRAIN_TWO_STREAMS=0: one stream.
    two = TRAIN_TWO_STREAMS and xa.is_cuda
    if two:
        main = torch.cuda.current_stream()
        if getattr(m, "_train_streams", None) is None:
            m._train_streams = (torch.cuda.Stream(), torch.cuda.Stream())
        s1, s2 = m._train_streams
        s1.wait_stream(main)
        s2.wait_stream(main)
        with torch.cuda.stream(s1):
            outs_d, grads_d = fwd(m.drift_net, m.input, tgt_d, 0, 2, use_dsm)
        with torch.cuda.stream(s2):
            outs_n, grads_n = fwd(m.noise_net, m.drift_noised_x, m.std_noise, 1, 6, use_nsm)
        iter_time = time.time() - st  # the reference times the forward only (:246,290)
        # drift first: its flat gradient buffer starts its all-reduce (RCCL's own stream) while the noise net's backward computes
        with torch.cuda.stream(s1):
            torch.autograd.backward(outs_d, grads_d)
        if sync is not None:
            main.wait_stream(s1)
            sync.start(m.drift_optimizer.flat_grads())
        with torch.cuda.stream(s2):
            torch.autograd.backward(outs_n, grads_n)
        main.wait_stream(s1)
        main.wait_stream(s2)
        # the per-parameter gradient tensors autograd left behind -> the flat buffers (one gather launch per optimizer; p.grad is a
        # view of its flat buffer from here on, zeros where a parameter received no gradient)
        flats_d, flats_n = m.drift_optimizer.flat_grads(), m.noise_optimizer.flat_grads()
        if sync is not None:
            sync.start(flats_n)
        return rec, iter_time, use_dsm, use_nsm
    outs_d, grads_d = fwd(m.drift_net, m.input, tgt_d, 0, 2, use_dsm)
    outs_n, grads_n = fwd(m.noise_net, m.drift_noised_x, m.std_noise, 1, 6, use_nsm)
    iter_time = time.time() - st
    # Two independent graph walks: drift first, and its flat gradient buffer starts its all-reduce while the noise net's backward
    # still computes.
    torch.autograd.backward(outs_d, grads_d)
    if sync is not None:
        sync.start(m.drift_optimizer.flat_grads())
    torch.autograd.backward(outs_n, grads_n)
    flats_n = m.noise_optimizer.flat_grads()  # gathers (see the two-stream branch)
    m.drift_optimizer.flat_grads()
    if sync is not None:
        sync.start(flats_n)
    return rec, iter_time, use_dsm, use_nsm


def train_step_inputRes(model):
    """One optimisation step (drift_noise_model.py:242-312): forward_backward_inputRes, the data-parallel gradient exchange,
    two Adam steps, loss bookkeeping with ONE device->host copy."""
    m = model
    rec, iter_time, use_dsm, use_nsm = forward_backward_inputRes(m)
    scale = m.grad_sync.finish() if m.grad_sync is not None else 1.0  # the exchanges were started during the backward
    m.noise_optimizer.grad_scale = m.drift_optimizer.grad_scale = scale
    m.noise_optimizer.step()
    m.drift_optimizer.step()
    for ema in (getattr(m, "dp_ema", None), getattr(m, "np_ema", None), m.dn_ema, m.nn_ema):
        if ema is not None:  # the reference builds the EMA objects but never calls update()
            ema.update()
    r = rec.cpu()  # the step's single device->host synchronisation (the reference does nine .item() calls)
    dl, nl = float(r[0]), float(r[1])
    dsml = float(r[2:6].sum()) / 2.0 if use_dsm else 0.0
    nsml = float(r[6:10].sum()) / 2.0 if use_nsm else 0.0
    loss = dl + nl + dsml + nsml
    li = m.loss_info
    li['latest'].update(l=loss, nsml=nsml, dsml=dsml, nl=nl, dl=dl)
    for k, v in (('l', loss), ('dl', dl), ('nl', nl), ('dsml', dsml), ('nsml', nsml)):
        li['avg'][k] += v
    li['num'] += 1
    return loss, iter_time
