"""Split-f16 GEMM (csrc/dcl_gemm.hip) and the token-major Linears of the Swin backbone built on it (reference models/Swin.py:97-116 Mlp,
:198-230 qkv / proj, :348-375 PatchMerging.reduction): forward, data gradient, weight gradient, fused epilogues."""
import torch
import torch.nn.functional as F

from ..debug import cfg as _dbg      # A/B switches of the tuning tools: one object (mscs_amd/debug.py)
from .ops_common import _stream


# ---- split-f16 GEMM (csrc/dcl_gemm.hip) ---------------------------------------------------------------------------

def gemm_supported(M, N, K, lda, a_kmajor, ldb, b_kmajor):
    from .. import _lib
    return bool(_lib.lib().dcl_gemm_supported(M, N, K, lda, int(a_kmajor), ldb, int(b_kmajor)))


def gemm_f16x3(a, a_kmajor, lda, b, b_kmajor, ldb, M, N, K, out, ldc, a_amax, b_amax, bias=None, batch=1,
               strides=(0, 0, 0), accumulate=False, c_amax=None, splitk=0, a_rowsum=None):
    """out[b][m][n] (+)= bias[n] + sum_k A[b](m, k) B[b](n, k) on dcl_gemm_f16x3 (fp32-equivalent split-f16 MFMA).

    ``a`` / ``b`` are the tensors whose storage holds the operands (used for their data pointers); X_kmajor says whether
    element (row, k) sits at X[row * ldx + k] (True) or X[k * ldx + row] (False).  ``a_amax`` / ``b_amax``: 1-D float
    tensors whose maxima bound max|A| / max|B| (models.amax.amax_of).  splitk = 0: the library's suggestion."""
    from .. import _lib
    L = _lib.lib()
    if splitk == 0:
        splitk = L.dcl_gemm_suggest_splitk(M, N, K, batch)
    ws = None
    if splitk > 1:
        ws = torch.empty(L.dcl_gemm_workspace_floats(M, N, batch, splitk), dtype=torch.float32, device=out.device)
    p = _lib.ptr
    _lib.check(L.dcl_gemm_f16x3(p(a), lda, int(a_kmajor), strides[0], p(b), ldb, int(b_kmajor), strides[1], M, N, K, batch,
                                p(a_amax), a_amax.numel(), p(b_amax), b_amax.numel(), p(bias), p(out), ldc, strides[2],
                                int(accumulate), p(c_amax), splitk, p(ws), p(a_rowsum), _stream(out)), "dcl_gemm_f16x3")
    return out


def gemm_f16x3_ascaled(a, a_kmajor, lda, b, ldb, M, N, K, out, a_amax, b_amax, a_scale, group, c_amax=None, a_rowsum=None,
                       ep=0, aux=None):
    """dcl_gemm_f16x3_ascaled: out [M, N] = (A with a per-token factor) . B, B row-contiguous ([K, N] read as its transpose);
    a_scale [tokens / group] multiplies row t of a k-major A or k row t of a row-contiguous A (see include/dcl_hip.h)."""
    from .. import _lib
    L = _lib.lib()
    splitk = 1 if ep else L.dcl_gemm_suggest_splitk(M, N, K, 1)
    ws = torch.empty(L.dcl_gemm_workspace_floats(M, N, 1, splitk), dtype=torch.float32, device=out.device) if splitk > 1 else None
    p = _lib.ptr
    _lib.check(L.dcl_gemm_f16x3_ascaled(p(a), lda, int(a_kmajor), p(b), ldb, 0, M, N, K, p(a_amax), a_amax.numel(), p(b_amax),
                                        b_amax.numel(), p(out), N, p(c_amax), splitk, p(ws), p(a_rowsum), p(a_scale), int(group),
                                        int(ep), p(aux), _stream(out)), "dcl_gemm_f16x3_ascaled")
    return out


def linear_f16x3(x2, weight, bias=None, tag_out=True):
    """y [M, N] = x2 [M, K] weight[N, K]^T + bias (the forward of nn.Linear on contiguous rows)."""
    from . import amax as _am
    m, k = x2.shape
    n = weight.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=x2.device)
    ca = _am.zeros(1, x2.device) if tag_out else None
    gemm_f16x3(x2, True, k, weight, True, k, m, n, k, y, n, _am.amax_of(x2), _am.amax_of(weight), bias=bias, c_amax=ca)
    if tag_out:
        _am.tag(y, ca)
    return y


def linear_dgrad_f16x3(gy2, weight, scale=None, group=1):
    """dx [M, K] = (gy2 [M, N], rows scaled by scale[row // group]) weight[N, K]."""
    from . import amax as _am
    m, n = gy2.shape
    k = weight.shape[1]
    gx = torch.empty((m, k), dtype=torch.float32, device=gy2.device)
    ca = _am.zeros(1, gy2.device)
    if scale is not None:
        gemm_f16x3_ascaled(gy2, True, n, weight, k, m, k, n, gx, _am.amax_of(gy2), _am.amax_of(weight), scale, group, c_amax=ca)
    else:
        gemm_f16x3(gy2, True, n, weight, False, k, m, k, n, gx, k, _am.amax_of(gy2), _am.amax_of(weight), c_amax=ca)
    return _am.tag(gx, ca)


def linear_wgrad_f16x3(gy2, x2, want_bias=False, scale=None, group=1):
    """dW [N, K] = gy2 [M, N]^T x2 [M, K] (contraction over the M rows, k-split slabs summed in fixed order); with
    ``want_bias`` also db [N] = the column sums of gy2, accumulated by the threads that stage the dy^T operand (no extra
    pass over gy2): returns (dW, db)."""
    from . import amax as _am
    m, n = gy2.shape
    k = x2.shape[1]
    gw = torch.empty((n, k), dtype=torch.float32, device=gy2.device)
    gb = torch.empty((n,), dtype=torch.float32, device=gy2.device) if want_bias else None
    if scale is not None:       # rows of gy2 (the contraction index here) scaled by scale[row // group], the bias gradient too
        gemm_f16x3_ascaled(gy2, False, n, x2, k, n, k, m, gw, _am.amax_of(gy2), _am.amax_of(x2), scale, group, a_rowsum=gb)
    else:
        gemm_f16x3(gy2, False, n, x2, False, k, n, k, m, gw, k, _am.amax_of(gy2), _am.amax_of(x2), a_rowsum=gb)
    return (gw, gb) if want_bias else gw


class _TokenLinear(torch.autograd.Function):
    """y = x W^T + b on [tokens, K] rows: forward, data gradient and weight gradient on dcl_gemm_f16x3.

    The library's fp32 GEMMs run these shapes at 60-110 TFLOP/s (forward / data gradient) and 10-40 TFLOP/s (dW = dY^T X
    reduces over 10^4..10^5 tokens into a tiny [N, K] output: one macro tile per output tile, the whole token axis
    serial); the split-f16 kernel reaches 190-360 with errors below the library's (tools/gemm_shapes.py), the weight
    gradient as k-split slabs summed in fixed order (deterministic).  Operand scales come from absmax tags: the GEMM's own
    epilogue tags its result, LayerNorm tags its output, GELU hands its input's bound through; anything else costs
    one dcl_absmax pass."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import amax as _am
        n, k = weight.shape
        x2 = _am.carry(x, x.reshape(-1, k))                 # (a view is a new tensor object: the tag travels along)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        y = linear_f16x3(x2, weight, bias)
        return _am.carry(y, y.view(*x.shape[:-1], n))

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        n, k = weight.shape
        from . import amax as _am
        gy2 = _am.carry(gy, gy.reshape(-1, n))
        if not gy2.is_contiguous():
            gy2 = _am.carry(gy2, gy2.contiguous())
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            g = linear_dgrad_f16x3(gy2, weight)
            gx = _am.carry(g, g.view(ctx.xshape))
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_gb:
                gw, gb = linear_wgrad_f16x3(gy2, x2, want_bias=True)
            else:
                gw = linear_wgrad_f16x3(gy2, x2)
        elif want_gb:
            gb = gy2.sum(0)
        return gx, gw, gb


def gemm_f16x3_ep(a, b, b_kmajor, M, N, K, out, a_amax, b_amax, ep, bias=None, c_amax=None, out2=None, aux=None,
                  rowscale=None, rows_per_scale=1):
    """dcl_gemm_f16x3_ep: out [M, N] = epilogue(a [M, K] (k-major rows) . b (+ bias)); b is W [N, K] (b_kmajor) or W [K, N] read as
    its transpose.  ep 1: out = v, out2 = gelu(v); ep 2: out = v * gelu'(aux); ep 3: out = aux + rowscale[row // rows_per_scale] * v."""
    from .. import _lib
    p = _lib.ptr
    _lib.check(_lib.lib().dcl_gemm_f16x3_ep(p(a), K, 1, p(b), K if b_kmajor else N, int(b_kmajor), M, N, K, p(a_amax),
                                            a_amax.numel(), p(b_amax), b_amax.numel(), p(bias), p(out), N, p(c_amax), int(ep),
                                            p(out2), p(aux), p(rowscale), int(rows_per_scale), _stream(out)), "dcl_gemm_f16x3_ep")
    return out


def _ep_gemm_ok(m, n, k):
    """The fused-epilogue entry takes one pass over the contraction (no k-split) and 32-bit element offsets into C."""
    from .. import _lib
    return m * n < (1 << 30) and _lib.lib().dcl_gemm_suggest_splitk(m, n, k, 1) == 1


class _TokenLinearResidual(torch.autograd.Function):
    """shortcut + scale * (x W^T + b) in ONE launch: the residual sum of a Swin block's attention half (reference
    models/Swin.py:318, shortcut + drop_path(proj(...))) rides in the projection GEMM's epilogue.  scale: [B] per-sample DropPath
    factors (mask / keep) or None; bound = 1 / keep bounds |scale|."""

    @staticmethod
    def forward(ctx, x, weight, bias, shortcut, scale, bound):
        from . import amax as _am
        n, k = weight.shape
        x2 = _am.carry(x, x.reshape(-1, k))
        m = x2.shape[0]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        ca = _am.zeros(1, x.device)
        gemm_f16x3_ep(x2, weight, True, m, n, k, y, _am.amax_of(x2), _am.amax_of(weight), 3, bias=bias, c_amax=ca,
                      aux=shortcut, rowscale=scale, rows_per_scale=m // scale.numel() if scale is not None else 1)
        _am.tag(y, ca)
        ctx.save_for_backward(x2, weight, scale)
        ctx.has_bias = bias is not None
        ctx.xshape, ctx.bound = x.shape, float(bound)
        return _am.carry(y, y.view(*x.shape[:-1], n))

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        x2, weight, scale = ctx.saved_tensors
        n, k = weight.shape
        gb2, sc, grp = _branch_grad(gy, n, scale, ctx.bound)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            g = linear_dgrad_f16x3(gb2, weight, sc, grp)
            gx = _am.carry(g, g.view(ctx.xshape))
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_gb:
                gw, gb = linear_wgrad_f16x3(gb2, x2, want_bias=True, scale=sc, group=grp)
            else:
                gw = linear_wgrad_f16x3(gb2, x2, scale=sc, group=grp)
        elif want_gb:
            gb = _scaled_rows(gy, n, scale, ctx.bound).sum(0)
        return gx, gw, gb, (gy if ctx.needs_input_grad[3] else None), None, None


ASCALE_IN_GEMM = _dbg.gemm_ascale     # the per-sample factor of a branch's gradient as an operand scale of the backward GEMMs


def _branch_grad(gy, n, scale, bound):
    """(rows, scale, group) for the backward GEMMs of a branch whose output was scaled per sample: the incoming gradient as
    contiguous [M, n] rows UNSCALED plus the factors for dcl_gemm_f16x3_ascaled when the GEMMs can apply them (whole k-steps of 32
    tokens per sample, factors that keep the scaled operand inside the f16 split's range), else the scaled rows and no factors."""
    from . import amax as _am
    if scale is None:
        return _scaled_rows(gy, n, None, bound), None, 1
    g2 = _am.carry(gy, gy.reshape(-1, n))
    if not g2.is_contiguous():
        g2 = _am.carry(g2, g2.contiguous())
    grp = g2.shape[0] // scale.numel()
    if ASCALE_IN_GEMM and grp % 32 == 0 and grp * scale.numel() == g2.shape[0] and bound <= 3.9 and g2.shape[0] % 32 == 0 \
            and scale.numel() <= 64:
        return g2, scale, grp
    return _scaled_rows(gy, n, scale, bound), None, 1


def _scaled_rows(gy, n, scale, bound):
    """gy as contiguous [M, n] rows, times the per-sample factors (the branch's share of a residual sum's gradient); the absmax
    tag travels along (|g scale| <= |g| * bound)."""
    from . import amax as _am
    g2 = _am.carry(gy, gy.reshape(-1, n))
    if not g2.is_contiguous():
        g2 = _am.carry(g2, g2.contiguous())
    if scale is None:
        return g2
    out = (g2.view(scale.numel(), -1, n) * scale.view(-1, 1, 1)).view(-1, n)
    t = _am.tag_of(g2)
    if t is not None:
        _am.tag(out, t * bound)
    return out


class _FusedMlp(torch.autograd.Function):
    """fc2(gelu(fc1(x))) (+ shortcut, scaled per sample) of a Swin Mlp (reference models/Swin.py:62-76, :321) with the
    element-wise passes inside the GEMMs: fc1's epilogue writes the pre-activation h AND gelu(h) (no GELU kernel), fc2's epilogue
    adds the residual (no addcmul), and in the backward fc2's data gradient leaves its epilogue already multiplied by gelu'(h)
    (no gelu_backward kernel) -- per block three passes over [tokens, 4 C] and one over [tokens, C] less."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, shortcut, scale, bound):
        from . import amax as _am
        hd, k = w1.shape
        n = w2.shape[0]
        x2 = _am.carry(x, x.reshape(-1, k))
        m = x2.shape[0]
        dev = x.device
        h = torch.empty((m, hd), dtype=torch.float32, device=dev)
        a = torch.empty((m, hd), dtype=torch.float32, device=dev)
        ch = _am.zeros(1, dev)
        gemm_f16x3_ep(x2, w1, True, m, hd, k, h, _am.amax_of(x2), _am.amax_of(w1), 1, bias=b1, c_amax=ch, out2=a)
        _am.tag(a, ch)                          # |gelu(v)| <= |v|
        if shortcut is not None:
            y = torch.empty((m, n), dtype=torch.float32, device=dev)
            cy = _am.zeros(1, dev)
            gemm_f16x3_ep(a, w2, True, m, n, hd, y, ch, _am.amax_of(w2), 3, bias=b2, c_amax=cy, aux=shortcut, rowscale=scale,
                          rows_per_scale=m // scale.numel() if scale is not None else 1)
            _am.tag(y, cy)
        else:
            y = linear_f16x3(a, w2, b2)
        ctx.save_for_backward(x2, h, a, w1, w2, scale)
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.xshape, ctx.bound, ctx.residual = x.shape, float(bound), shortcut is not None
        return _am.carry(y, y.view(*x.shape[:-1], n))

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        x2, h, a, w1, w2, scale = ctx.saved_tensors
        hd, k = w1.shape
        n = w2.shape[0]
        m = x2.shape[0]
        g2, sc, grp = _branch_grad(gy, n, scale if ctx.residual else None, ctx.bound)
        need = ctx.needs_input_grad
        gx = gw1 = gb1 = gw2 = gb2 = None
        # fc2: weight / bias gradient from (dy, a); data gradient with gelu'(h) applied in its epilogue = fc1's dy
        if need[3]:
            if ctx.has_b2 and need[4]:
                gw2, gb2 = linear_wgrad_f16x3(g2, a, want_bias=True, scale=sc, group=grp)
            else:
                gw2 = linear_wgrad_f16x3(g2, a, scale=sc, group=grp)
        elif ctx.has_b2 and need[4]:
            gb2 = _scaled_rows(gy, n, scale if ctx.residual else None, ctx.bound).sum(0)
        if need[0] or need[1] or (ctx.has_b1 and need[2]):
            gh = torch.empty((m, hd), dtype=torch.float32, device=gy.device)
            cg = _am.zeros(1, gy.device)
            if sc is not None:
                gemm_f16x3_ascaled(g2, True, n, w2, hd, m, hd, n, gh, _am.amax_of(g2), _am.amax_of(w2), sc, grp, c_amax=cg, ep=2, aux=h)
            else:
                gemm_f16x3_ep(g2, w2, False, m, hd, n, gh, _am.amax_of(g2), _am.amax_of(w2), 2, c_amax=cg, aux=h)
            _am.tag(gh, cg)
            if need[0]:
                g = linear_dgrad_f16x3(gh, w1)
                gx = _am.carry(g, g.view(ctx.xshape))
            if need[1]:
                if ctx.has_b1 and need[2]:
                    gw1, gb1 = linear_wgrad_f16x3(gh, x2, want_bias=True)
                else:
                    gw1 = linear_wgrad_f16x3(gh, x2)
            elif ctx.has_b1 and need[2]:
                gb1 = gh.sum(0)
        return gx, gw1, gb1, gw2, gb2, (gy if (ctx.residual and need[5]) else None), None, None


FUSED_MLP = _dbg.fused_mlp  # module switch (A/B runs, tests of the unfused path): False = TokenLinear -> tagged_gelu -> TokenLinear


def fused_mlp_ok(x, fc1, fc2):
    """Both Linears of a Swin Mlp on the split-f16 GEMM with fused epilogues: the TokenLinear conditions for each, and
    products that need no k-split."""
    if not (FUSED_MLP and isinstance(fc1, TokenLinear) and isinstance(fc2, TokenLinear) and fc1.f16x3 and fc2.f16x3
            and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and fc1.weight.dtype == torch.float32 and fc1.weight.requires_grad and fc2.weight.requires_grad
            and _token_gemm_ok(x, fc1.weight)):
        return False
    hd, k = fc1.weight.shape
    n = fc2.weight.shape[0]
    m = x.numel() // k
    # (few tokens per hidden column -- the last stage: the erf evaluations sit exposed at the end of long tiles and cost more than
    # the cache-resident element-wise kernels they replace: +45 / +55 us per launch at 6 400 x 6 144, tools/probes/gemm_ep_time.py)
    return (fc2.weight.shape[1] == hd and n % 32 == 0 and fc2.weight.is_contiguous() and m * max(n, hd) * 4 < (1 << 32)
            and m >= 4 * hd and _ep_gemm_ok(m, hd, k) and _ep_gemm_ok(m, n, hd) and _ep_gemm_ok(m, hd, n))


def fused_mlp(x, fc1, fc2, shortcut=None, scale=None, bound=1.0):
    """fc2(gelu(fc1(x))), or shortcut + scale * that (scale [B] per-sample factors or None); see _FusedMlp."""
    if shortcut is not None:
        shortcut = shortcut.contiguous()
    return _FusedMlp.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, shortcut,
                           scale.reshape(-1).contiguous() if scale is not None else None, bound)


def linear_residual_ok(x, lin):
    n, k = lin.weight.shape
    return (FUSED_MLP and isinstance(lin, TokenLinear) and lin.f16x3 and x.is_cuda and x.dtype == torch.float32
            and torch.is_grad_enabled() and not torch.is_autocast_enabled() and lin.weight.dtype == torch.float32
            and lin.weight.requires_grad and _token_gemm_ok(x, lin.weight) and _ep_gemm_ok(x.numel() // k, n, k))


def linear_residual(x, lin, shortcut, scale=None, bound=1.0):
    """shortcut + scale * lin(x) with the sum in the GEMM's epilogue (see _TokenLinearResidual)."""
    return _TokenLinearResidual.apply(x, lin.weight, lin.bias, shortcut.contiguous(),
                                      scale.reshape(-1).contiguous() if scale is not None else None, bound)


def _token_gemm_ok(x, weight):
    """Shapes the split-f16 GEMM takes for all three products of a Linear: every extent a multiple of 32 (each is the
    contraction of one of them), the token rows contiguous."""
    n, k = weight.shape
    m = x.numel() // k
    return m >= 1024 and m % 32 == 0 and k % 32 == 0 and n % 32 == 0 and x.is_contiguous() and weight.is_contiguous() \
        and m * max(n, k) * 4 < (1 << 32)


class TokenLinear(torch.nn.Linear):
    """nn.Linear (same parameters / state_dict keys) for token-major fp32 CUDA rows in training: the three GEMMs on the
    split-f16 kernel (csrc/dcl_gemm.hip); anything else is nn.Linear.forward.  ``f16x3 = False`` (class switch, the
    eager comparator of tools / tests) keeps the library's GEMMs."""

    f16x3 = True

    def forward(self, x):
        if (self.f16x3 and x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32
                and self.weight.requires_grad and torch.is_grad_enabled() and not torch.is_autocast_enabled()
                and _token_gemm_ok(x, self.weight)):
            return _TokenLinear.apply(x, self.weight, self.bias)
        return super().forward(x)


# ---- LayerNorm over token-major rows (csrc/dcl_layernorm.hip) -----------------------------------------------------

class _TaggedGelu(torch.autograd.Function):
    """nn.GELU() (exact erf form) that hands absmax BOUNDS through in both directions: |gelu(v)| <= |v| and
    |gelu'(v)| <= 1.13, so the tag of the input bounds the output and 1.13 x the tag of the incoming gradient bounds the
    outgoing one -- the fc2 / fc1 GEMMs of a Swin Mlp find their operand scales without a pass over the 4C-wide tensors."""

    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return torch.nn.functional.gelu(h)

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        (h,) = ctx.saved_tensors
        gx = torch.ops.aten.gelu_backward(gy, h, approximate="none")
        t = _am.tag_of(gy)
        if t is not None:
            _am.tag(gx, t * 1.13)
        return gx


def tagged_gelu(h):
    """nn.GELU() (exact erf form); on CUDA fp32 the absmax tags travel through it (see _TaggedGelu)."""
    if not (h.is_cuda and h.dtype == torch.float32):
        return torch.nn.functional.gelu(h)
    from . import amax as _am
    out = _TaggedGelu.apply(h) if (h.requires_grad and torch.is_grad_enabled()) else torch.nn.functional.gelu(h)
    t = _am.tag_of(h)
    if t is not None:
        _am.tag(out, t)
    return out


class LinearTagGroup:
    """absmax tags of all TokenLinear weights of a model by ONE launch per optimizer step (dcl_absmax_multi) instead of one
    small dcl_absmax launch per Linear; ``refresh()`` at the start of the model's forward does nothing while no weight
    was modified."""

    def __init__(self, module: torch.nn.Module):
        self.lins = [m for m in module.modules() if isinstance(m, TokenLinear)]
        self.key = None
        self.tables = None

    def _build(self, dev):
        import numpy as np
        jobs = np.zeros(len(self.lins), dtype=[("x", "<u8"), ("out", "<u8"), ("n", "<i8"), ("fb", "<i4"), ("pad", "<i4")])
        self.amax = torch.zeros(len(self.lins), dtype=torch.float32, device=dev)
        b2j = []
        for i, m in enumerate(self.lins):
            w = m.weight
            jobs[i] = (w.data_ptr(), self.amax[i:i + 1].data_ptr(), w.numel(), len(b2j), 0)
            b2j += [i] * ((w.numel() + 4095) // 4096)
        self.tables = (torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev),
                       torch.tensor(b2j, dtype=torch.int32, device=dev), len(b2j))
        self.ptrs = tuple(m.weight.data_ptr() for m in self.lins)

    def refresh(self):
        from .. import _lib
        from . import amax as _am
        if not self.lins or not self.lins[0].weight.is_cuda or self.lins[0].weight.dtype != torch.float32:
            return
        key = tuple(m.weight._version for m in self.lins)
        ptrs = tuple(m.weight.data_ptr() for m in self.lins)
        if self.tables is None or ptrs != self.ptrs:
            self._build(self.lins[0].weight.device)
            self.key = None
        if key == self.key:
            return
        jobs, b2j, nb = self.tables
        self.amax.zero_()
        _lib.check(_lib.lib().dcl_absmax_multi(_lib.ptr(jobs), _lib.ptr(b2j), nb, _stream(self.amax)), "dcl_absmax_multi")
        for i, m in enumerate(self.lins):
            _am.tag(m.weight, self.amax[i:i + 1])
        self.key = key
