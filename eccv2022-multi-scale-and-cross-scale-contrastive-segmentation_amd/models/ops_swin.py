"""LayerNorm and 7 x 7 window attention of the Swin blocks (csrc/dcl_layernorm.hip, dcl_winattn*.hip; reference models/Swin.py:198-230,
:275-332)."""
import torch
import torch.nn.functional as F

from ..debug import cfg as _dbg      # A/B switches of the tuning tools: one object (mscs_amd/debug.py)
from .ops_common import _stream


def _amax_mod():
    from . import amax
    return amax


def _ln_forward(x, weight, bias, eps):
    from .. import _lib
    from . import amax as _am
    c = x.shape[-1]
    m = x.numel() // c
    y = torch.empty_like(x)
    stats = torch.empty((2, m), dtype=torch.float32, device=x.device)
    ybuf = _am.zeros(_am.SLOTS, x.device)       # absmax tag of y for the Linear behind the norm (the GEMM's operand scale)
    _am.tag(y, ybuf)
    _lib.check(_lib.lib().dcl_layernorm_fwd(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), m, c, float(eps),
                                            _lib.ptr(y), _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(ybuf),
                                            _stream(x)), "dcl_layernorm_fwd")
    return y, stats


def _ln_backward(gy, x, weight, stats, addend=None):
    """(gx (+ addend), dgamma, dbeta); gx is tagged with its absmax (emitted by the kernel)."""
    from .. import _lib
    from . import amax as _am
    c = x.shape[-1]
    m = x.numel() // c
    gy = gy.contiguous()
    if addend is not None:
        addend = addend.contiguous()
    L = _lib.lib()
    gx = torch.empty_like(x)
    parts = torch.empty((L.dcl_layernorm_bwd_parts(m, c), 2, c), dtype=torch.float32, device=x.device)
    gwb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    gam = _am.zeros(_am.SLOTS, x.device)
    _lib.check(L.dcl_layernorm_bwd(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(weight), _lib.ptr(stats[0]),
                                   _lib.ptr(stats[1]), m, c, _lib.ptr(gx), _lib.ptr(parts), _lib.ptr(gwb),
                                   _lib.ptr(addend), _lib.ptr(gam), _stream(x)), "dcl_layernorm_bwd")
    _am.tag(gx, gam)
    return gx, gwb[0], gwb[1]


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, stats = _ln_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, stats = ctx.saved_tensors
        gx, gw, gb = _ln_backward(gy, x, weight, stats)
        return gx, gw, gb, None


class _LayerNormResidualFn(torch.autograd.Function):
    """(LayerNorm(x), alias of x): for ``x -> norm -> branch`` with ``x`` also feeding the residual sum behind the branch
    (both halves of a Swin block, reference models/Swin.py:286-321).  The backward receives the branch's gradient AND the
    shortcut's and adds them inside the norm's backward kernel -- autograd's own sum of the two would be one more
    element-wise pass (3 tensor passes) per norm."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, stats = _ln_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, stats)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gres):
        x, weight, stats = ctx.saved_tensors
        if gy is None:
            return gres, None, None, None
        gx, gw, gb = _ln_backward(gy, x, weight, stats, addend=gres)
        return gx, gw, gb, None


class FusedLayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm over the last axis (same parameters / state_dict keys) on the HIP kernels of
    csrc/dcl_layernorm.hip for contiguous fp32 CUDA rows of a supported length; anything else (CPU, autocast,
    no affine, several normalised axes) is nn.LayerNorm.forward."""

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and len(self.normalized_shape) == 1 and self.weight is not None
                and self.bias is not None and self.weight.dtype == torch.float32 and x.is_contiguous()
                and not torch.is_autocast_enabled() and x.numel() > 0):
            from .. import _lib
            if _lib.lib().dcl_layernorm_supported(x.shape[-1]):
                return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)

    def with_shortcut(self, x):
        """(self(x), x') where x' carries x into the residual sum: on the HIP path the two gradients of x meet inside the
        norm's backward kernel (see _LayerNormResidualFn); elsewhere x' is x."""
        if (x.is_cuda and x.dtype == torch.float32 and len(self.normalized_shape) == 1 and self.weight is not None
                and self.bias is not None and self.weight.dtype == torch.float32 and x.is_contiguous()
                and not torch.is_autocast_enabled() and x.numel() > 0 and torch.is_grad_enabled() and x.requires_grad):
            from .. import _lib
            if _lib.lib().dcl_layernorm_supported(x.shape[-1]):
                return _LayerNormResidualFn.apply(x, self.weight, self.bias, self.eps)
        return self(x), x


# ---- Swin window attention (csrc/dcl_winattn.hip) -----------------------------------------------------------------

class _WindowAttention(torch.autograd.Function):
    """softmax(q k^T * scale + bias (+ shift mask)) v over 7 x 7 windows of tokens kept in their natural order."""

    @staticmethod
    def forward(ctx, qkv, qkv_bias, bias, H, W, heads, shift, scale):
        from .. import _lib
        L = _lib.lib()
        B, Ltok, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty((B, Ltok, C), dtype=torch.float32, device=qkv.device)
        nW = ((H + 6) // 7) * ((W + 6) // 7)
        lse = torch.empty((B, nW, heads, 49), dtype=torch.float32, device=qkv.device)
        _lib.check(L.dcl_winattn_fwd(_lib.ptr(qkv), _lib.ptr(qkv_bias), _lib.ptr(bias), B, H, W, C, heads, shift,
                                     scale, _lib.ptr(out), _lib.ptr(lse), _lib.stream_ptr(qkv.device)),
                   "dcl_winattn_fwd")
        ctx.save_for_backward(qkv, qkv_bias, bias, lse)
        ctx.geom = (H, W, heads, shift, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .. import _lib
        L = _lib.lib()
        qkv, qkv_bias, bias, lse = ctx.saved_tensors
        H, W, heads, shift, scale = ctx.geom
        B, Ltok, C3 = qkv.shape
        C = C3 // 3
        dout = dout.contiguous()
        npad = L.dcl_winattn_npad(H, W)
        nwaves = L.dcl_winattn_bwd_waves(B, H, W, heads)
        dqkv = torch.empty_like(qkv)
        dpad = torch.empty((B, npad, C3), dtype=torch.float32, device=qkv.device) if npad else None
        part = torch.empty((nwaves, 49, 49), dtype=torch.float32, device=qkv.device)
        from . import amax as _am
        gam = _am.zeros(_am.SLOTS, qkv.device)      # max|dqkv|: the operand scale of the qkv Linear's backward GEMMs
        _lib.check(L.dcl_winattn_bwd(_lib.ptr(qkv), _lib.ptr(qkv_bias), _lib.ptr(bias), _lib.ptr(lse), _lib.ptr(dout),
                                     B, H, W, C, heads, shift, scale, _lib.ptr(dqkv), _lib.ptr(dpad), _lib.ptr(part),
                                     _lib.ptr(gam), _lib.stream_ptr(qkv.device)), "dcl_winattn_bwd")
        _am.tag(dqkv, gam)
        dbias = part.view(nwaves // heads, heads, 49, 49).sum(0) if ctx.needs_input_grad[2] else None
        dqb = None
        if ctx.needs_input_grad[1]:
            dqb = dpad.sum((0, 1)) if npad else torch.zeros_like(qkv_bias)
        return dqkv, dqb, dbias, None, None, None, None, None


def window_attention(qkv, qkv_bias, bias, H, W, heads, shift, scale):
    """qkv [B, H*W, 3C] (projection of the tokens in natural order), qkv_bias [3C] (qkv of zero-padded tokens),
    bias [heads, 49, 49] -> [B, H*W, C]; window 7, head_dim 32, fp32 (csrc/dcl_winattn.hip)."""
    return _WindowAttention.apply(qkv.contiguous(), qkv_bias.contiguous(), bias.contiguous(), int(H), int(W),
                                  int(heads), int(shift), float(scale))
