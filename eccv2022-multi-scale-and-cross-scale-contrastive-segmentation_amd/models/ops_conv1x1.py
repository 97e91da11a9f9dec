"""1x1 convolutions as batched split-f16 GEMMs: plain NCHW, from token-major maps, with Dropout2d folded in, pixel-major output
(reference models/Projector.py:59-63, models/UPerNet.py:70-107)."""
import torch
import torch.nn.functional as F

from ..debug import cfg as _dbg      # A/B switches of the tuning tools: one object (mscs_amd/debug.py)
from . import ops_conv as _oc
from .ops_conv import DirectConv2d, _conv1x1_by_gemm, conv1x1_gemm
from .ops_linear import gemm_f16x3, gemm_supported


class _Conv1x1Gemm(torch.autograd.Function):
    """1x1 / stride 1 convolution on NCHW as batched f32 GEMMs in all three directions.  The library's own weight
    gradient for this case goes through an NHWC implicit-GEMM kernel with layout transposes around it (7 ms per
    HRNet-W48 step); in NCHW it is simply dW = sum_n gy_n [Co, HW] @ x_n^T [HW, Ci]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, ci, h, w = x.shape
        co = weight.shape[0]
        y = torch.empty((n, co, h, w), dtype=x.dtype, device=x.device)       # returned as a base tensor, not a view:
        if _conv1x1_by_gemm(co, ci, x, False):                                # callers relu_() it
            from .amax import amax_of
            conv1x1_gemm(x, weight.view(co, ci), y, amax_of(x), amax_of(weight))
        else:
            torch.matmul(weight.view(co, ci), x.view(n, ci, h * w), out=y.view(n, co, h * w))
        if bias is not None:
            y += bias.view(1, co, 1, 1)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = weight.shape[0]
        gy = gy.contiguous()
        g2 = gy.view(n, co, h * w)
        gx = gw = gb = None
        from .amax import amax_of
        hw = h * w
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            if _conv1x1_by_gemm(ci, co, gy, True):
                conv1x1_gemm(gy, weight.view(co, ci), gx, amax_of(gy), amax_of(weight), transposed=True)
            else:
                torch.matmul(weight.view(co, ci).t(), g2, out=gx.view(n, ci, h * w))
        if ctx.needs_input_grad[1]:
            if _oc.GEMM_CONV1X1 and hw % 32 == 0 and co >= 64 and ci >= 64 and ci % 4 == 0 and co * hw * 4 < (1 << 32) \
                    and ci * hw * 4 < (1 << 32):
                # dW = sum_n gy_n [Co, HW] x_n^T: one batched GEMM (both operands k-major: the pixel axis), k-split slabs
                # per image, then the fixed-order sum over the images
                part = torch.empty((n, co, ci), dtype=torch.float32, device=x.device)
                gemm_f16x3(gy, True, hw, x, True, hw, co, ci, hw, part, ci, amax_of(gy), amax_of(x), batch=n,
                           strides=(co * hw, ci * hw, co * ci))
                gw = (part.sum(0) if n > 1 else part[0]).view_as(weight)
            else:
                gw = torch.bmm(g2, x.view(n, ci, h * w).transpose(1, 2)).sum(0).view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum((0, 2))
        return gx, gw, gb


class _Conv1x1FromTokens(torch.autograd.Function):
    """1x1 convolution (no bias) of a map given TOKEN-MAJOR, result NCHW: y[b] [Co, HW] = W [Co, C] . tok[b]^T with tok [B, HW, C]
    -- a Swin stage output as its LayerNorm wrote it (models/Swin.TokenMap) feeding a lateral convolution of the UPerNet decoder
    (reference models/Swin.py:452-455 permute + contiguous, then models/UPerNet.py:88-92 fpn_in).  All three products are batched
    split-f16 GEMMs on the operands as they lie: forward (W k-major, tokens k-major), data gradient dtok[b] [HW, C] = dy[b]^T W
    (both row-contiguous; written token-major, i.e. contiguous for the LayerNorm's backward), weight gradient dW = sum_b dy[b] tok[b]
    (dy k-major over the pixels, tokens row-contiguous).  No layout copy in either direction."""

    @staticmethod
    def forward(ctx, tok, weight, H, W):
        from . import amax as _am
        b, hw, c = tok.shape
        co = weight.shape[0]
        y = torch.empty((b, co, H, W), dtype=torch.float32, device=tok.device)
        ca = _am.zeros(1, tok.device)
        gemm_f16x3(weight, True, c, tok, True, c, co, hw, c, y, hw, _am.amax_of(weight), _am.amax_of(tok), batch=b,
                   strides=(0, hw * c, co * hw), c_amax=ca)
        _am.tag(y, ca)
        ctx.save_for_backward(tok, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import amax as _am
        tok, weight = ctx.saved_tensors
        b, hw, c = tok.shape
        co = weight.shape[0]
        g = _am.carry(gy, gy.contiguous())
        gt = gw = None
        if ctx.needs_input_grad[0]:
            gt = torch.empty_like(tok)
            cg = _am.zeros(1, tok.device)
            gemm_f16x3(g, False, hw, weight, False, c, hw, c, co, gt, c, _am.amax_of(g), _am.amax_of(weight), batch=b,
                       strides=(co * hw, 0, hw * c), c_amax=cg)
            _am.tag(gt, cg)
        if ctx.needs_input_grad[1]:
            part = torch.empty((b, co, c), dtype=torch.float32, device=tok.device)
            gemm_f16x3(g, True, hw, tok, False, c, co, c, hw, part, c, _am.amax_of(g), _am.amax_of(tok), batch=b,
                       strides=(co * hw, hw * c, co * c))
            gw = (part.sum(0) if b > 1 else part[0]).view_as(weight)
        return gt, gw, None, None


def conv1x1_from_tokens_ok(tok, conv, H, W):
    if not (_oc.GEMM_CONV1X1 and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and tok.is_cuda and tok.dtype == torch.float32 and tok.dim() == 3 and tok.is_contiguous()
            and conv.weight.dtype == torch.float32 and torch.is_grad_enabled() and not torch.is_autocast_enabled()):
        return False
    b, hw, c = tok.shape
    co = conv.out_channels
    return (hw == H * W and c == conv.in_channels and c % 32 == 0 and hw % 32 == 0 and co % 4 == 0
            and max(co, c) * hw * 4 < (1 << 32) and gemm_supported(co, hw, c, c, True, c, True)
            and gemm_supported(hw, c, co, hw, False, c, False) and gemm_supported(co, c, hw, hw, True, c, False))


def conv1x1_from_tokens(tok, conv, H, W):
    """conv(tokens as an NCHW map) -> [B, Co, H, W]; see _Conv1x1FromTokens."""
    return _Conv1x1FromTokens.apply(tok, conv.weight, int(H), int(W))


class _FeatureDropoutConv1x1(torch.autograd.Function):
    """conv1x1(dropout2d(x)) without the two passes over x: Dropout2d multiplies whole channels by a per-(sample, channel) factor
    m (0 or 1 / (1 - p)), and a 1x1 convolution is linear in its input channels, so y_n = (W . diag(m_n)) x_n -- the factor moves
    into a per-sample copy of the (tiny) weight matrix.  Backward: dx_n = (W diag(m_n))^T dy_n, dW = sum_n (dy_n x_n^T) diag(m_n).
    UPerNet's classifier (reference models/UPerNet.py:66-68: conv3x3 block -> Dropout2d -> conv1x1) on 16 x 512 x 160 x 160:
    2 x 3 passes over 839 MB less per step.  ``noise`` is the [N, C] factor tensor, drawn by the caller exactly as
    ``F.dropout2d`` draws it (same generator consumption)."""

    @staticmethod
    def forward(ctx, x, weight, bias, noise):
        n, ci, h, w = x.shape
        co = weight.shape[0]
        wb = weight.view(1, co, ci) * noise.view(n, 1, ci)
        y = torch.empty((n, co, h, w), dtype=x.dtype, device=x.device)
        torch.bmm(wb, x.view(n, ci, h * w), out=y.view(n, co, h * w))
        if bias is not None:
            y += bias.view(1, co, 1, 1)
        ctx.save_for_backward(x, wb, noise)
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wb, noise = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = wb.shape[1]
        g2 = gy.contiguous().view(n, co, h * w)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            torch.bmm(wb.transpose(1, 2), g2, out=gx.view(n, ci, h * w))
        if ctx.needs_input_grad[1]:
            hw = h * w
            if _oc.GEMM_CONV1X1 and hw % 32 == 0 and co >= 64 and ci >= 64 and ci % 4 == 0 and co * hw * 4 < (1 << 32) \
                    and ci * hw * 4 < (1 << 32):
                from .amax import amax_of                       # (as _Conv1x1Gemm.backward: per-image products on the split-f16 GEMM;
                part = torch.empty((n, co, ci), dtype=torch.float32, device=x.device)   # the library runs this shape at 10-40 TFLOP/s)
                gy_c = g2.view(n, co, h, w)
                gemm_f16x3(gy_c, True, hw, x, True, hw, co, ci, hw, part, ci, amax_of(gy_c), amax_of(x), batch=n,
                           strides=(co * hw, ci * hw, co * ci))
            else:
                part = torch.bmm(g2, x.view(n, ci, hw).transpose(1, 2))              # [n, co, ci]
            gw = (part * noise.view(n, 1, ci)).sum(0).view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum((0, 2))
        return gx, gw, gb, None


def dropout2d_conv1x1(x, drop, conv):
    """``conv(drop(x))`` for nn.Dropout2d followed by a plain 1x1 convolution; in training with p > 0 on contiguous fp32 CUDA
    maps the channel factors ride in per-sample weights (_FeatureDropoutConv1x1), otherwise the two modules run as they are."""
    if (drop.training and 0.0 < drop.p < 1.0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
            and torch.is_grad_enabled() and not torch.is_autocast_enabled() and _dbg.fold_dropout2d):
        n, c = x.shape[:2]
        # F.dropout2d's draw: noise = empty([N, C, 1, 1]).bernoulli_(1 - p).div_(1 - p)
        noise = x.new_empty((n, c, 1, 1)).bernoulli_(1.0 - drop.p).div_(1.0 - drop.p)
        return _FeatureDropoutConv1x1.apply(x, conv.weight, conv.bias, noise.view(n, c))
    return conv(drop(x))


class _Conv1x1ToNHWC(torch.autograd.Function):
    """The projection heads' last 1x1 convolution (reference models/Projector.py:56-63) with its output written PIXEL-MAJOR:
    y[n, pix, :] = W x[n, :, pix] + b as one batched split-f16 GEMM per direction.  The result is handed out as the reference's
    [n, d, h, w] tensor with channels-last strides (same shape, same values), so that the contrastive loss -- the only reader of
    the embedding (reference losses/DenseContrastiveLossV2.py:113-124) -- gathers ONE contiguous 1-KiB row per sampled pixel
    (K3) and scatters one per pixel in the backward (K6) instead of 256 strided 4-byte accesses (K3 FETCH_SIZE on the NCHW map:
    294 MB per launch for 10 MB of rows, profiles/r03_loss_pmc_fetch.csv).  Backward: the gradient arrives in the same layout
    (K6 writes the strides it was given); dx and dW are GEMMs over it, no layout copy in either direction."""

    @staticmethod
    def forward(ctx, x, weight, bias, wamax):
        from .amax import amax_of
        n, ci, h, w = x.shape
        co = weight.shape[0]
        hw = h * w
        y = torch.empty((n, h, w, co), dtype=torch.float32, device=x.device)
        # both operands with the contraction index (the input channels) OUTERMOST: a ragged K (48 channels = 1.5 k-steps) is
        # only legal for row-contiguous operands, so the (tiny) weight goes in transposed, [ci, co]
        wt = weight.t().contiguous()
        gemm_f16x3(x, False, hw, wt, False, co, hw, co, ci, y, co, amax_of(x), wamax, bias=bias, batch=n,
                   strides=(ci * hw, 0, hw * co), splitk=1)
        ctx.save_for_backward(x, weight, wamax)
        ctx.has_bias = bias is not None
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from .amax import amax_of, carry
        x, weight, wamax = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = weight.shape[0]
        hw = h * w
        g = carry(gy, gy.permute(0, 2, 3, 1))
        if not g.is_contiguous():
            g = g.contiguous()
        ga = amax_of(g)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            gemm_f16x3(weight, False, ci, g, True, co, ci, hw, co, gx, hw, wamax, ga, batch=n,
                       strides=(0, hw * co, ci * hw), splitk=1)
        if ctx.needs_input_grad[1]:
            part = torch.empty((n, co, ci), dtype=torch.float32, device=x.device)
            gemm_f16x3(g, False, co, x, True, hw, co, ci, hw, part, ci, ga, amax_of(x), batch=n,
                       strides=(hw * co, ci * hw, co * ci))
            gw = (part.sum(0) if n > 1 else part[0]).view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.reshape(-1, co).sum(0)
        return gx, gw, gb, None


def conv1x1_nhwc_supported(x, conv):
    """True when ``conv1x1_to_nhwc`` applies: a plain 1x1 convolution on a contiguous fp32 CUDA map whose three GEMM shapes the
    split-f16 kernel takes."""
    if not (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and conv.weight.dtype == torch.float32 and not torch.is_autocast_enabled()):
        return False
    n, ci, h, w = x.shape
    co, hw = conv.out_channels, h * w
    if max(ci, co) * hw * 4 >= (1 << 32):
        return False
    return (gemm_supported(hw, co, ci, hw, False, co, False) and gemm_supported(ci, hw, co, ci, False, co, True)
            and gemm_supported(co, ci, hw, co, False, hw, True))


def conv1x1_to_nhwc(x, conv):
    """conv(x) as an [n, d, h, w] tensor with channels-last strides (see _Conv1x1ToNHWC)."""
    if isinstance(conv, DirectConv2d):
        wamax = conv.packed_weights()[0]
    else:
        from .amax import amax_of
        wamax = amax_of(conv.weight.detach())
    return _Conv1x1ToNHWC.apply(x, conv.weight.view(conv.out_channels, -1), conv.bias, wamax)


class GemmConv1x1(torch.nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) whose 1x1 / stride 1 / unpadded case runs as batched GEMMs for
    contiguous fp32 CUDA inputs in training; anything else falls through to nn.Conv2d.forward."""

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
                and self.weight.dtype == torch.float32 and not torch.is_autocast_enabled()
                and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
            return _Conv1x1Gemm.apply(x, self.weight, self.bias)
        return super().forward(x)


def use_gemm_conv1x1(module: torch.nn.Module) -> torch.nn.Module:
    """Switch every plain 1x1 / stride 1 / pad 0 / groups 1 nn.Conv2d of a module tree to GemmConv1x1 in place."""
    for m in module.modules():
        if type(m) is torch.nn.Conv2d and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0) \
                and m.dilation == (1, 1) and m.groups == 1:
            m.__class__ = GemmConv1x1
    return module
