"""Fused bilinear up-sampling + cross-entropy on the 1/4-resolution logits (csrc/dcl_upce.hip; reference models/HRNet.py:638,
losses/LossWrapper.py:26-30, :82)."""
import torch
import torch.nn.functional as F

from ..debug import cfg as _dbg      # A/B switches of the tuning tools: one object (mscs_amd/debug.py)
from .ops_resize import upsample_bilinear


class _UpsampleCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, target, weight, ignore_index, H, W, align_corners, holder):
        from .. import _lib
        L = _lib.lib()
        n, c, h, w = z.shape
        dev = z.device
        lse = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        pred = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
        partial = torch.empty((n * H, 2), dtype=torch.float32, device=dev)
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(L.dcl_upsample_ce_fwd(_lib.ptr(z), n, c, h, w, H, W, 1 if align_corners else 0, _lib.ptr(target),
                                         _lib.ptr(weight), int(ignore_index), _lib.ptr(lse), _lib.ptr(pred),
                                         _lib.ptr(partial), _lib.ptr(out2), _lib.stream_ptr(dev)),
                   "dcl_upsample_ce_fwd")
        ctx.save_for_backward(z, target, weight, lse, out2)
        ctx.geom = (H, W, bool(align_corners), int(ignore_index))
        if holder is not None:
            holder["pred"] = pred
        return out2[0].clone()

    @staticmethod
    def backward(ctx, gout):
        from .. import _lib
        L = _lib.lib()
        z, target, weight, lse, out2 = ctx.saved_tensors
        H, W, align, ignore = ctx.geom
        n, c, h, w = z.shape
        gscale = (gout.reshape(1).to(torch.float32) / out2[1:2]).contiguous()
        dz = torch.empty_like(z)
        _lib.check(L.dcl_upsample_ce_bwd(_lib.ptr(z), n, c, h, w, H, W, 1 if align else 0, _lib.ptr(target),
                                         _lib.ptr(weight), ignore, _lib.ptr(lse), _lib.ptr(gscale), _lib.ptr(dz),
                                         _lib.stream_ptr(z.device)), "dcl_upsample_ce_bwd")
        return dz, None, None, None, None, None, None, None


class UpsampledLogits:
    """Logits that exist at 1/4 resolution only: ``lowres`` [N, C, h, w] (part of the autograd graph) plus the size
    and align_corners flag of the bilinear up-sampling the reference applies to them (models/HRNet.py:638).  Returned
    by HRNet when graph['lazy_logits'] is set (an extension: the default returns the up-sampled tensor like the
    reference); this repo's LossWrapper / TwoScaleLoss / metrics consume it through the fused kernels of
    csrc/dcl_upce.hip, anything else calls ``materialize()``."""

    def __init__(self, lowres, size, align_corners):
        self.lowres, self.size, self.align_corners = lowres, (int(size[0]), int(size[1])), bool(align_corners)
        self.pred = None                                   # uint8 [N, H, W] arg-max map, filled by cross_entropy()
        self._full = None

    @property
    def shape(self):
        return torch.Size((self.lowres.shape[0], self.lowres.shape[1]) + self.size)

    @property
    def device(self):
        return self.lowres.device

    def materialize(self):
        if self._full is None:
            self._full = upsample_bilinear(self.lowres, self.size, self.align_corners)
        return self._full

    def cross_entropy(self, target, weight=None, ignore_index=-100):
        """nn.CrossEntropyLoss(weight, ignore_index)(materialize(), target) without materialising."""
        z = self.lowres
        if not (z.is_cuda and z.dtype == torch.float32 and z.shape[1] <= 255 and not torch.is_autocast_enabled()):
            return F.cross_entropy(self.materialize(), target, weight=weight, ignore_index=ignore_index)
        holder = {}
        loss = _UpsampleCE.apply(z.contiguous(), target.contiguous().long(), None if weight is None else
                                 weight.to(device=z.device, dtype=torch.float32).contiguous(), int(ignore_index),
                                 self.size[0], self.size[1], self.align_corners, holder)
        self.pred = holder["pred"]
        return loss
