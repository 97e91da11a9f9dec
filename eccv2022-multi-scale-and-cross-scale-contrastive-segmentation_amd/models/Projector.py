"""Per-scale projection head: 1x1-conv MLP from backbone features to the d-dimensional embedding
the contrastive loss samples from.  Drop-in for the reference class (models/Projector.py:7-91):
same config keys (``d``, ``c_in`` int | list, ``mlp`` [[k, c, s], ...] with c = -1 meaning c_in,
``use_bn``), same sub-module names (``project`` / ``project{s}``) and Sequential indices, hence the
same state_dict keys; list in -> list out."""
from typing import Union

import torch
from torch import nn

from ..utils import printlog


class LazyProjection:
    """Extension (off = the reference's return value, graph key ``lazy_projector``): the projection head WITHOUT its last
    1x1 convolution applied.  The dense contrastive loss reads the d-dimensional embedding at the few thousand pixels it
    samples per scale (reference losses/DenseContrastiveLossV2.py:123) and nowhere else, and a 1x1 convolution acts pixel
    by pixel: ``rows()`` evaluates it on exactly those pixels -- a [rows, c] x [c, d] product instead of the [n, d, h, w]
    map (403 MB at scale 0 of the benchmark, written once, zero-filled once and read twice per step).  Same values up to
    fp32 round-off; everything that wants the map (other losses, evaluation, user code) calls ``materialize()``.
    Quacks like the map for shape checks: ``shape``, ``dim()``, ``dtype``, ``device``."""

    def __init__(self, hidden: torch.Tensor, conv: nn.Conv2d):
        assert conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1
        self.hidden, self.conv = hidden, conv
        n, _, h, w = hidden.shape
        self.shape = torch.Size((n, conv.out_channels, h, w))
        self.dtype, self.device, self.is_cuda = hidden.dtype, hidden.device, hidden.is_cuda

    def dim(self):
        return 4

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def float(self):
        return self

    def materialize(self) -> torch.Tensor:
        return self.conv(self.hidden)

    def rows(self, pair_b: torch.Tensor, pix: torch.Tensor) -> torch.Tensor:
        """Embeddings [T * V, d] of pixel ``pix[t, v]`` (offset in the h x w plane) of image ``pair_b[t]``, row t * V + v;
        differentiable with respect to the hidden map and the convolution's parameters."""
        n, c, h, w = self.hidden.shape
        T, V = pix.shape
        b = pair_b.long().view(T, 1).expand(T, V).reshape(-1)
        hs = self.hidden.reshape(n, c, h * w)[b, :, pix.long().reshape(-1)]
        wt = self.conv.weight.view(self.conv.out_channels, c)
        return torch.addmm(self.conv.bias, hs, wt.t()) if self.conv.bias is not None else hs @ wt.t()


class Projector(nn.Module):
    lazy = False         # set by the model from its graph key ``lazy_projector``: training forward returns LazyProjection(s)

    def __init__(self, config):
        super().__init__()
        self.d = config['d'] if 'd' in config else 128
        self.c_in = config['c_in']
        assert isinstance(self.c_in, (list, int))
        self.mlp = config['mlp'] if 'mlp' in config else []
        self.use_bn = config['use_bn'] if 'use_bn' in config else False
        self.transformer = config['trans'] if 'trans' in config else False
        if self.transformer:
            raise NotImplementedError("Projector(trans=True) is not part of the MI355X hot path "
                                      "(no shipped config uses it; reference: models/Transformers.py)")
        assert isinstance(self.mlp, list), 'config["mlp"] must be [[k_1, c_1, s_1], ..., [k_n, c_n, s_n]] or []'
        for layer in self.mlp:
            assert isinstance(layer, list) and len(layer) == 3 and layer[2] in [1, 2], \
                f'mlp layers are [kernel, channels, stride], got {layer}'
            if layer[1] > 0:
                assert layer[0] < layer[1], f'kernel size is the first element, got {layer}'
        self.is_ms = isinstance(self.c_in, list)
        if self.is_ms:
            for feat_id, c_in in enumerate(self.c_in):
                setattr(self, f'project{feat_id}', self._head(c_in))
        else:
            self.project = self._head(self.c_in)

    def _head(self, c_in: int) -> nn.Sequential:
        layers = []
        c_prev = c_in
        for layer_id, (k, c_out, s) in enumerate(self.mlp):
            if layer_id == 0 and c_out == -1:
                c_out = c_prev
            p = (k - s + 1) // 2
            layers.append(nn.Conv2d(c_prev, c_out, kernel_size=k, stride=s, padding=p, bias=not self.use_bn))
            layers.append(nn.ReLU(inplace=True))
            if self.use_bn:
                layers.append(nn.BatchNorm2d(c_out, momentum=0.0003))
            c_prev = c_out
        layers.append(nn.Conv2d(c_prev, self.d, kernel_size=1, stride=1))
        printlog(f'Projector head {c_in} -> {self.d} ({len(self.mlp)} hidden layer(s), bn={self.use_bn})')
        return nn.Sequential(*layers)

    nhwc = True          # training on the GPU: the embedding maps are written pixel-major (channels-last strides, same shape
                         # and values as the reference's tensor; models/ops.py _Conv1x1ToNHWC) -- what K3 / K6 gather / scatter

    @staticmethod
    def _hidden(head: nn.Sequential, x: torch.Tensor):
        """head[:-1](x).  A hidden layer is conv -> ReLU(inplace) -> BatchNorm (reference models/Projector.py:46-51); with this
        package's fused norm the ReLU's backward rides in the norm's backward kernel (models/fused_bn.py relu_then_bn)."""
        from .fused_bn import FusedBatchNorm2d, relu_then_bn
        layers = list(head[:-1])
        i = 0
        while i < len(layers):
            if (i + 1 < len(layers) and isinstance(layers[i], nn.ReLU) and layers[i].inplace
                    and isinstance(layers[i + 1], FusedBatchNorm2d) and x.is_cuda):
                x = relu_then_bn(layers[i + 1], x)
                i += 2
            else:
                x = layers[i](x)
                i += 1
        return x

    def _run_head(self, head: nn.Sequential, x: torch.Tensor):
        if self.lazy and self.training and x.is_cuda and x.dtype == torch.float32:
            return LazyProjection(self._hidden(head, x), head[-1])
        if self.nhwc and self.training and x.is_cuda and torch.is_grad_enabled():
            from .ops import conv1x1_nhwc_supported, conv1x1_to_nhwc
            hidden = self._hidden(head, x)
            if conv1x1_nhwc_supported(hidden, head[-1]):
                return conv1x1_to_nhwc(hidden, head[-1])
            return head[-1](hidden)
        return head(x)

    def forward(self, x: Union[list, torch.Tensor]):
        if self.is_ms:
            assert isinstance(x, (list, tuple)), \
                f'if multiscale projector is used a list is expected as input instead got {type(x)}'
            return [self._run_head(getattr(self, f"project{i}"), x_i) for i, x_i in enumerate(x)]
        if isinstance(x, list):
            if len(x) != 1:
                raise ValueError(f'x is {type(x)}, of length {len(x)}')
            x = x[0]
        return self._run_head(self.project, x)
