"""UPerNet decoder (PPM + FPN) over a Swin backbone, auxiliary head and projector placement
``backbone`` / ``fpn`` / ``fused_feats`` -- drop-in for the reference (models/UPerNet.py:14-261):
constructor ``UPerNet(config=graph_dict, experiment=int)``, attributes ``out_stride``,
``projector_model``, ``return_features``, ``align_corners``, ``num_classes``, ``get_intermediate``;
forward returns ``(interm_logits, logits, proj_feats)`` / ``(logits, proj_feats)`` / ... exactly as
the reference does; state_dict keys identical (``backbone.*``, ``fpn.{ppm_conv,ppm_last_conv,fpn_in,
fpn_out,conv_last}.*``, ``aux_head.*``, ``projector_model.project{s}.*``).

Reference quirks kept: the PPM up-sampling always uses ``align_corners=False`` (UPerNet.py:78) while
the FPN top-down path and the logits use the configured value; only Swin backbones are available
(the reference's torchvision ResNet backbones are outside the BASELINE path)."""
import torch
import torch.nn.functional as F
from torch import nn

from ..utils import DATASETS_INFO, printlog
from .fused_bn import FusedBatchNorm2d, bn_act
from .ops import upsample_bilinear, upsample_concat
from .Projector import Projector
from ..debug import cfg as _dbg
from .Swin import SwinTransformer, as_nchw
from .Swin import backbone_config as backbone_config_swin


class _ConvBNAct(nn.Sequential):
    """Sequential(conv, BatchNorm2d[, ReLU]) -- the reference's layout and state_dict keys -- whose forward hands the
    ReLU to the norm layer when that can fuse it (models/fused_bn.py: one statistics + one apply pass on the GPU)."""

    def forward(self, x):
        if len(self) >= 2 and isinstance(self[1], FusedBatchNorm2d) and (len(self) == 2 or isinstance(self[2], nn.ReLU)):
            return bn_act(self[1], self[0](x), relu=len(self) == 3)
        return super().forward(x)


def conv3x3(in_planes, out_planes, batch_norm, relu, stride=1, norm=nn.BatchNorm2d):
    """conv3x3 (+BN) (+ReLU) builder with the reference's Sequential layout (utils/torch_utils.py:107-123)."""
    layers = [nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)]
    if batch_norm:
        layers.append(norm(out_planes))
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return layers[0] if len(layers) == 1 else _ConvBNAct(*layers)


def _num_classes(dataset, experiment):
    names = DATASETS_INFO[dataset].CLASS_INFO[experiment][1]
    return len(names) - 1 if 255 in names.keys() else len(names)


def _conv1x1_bn_relu(cin, cout, norm=nn.BatchNorm2d):
    return _ConvBNAct(nn.Conv2d(cin, cout, kernel_size=1, bias=False), norm(cout), nn.ReLU(inplace=True))


class _MatrixAdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    """nn.AdaptiveAvgPool2d(s) (same bins: [floor(i H / s), ceil((i + 1) H / s)), reference models/UPerNet.py:43-46) as ONE
    matrix product: y[nc, (i, j)] = x[nc, (h, w)] K[(h, w), (i, j)] with K the Kronecker product of the two axes' averaging
    matrices.  Forward and backward are then plain fp32 GEMMs; ATen's backward scatters with float atomics
    (`atomic_adaptive_average_gradinput`: 0.36 ms per pyramid level on the 16 x 1536 x 20 x 20 map of config 5, and not
    reproducible bit for bit).  No parameters: the state_dict is unchanged.  Other inputs (CPU, non-fp32) take ATen's path."""

    _cache = {}

    @staticmethod
    def _axis(n_in: int, n_out: int) -> torch.Tensor:
        m = torch.zeros(n_out, n_in, dtype=torch.float64)
        for i in range(n_out):
            a, b = (i * n_in) // n_out, -((-(i + 1) * n_in) // n_out)
            m[i, a:b] = 1.0 / (b - a)
        return m

    def forward(self, x):
        s = self.output_size
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and isinstance(s, int)):
            return super().forward(x)
        n, c, h, w = x.shape
        key = (h, w, s, x.device)
        k = self._cache.get(key)
        if k is None:
            k = torch.kron(self._axis(h, s), self._axis(w, s)).t().contiguous().to(device=x.device, dtype=torch.float32)
            self._cache[key] = k                                     # [h w, s s]
        return (x.reshape(n * c, h * w) @ k).view(n, c, s, s)


class FPN(nn.Module):
    def __init__(self, config, experiment):
        super().__init__()
        self.dropout = config.get('dropout_rate', 0.0)
        self.align_corners = config.get('align_corners', True)
        self.dataset = config['dataset']
        self.num_classes = _num_classes(self.dataset, experiment)
        self.pool_scales = config.get('pool_scales', [1, 2, 3, 6])
        self.in_channels = config['input_channels']
        self.in_scales = config['input_scales']
        self.ppm_num_ch = config.get('ppm_num_ch', 512)
        self.fpn_num_ch = config.get('fpn_num_ch', 512)
        self.fpn_num_lvl = min(max(config.get('fpn_num_lvl', len(self.in_scales)), 1), len(self.in_scales))
        self.interpolate_result_up = config.get('interpolate_result_up', True)
        self.head_split = bool(config.get('head_split', True))
        self.return_features = True
        # 'hip_decoder' (default on): fused BN(+ReLU) kernels, HIP bilinear resize (forward with the lateral add folded
        # in, deterministic gather backward) and 1x1 convolutions as batched GEMMs -- same parameters / state_dict
        norm = FusedBatchNorm2d if config.get('hip_decoder', True) else nn.BatchNorm2d
        top = self.in_channels[-1]
        self.ppm_pooling = nn.ModuleList([(_MatrixAdaptiveAvgPool2d if config.get('hip_decoder', True)
                                           else nn.AdaptiveAvgPool2d)(s) for s in self.pool_scales])
        self.ppm_conv = nn.ModuleList([_conv1x1_bn_relu(top, self.ppm_num_ch, norm) for _ in self.pool_scales])
        self.ppm_last_conv = conv3x3(top + len(self.pool_scales) * self.ppm_num_ch, self.fpn_num_ch,
                                     batch_norm=True, relu=True, norm=norm)
        self.fpn_in = nn.ModuleList([_conv1x1_bn_relu(c, self.fpn_num_ch, norm)
                                     for c in self.in_channels[-self.fpn_num_lvl:-1]])
        self.fpn_out = nn.ModuleList([nn.Sequential(conv3x3(self.fpn_num_ch, self.fpn_num_ch, True, True, norm=norm))
                                      for _ in range(self.fpn_num_lvl - 1)])
        self.conv_last = nn.Sequential(
            conv3x3(self.fpn_num_lvl * self.fpn_num_ch, self.fpn_num_ch, batch_norm=True, relu=True, norm=norm),
            nn.Dropout2d(self.dropout),
            nn.Conv2d(self.fpn_num_ch, self.num_classes, kernel_size=1))

    def _classify(self, y):
        """conv_last[1:] = Dropout2d -> conv1x1 (reference models/UPerNet.py:66-68); the channel dropout folded into per-sample
        weights of the 1x1 convolution when it is active (models/ops.py dropout2d_conv1x1)."""
        tail = list(self.conv_last)[1:]
        if len(tail) == 2 and isinstance(tail[0], nn.Dropout2d):
            from .ops import dropout2d_conv1x1
            return dropout2d_conv1x1(y, tail[0], tail[1])
        for layer in tail:
            y = layer(y)
        return y

    def _lateral(self, block, x):
        """fpn_in[k](backbone level) (reference models/UPerNet.py:88-92).  A level handed over token-major (models/Swin.TokenMap) goes
        into the 1x1 convolution as it lies: no NCHW copy forward, no copy back in front of the backbone norm's backward."""
        from .Swin import TokenMap
        if isinstance(x, TokenMap):
            from .ops import conv1x1_from_tokens, conv1x1_from_tokens_ok
            if (isinstance(block, _ConvBNAct) and len(block) in (2, 3) and isinstance(block[1], FusedBatchNorm2d)
                    and (len(block) == 2 or isinstance(block[2], nn.ReLU)) and conv1x1_from_tokens_ok(x.tokens, block[0], x.H, x.W)):
                return bn_act(block[1], conv1x1_from_tokens(x.tokens, block[0], x.H, x.W), relu=len(block) == 3)
            x = x.nchw()
        return block(x)

    def forward(self, conv_out):
        from .Swin import as_nchw
        c5 = as_nchw(conv_out[-1])
        size5 = c5.shape[2:]
        ppm = [c5] + [conv(upsample_bilinear(pool(c5), size5, False))
                      for pool, conv in zip(self.ppm_pooling, self.ppm_conv)]
        feature = self.ppm_last_conv(torch.cat(ppm, 1))
        pyramid = [feature]
        for i in range(2, self.fpn_num_lvl + 1):
            lateral = self._lateral(self.fpn_in[-i + 1], conv_out[-i])
            feature = upsample_bilinear(feature, lateral.shape[2:], self.align_corners, add=lateral)
            pyramid.append(self.fpn_out[-i + 1](feature))
        pyramid.reverse()                                       # [P2 .. P5]
        # concat order is [P2, P5, P4, P3]: the reference walks the reversed list from its END
        # (UPerNet.py:96-101), and conv_last's input channels are laid out accordingly
        # (upsample_concat: the up-sampled levels are written straight into their channel slice of the fusion
        # convolution's 2048-channel input and the gradient is read out of the slice in place -- no cat copy of 537 MB)
        parts = [pyramid[0]] + [pyramid[-i + 1] for i in range(2, self.fpn_num_lvl + 1)]
        block = self.conv_last[0]
        conv = block[0] if isinstance(block, nn.Sequential) else block
        from .ops import DirectConv2d, LazyConcat, conv3x3_over_upsampled
        if (self.head_split and isinstance(block, _ConvBNAct) and isinstance(conv, DirectConv2d) and len(parts) > 1
                and parts[0].is_cuda and parts[0].dtype == torch.float32 and not torch.is_autocast_enabled()
                and all(t.is_contiguous() for t in parts)):
            # graph key `head_split` (default on): the fusion convolution over the PARTS -- the channel products of the
            # levels that are >= 4x coarser run at their own resolution (models/ops.py conv3x3_over_upsampled: 47 % of the
            # 2048 -> 512 convolution's multiply-adds move to 1/16 and 1/64 of the pixels); the concatenation is only formed
            # if somebody asks for it (a single-scale projector on 'fused_feats')
            fused = LazyConcat(parts, self.align_corners)
            y = conv3x3_over_upsampled(parts, self.align_corners, conv.weight, conv.bias)
            if isinstance(block[1], FusedBatchNorm2d) and (len(block) == 2 or isinstance(block[2], nn.ReLU)):
                y = bn_act(block[1], y, relu=len(block) == 3)
            else:
                for layer in list(block)[1:]:
                    y = layer(y)
            x = self._classify(y)
        else:
            fused = upsample_concat(parts, self.align_corners)
            x = self._classify(self.conv_last[0](fused))
        if self.return_features:
            return x, pyramid, fused
        return x


class UPerNet(nn.Module):
    eligible_backbones = ['swinT', 'swinS', 'swinB', 'swinL']
    valid_projector_positions = ['fpn', 'backbone', 'fused_feats']

    def __init__(self, config, experiment):
        super().__init__()
        self.config = config
        self.experiment = experiment
        self.out_stride = 32
        self.dataset = config['dataset']
        self.backbone_name = config['backbone']
        self.norm = FusedBatchNorm2d if config.get('hip_decoder', True) else nn.BatchNorm2d
        assert self.backbone_name in self.eligible_backbones, \
            f'backbone must be in {self.eligible_backbones} (torchvision ResNets of the reference are not built here)'
        self.num_classes = _num_classes(self.dataset, experiment)
        self.align_corners = config.get('align_corners', True)
        self.return_backbone_feats = False
        if 'return_all_scales' in config:
            self.return_backbone_feats = config['return_all_scales']
            self.return_features = True
        settings = dict(backbone_config_swin[self.backbone_name])
        settings['pretrained'] = config.get('pretrained', True)
        if 'drop_path_rate' in config:          # extension: the reference fixes 0.3 in its backbone table (Swin.py:31)
            settings['drop_path_rate'] = float(config['drop_path_rate'])
        self.backbone = SwinTransformer(**settings)
        if not config.get('hip_attention', True):       # keep the library (SDPA) attention path: calibration runs
            for m in self.backbone.modules():
                if hasattr(m, 'hip_attention'):
                    m.hip_attention = False
        self.config['input_channels'] = settings['out_channels']
        self.config['input_scales'] = [4, 8, 16, 32]
        self.fpn = FPN(config=self.config, experiment=experiment)
        self._get_aux_head()
        self._get_projector()
        if self.projector_model is not None:
            self.projector_model.lazy = bool(self.config.get('lazy_projector', False))     # see models/Projector.LazyProjection
            # (training on the GPU) embedding maps written pixel-major, handed out with channels-last strides: models/ops.py _Conv1x1ToNHWC
            self.projector_model.nhwc = bool(self.config.get('nhwc_projector', True))
        # the FPN's / aux head's 3x3 convolutions on the direct split-f16 kernels (models/ops.py, fp32-equivalent);
        # same parameters and state_dict.  config['direct_conv'] = False keeps them on the library.
        self._conv_packs = None
        if config.get('direct_conv', True):
            from .ops import ConvPackGroup, use_direct_conv3x3
            use_direct_conv3x3(self.fpn)
            if self.aux_head is not None:
                use_direct_conv3x3(self.aux_head)
            self._conv_packs = ConvPackGroup(self)
        from .ops import LinearTagGroup
        self._linear_tags = LinearTagGroup(self)     # absmax tags of the Swin Linears' weights: one launch per step
        if config.get('hip_decoder', True):
            from .ops import use_gemm_conv1x1
            use_gemm_conv1x1(self.fpn)
            if self.aux_head is not None:
                use_gemm_conv1x1(self.aux_head)
            if self.projector_model is not None:
                # the projector heads (conv1x1 -> ReLU -> BN -> conv1x1 on every pyramid level, reference
                # models/Projector.py:46-51): fused norm kernels (class switch: same parameters / buffers / state_dict keys)
                # and the 512 -> 512 / 512 -> d convolutions as batched split-f16 GEMMs
                for m in self.projector_model.modules():
                    if type(m) is nn.BatchNorm2d:
                        m.__class__ = FusedBatchNorm2d
                use_gemm_conv1x1(self.projector_model)
            if self.projector_model is not None:
                use_gemm_conv1x1(self.projector_model)

    def _get_aux_head(self):
        if 'aux_head' in self.config:
            ah = self.config['aux_head']
            self.aux_in_index = ah['in_index']
            self.aux_in_channels = self.config['input_channels'][self.aux_in_index]
            self.aux_out_channels = ah.get('out_channels', 256)
            self.aux_dropout = ah.get('dropout_rate', 0.0)
            self.aux_head = nn.Sequential(
                nn.Conv2d(self.aux_in_channels, self.aux_out_channels, kernel_size=3, stride=1, padding=1),
                self.norm(self.aux_out_channels), nn.ReLU(inplace=True), nn.Dropout2d(self.aux_dropout),
                nn.Conv2d(self.aux_out_channels, self.num_classes, kernel_size=1, stride=1, padding=0, bias=True))
            self.get_intermediate = True
        else:
            self.in_index = None
            self.aux_head = None
            self.get_intermediate = False

    def _aux(self, x):
        """aux_head = Sequential(conv3x3 + bias, norm, ReLU, Dropout2d, conv1x1): norm + ReLU in one fused pass."""
        h = self.aux_head
        if isinstance(h[1], FusedBatchNorm2d):
            return h[4](h[3](bn_act(h[1], h[0](x), relu=True)))
        return h(x)

    def _get_projector(self):
        self.projector_position = None
        self.projector_model = None
        self.return_features = False
        self.use_ms_projector = False
        if 'projector' in self.config:
            # the reference reads self.backbone_out_channels here, which only its ResNet branches set
            # (UPerNet.py:194 would raise AttributeError for Swin); the fused FPN map has
            # fpn_num_lvl * fpn_num_ch channels, which is what this position actually needs
            self.return_features = True
            self.projector_position = 'fused_feats'
            self.config['projector']['c_in'] = self.fpn.fpn_num_lvl * self.fpn.fpn_num_ch
            self.projector_model = Projector(config=self.config['projector'])
        elif 'ms_projector' in self.config:
            mp = self.config['ms_projector']
            self.ms_projector_scales = mp['scales'] if 'scales' in mp else self.fpn.fpn_num_lvl
            self.return_features = True
            self.use_ms_projector = True
            self.projector_position = mp['position']
            if self.projector_position == 'backbone':
                mp['c_in'] = self.config['input_channels'][:self.ms_projector_scales]
            elif self.projector_position == 'fpn':
                mp['c_in'] = [self.fpn.fpn_num_ch] * self.ms_projector_scales
            else:
                raise ValueError(f"ms_projector position must be 'backbone' or 'fpn', got {self.projector_position}")
            self.projector_model = Projector(config=mp)
        if self.projector_model is not None:
            printlog(f'added projector(s) {self.projector_model.c_in} -> {self.projector_model.d} | '
                     f'position {self.projector_position}')

    def forward(self, x):
        size = x.shape[-2:]
        if self._conv_packs is not None and x.is_cuda:
            self._conv_packs.refresh()
        if x.is_cuda and self.training:
            self._linear_tags.refresh()
        # the decoder's lateral convolutions read the backbone levels token-major (models/Swin.TokenMap; graph key `token_laterals`,
        # default on): every other reader below goes through as_nchw
        self.backbone.token_outputs = bool(self.config.get('token_laterals', True)) and self.config.get('hip_decoder', True) \
            and _dbg.token_laterals
        feats = self.backbone(x)
        logits, fpn_feats, fused = self.fpn(feats)
        # graph key `lazy_logits` (extension, default off = the reference's return values): the logits of both heads stay
        # at their own resolution; this repo's LossWrapper / TwoScaleLoss / metrics apply up-sampling + cross-entropy /
        # arg-max in fused kernels (models/ops.py UpsampledLogits) -- 2 x 16 x 150 x 512 x 512 floats are never written
        lazy = self.config.get('lazy_logits', False) and self.training and logits.is_cuda
        if lazy:
            from .ops import UpsampledLogits
            up = lambda t: UpsampledLogits(t, size, self.align_corners)
        else:
            up = lambda t: upsample_bilinear(t, size, self.align_corners)
        logits = up(logits)
        interm = None
        if self.get_intermediate and self.aux_head is not None:
            interm = up(self._aux(as_nchw(feats[self.aux_in_index])))
        if self.projector_model is not None:
            if self.use_ms_projector:
                if self.projector_position == 'backbone':
                    proj = self.projector_model([as_nchw(f) for f in feats[:self.ms_projector_scales]])
                else:
                    proj = self.projector_model(fpn_feats)
            else:
                proj = self.projector_model(fused.materialize() if hasattr(fused, 'materialize') else fused)
            if self.return_features:
                return (interm, logits, proj) if self.get_intermediate else (logits, proj)
        if self.return_backbone_feats:
            feats = [as_nchw(f) for f in feats]
        if self.get_intermediate:
            return (interm, logits, list(feats)) if self.return_backbone_feats else (interm, logits)
        return (logits, list(feats)[::-1]) if self.return_backbone_feats else logits
