"""HRNetV2 backbone (W18 / W32 / W48) + segmentation head + (multi-scale) projector.

Drop-in for the reference (models/HRNet.py:56-705, models/hrnet_config.py): same constructor
``HRNet(config=graph_dict, experiment=int)``, same attributes (``out_stride``, ``projector_model``,
``return_features``, ``align_corners``, ``num_classes``), same forward return arity, and the same
module tree, so state_dict keys match the reference's 1,870 entries (``backbone.*``,
``cls_head.{0,1,2}.*``, ``projector_model.project{s}.{0,2,3}.*``) and its checkpoints load.

Architecture (HRNetV2, Sun et al. 2019): a stem of two stride-2 3x3 convs, a bottleneck stage, then
three multi-resolution stages.  Each stage is a chain of exchange modules: every branch runs 4 basic
residual blocks at its own resolution, then every output resolution sums contributions from all
branches (1x1 conv + bilinear up-sampling from coarser branches, strided 3x3 conv chains from finer
ones).  Convolutions run through MIOpen (dense contractions on the matrix cores); ``memory_format``
channels_last is supported end to end so the projector output feeds the loss's gather as contiguous
1-KiB rows.

Reference quirks kept on purpose: exchange modules interpolate with
``align_corners=False`` (they are constructed without the flag, HRNet.py:486-493) while the final
concat and the logits up-sampling use the configured value; ``HRNet`` always builds the W48 backbone
unless ``config['backbone']`` names another factory (an extension: the reference hard-codes it).
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..debug import cfg as _dbg
from ..utils import DATASETS_INFO, printlog
from .Projector import Projector
from .ops import (ConvPackGroup, DirectConv2d, GradToken, LazyConcat, conv3x3_over_upsampled, use_gemm_conv1x1,
                  head_norm_classifier, head_norm_classifier_ok,
                  upsample_bilinear, use_direct_conv3x3, use_direct_conv1x1, upsample_concat, fan_out)
from .amax import record_stream as _amax_record_stream
from .fused_bn import FusedBatchNorm2d, bn_act, bn_act_group, can_group, can_group_static

__all__ = ['hrnet18', 'hrnet32', 'hrnet48', 'HRNet', 'HighResolutionNet', 'MODEL_CONFIGS']


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _stage(modules, branches, block, channels):
    return _Cfg(NUM_MODULES=modules, NUM_BRANCHES=branches, NUM_BLOCKS=[4] * branches,
                NUM_CHANNELS=channels, BLOCK=block, FUSE_METHOD='SUM')


def _arch(w):
    return _Cfg(FINAL_CONV_KERNEL=1,
                STAGE1=_stage(1, 1, 'BOTTLENECK', [64]),
                STAGE2=_stage(1, 2, 'BASIC', [w, 2 * w]),
                STAGE3=_stage(4, 3, 'BASIC', [w, 2 * w, 4 * w]),
                STAGE4=_stage(3, 4, 'BASIC', [w, 2 * w, 4 * w, 8 * w]))


MODEL_CONFIGS = {'hrnet18': _arch(18), 'hrnet32': _arch(32), 'hrnet48': _arch(48)}


class _ConvBN(nn.Sequential):
    """Sequential(conv, norm[, ReLU]) -- same indices / state_dict keys as the reference's Sequentials --
    whose forward hands the ReLU (and an optional residual) to the norm layer when it can fuse them."""

    def forward(self, x, residual=None, relu=False, consumer=None):
        """``consumer``: the convolution that is the ONLY reader of the result -- when it applies norm + ReLU itself while staging its
        operand (DirectConv2d.fuses_input_norm), the normalised tensor is not written (FusedBatchNorm2d, ``defer``)."""
        z = self[0](x)
        relu = relu or len(self) == 3
        defer = consumer is not None and residual is None and relu and _defers(self[1], z, consumer)
        return bn_act(self[1], z, residual=residual, relu=relu, defer=defer)


def _conv_bn(cin, cout, k, stride=1, relu=False, norm=nn.BatchNorm2d):
    layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False), norm(cout)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return _ConvBN(*layers)


# A/B switches of the tuning tools: fields of the one debug configuration object (mscs_amd/debug.py); module-level
# names so that library_kernels_only() and the tools can flip them on one box
_FUSE_RESIDUAL_GRAD = _dbg.fuse_residual_grad
_BRANCH_STREAMS = _dbg.branch_streams
_SIDE_STREAMS = {}
_DEFER_JOIN = _dbg.defer_join
_STAGE_CONTINUITY = _dbg.stage_continuity
_FANOUT_ON_BRANCH_STREAM = _dbg.fanout_on_branch_stream
_BRANCH_STREAM_MAP = list(_dbg.branch_stream_map)     # stream per branch (0 = the main stream); default: one each


def _side_streams(device, n):
    """n persistent side streams per device, shared by all modules."""
    key = (device.type, device.index)
    have = _SIDE_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def _defers(bn, z, conv):
    """conv(relu(bn(z))) without the normalised tensor: the norm is a fused one on its training path and the convolution
    says it applies the map itself (DirectConv2d.fuses_input_norm)."""
    return (isinstance(bn, FusedBatchNorm2d) and isinstance(conv, DirectConv2d) and bn._fusable(z, None)
            and conv.fuses_input_norm(z))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        # x feeds conv1 AND the residual add: its two gradients are summed inside conv1's data-gradient kernel
        # (the token carries bn2's residual gradient there) instead of by a separate autograd add
        tok = None
        if _FUSE_RESIDUAL_GRAD and self.downsample is None and isinstance(self.conv1, DirectConv2d) \
                and isinstance(self.bn2, FusedBatchNorm2d) and self.conv1.fuses_residual_grad(x):
            tok = GradToken()
            out = self.conv1(x, grad_token=tok)
        else:
            out = self.conv1(x)
        # bn1 + ReLU inside conv2's operand staging when conv2 takes it: the normalised tensor is never written
        out = bn_act(self.bn1, out, defer=_defers(self.bn1, out, self.conv2))
        return bn_act(self.bn2, self.conv2(out), residual=identity, grad_token=tok)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        # as in BasicBlock: x feeds conv1 AND the residual add; the token carries bn3's residual gradient into conv1's
        # data-gradient epilogue instead of leaving the sum of two 400 MB tensors to an autograd add
        tok = None
        if _FUSE_RESIDUAL_GRAD and self.downsample is None and isinstance(self.conv1, DirectConv2d) \
                and isinstance(self.bn3, FusedBatchNorm2d) and self.conv1.fuses_residual_grad(x):
            tok = GradToken()
            out = self.conv1(x, grad_token=tok)
        else:
            out = self.conv1(x)
        out = bn_act(self.bn1, out, defer=_defers(self.bn1, out, self.conv2))
        out = bn_act(self.bn2, self.conv2(out))
        return bn_act(self.bn3, self.conv3(out), residual=identity, grad_token=tok)


blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}


def _residual_chain(block, inplanes, planes, n_blocks, norm, stride=1):
    down = None
    if stride != 1 or inplanes != planes * block.expansion:
        down = _conv_bn(inplanes, planes * block.expansion, 1, stride, norm=norm)
    layers = [block(inplanes, planes, stride, down, norm_layer=norm)]
    layers += [block(planes * block.expansion, planes, norm_layer=norm) for _ in range(1, n_blocks)]
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """One exchange module: parallel residual branches followed by all-to-all fusion."""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True, norm_layer=None, align_corners=False):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError(f'NUM_BRANCHES({num_branches}) must match NUM_BLOCKS({len(num_blocks)}), '
                             f'NUM_CHANNELS({len(num_channels)}) and NUM_INCHANNELS({len(num_inchannels)})')
        self.norm_layer = norm_layer or nn.BatchNorm2d
        self.align_corners = align_corners
        self.num_inchannels = list(num_inchannels)
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        branches = []
        for b in range(num_branches):
            branches.append(_residual_chain(block, self.num_inchannels[b], num_channels[b], num_blocks[b],
                                            self.norm_layer))
            self.num_inchannels[b] = num_channels[b] * block.expansion
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._fusion()
        self.relu = nn.ReLU(inplace=True)

    def _fusion(self):
        if self.num_branches == 1:
            return None
        ch = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:       # coarser -> finer: 1x1 conv + BN (bilinear up-sampling in forward)
                    row.append(_conv_bn(ch[j], ch[i], 1, norm=self.norm_layer))
                elif j == i:
                    row.append(None)
                else:           # finer -> coarser: (i - j) stride-2 3x3 convs, ReLU between them
                    steps = [_conv_bn(ch[j], ch[j], 3, 2, relu=True, norm=self.norm_layer)
                             for _ in range(i - j - 1)]
                    steps.append(_conv_bn(ch[j], ch[i], 3, 2, norm=self.norm_layer))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def _groupable(self, x):
        """SyncBatchNorm on several ranks: the norms of the branches at one block depth can share ONE statistics exchange
        per direction (fused_bn.bn_act_group) when every branch is a chain of plain BasicBlocks of equal length."""
        if not (x[0].is_cuda and self.num_branches > 1):
            return False
        chains = [list(br) for br in self.branches]
        if len({len(c) for c in chains}) != 1:
            return False
        if not all(type(blk) is BasicBlock and blk.downsample is None and isinstance(blk.bn1, FusedBatchNorm2d)
                   for c in chains for blk in c):
            return False
        first = [c[0] for c in chains]
        bns = [b.bn1 for b in first]
        # the schedule (and with it the number and order of the SyncBatchNorm collectives) is decided from the model structure
        # and the process group alone; a tensor the fused norm cannot take is an ERROR on this rank, not a silent switch to
        # the other schedule (the peers would wait in a collective this rank never issues)
        if not can_group_static(bns):
            return False
        if not can_group(bns, x):
            raise RuntimeError("HighResolutionModule: stacked SyncBatchNorm schedule selected (several ranks, fused norms) but a "
                               "branch input is not a contiguous float32 CUDA tensor: " +
                               ", ".join(f"{tuple(t.shape)}/{t.dtype}/contig={t.is_contiguous()}" for t in x))
        if _BRANCH_STREAM_MAP:
            raise RuntimeError("DCL_BRANCH_STREAM_MAP is a single-rank tuning switch: the stacked SyncBatchNorm schedule runs one "
                               "stream per branch")
        return True

    def _run_branches_grouped(self, x):
        """Depth-major schedule of the branches for SyncBatchNorm on several ranks: at every block depth the branches'
        convolutions run on their own streams as usual, but the 2 x num_branches statistics exchanges of the depth (bn1
        and bn2 of every branch, forward, and again backward) become 2 stacked all-reduces (reference semantics:
        nn.SyncBatchNorm over the process group, BaseManager.py:450-451 -- 208 of HRNet-W48's 310 norms sit in these
        chains).  Same kernels and arithmetic as the free-running schedule."""
        nb = self.num_branches
        main = torch.cuda.current_stream(x[0].device)
        side = _side_streams(x[0].device, nb - 1)
        streams = [None] + list(side) if _BRANCH_STREAMS else [None] * nb
        for k in range(1, nb):
            if streams[k] is not None and getattr(x[k], '_dcl_stream', None) is not streams[k]:
                streams[k].wait_stream(main)
        cur = list(x)
        for k in range(1, nb):
            if streams[k] is not None:
                _amax_record_stream(cur[k], streams[k])

        def on(k):
            return torch.cuda.stream(streams[k]) if streams[k] is not None else contextlib.nullcontext()

        depth = len(self.branches[0])
        for d in range(depth):
            blocks = [self.branches[b][d] for b in range(nb)]
            toks, z = [], []
            for b, blk in enumerate(blocks):
                with on(b):
                    tok = None
                    if _FUSE_RESIDUAL_GRAD and isinstance(blk.conv1, DirectConv2d) and blk.conv1.fuses_residual_grad(cur[b]):
                        tok = GradToken()
                        z.append(blk.conv1(cur[b], grad_token=tok))
                    else:
                        z.append(blk.conv1(cur[b]))
                    toks.append(tok)
            a = bn_act_group([blk.bn1 for blk in blocks], z, relu=True, streams=streams,
                             defers=[_defers(blk.bn1, z[b], blk.conv2) for b, blk in enumerate(blocks)])
            z2 = []
            for b, blk in enumerate(blocks):
                with on(b):
                    z2.append(blk.conv2(a[b]))
            cur = bn_act_group([blk.bn2 for blk in blocks], z2, residuals=cur, relu=True, tokens=toks, streams=streams)
        for k in range(1, nb):
            if streams[k] is not None:
                main.wait_stream(streams[k])
                _amax_record_stream(cur[k], main)
        return cur

    def _run_branches(self, x):
        """The branches of a module are independent until the fuse layers: on CUDA each runs on its own HIP stream
        (the low-resolution branches launch far fewer workgroups than the chip has CUs, so their kernels overlap
        with each other and with the high-resolution branch); autograd replays the same streams in the backward.
        Tensors that cross streams are registered with the caching allocator (record_stream)."""
        if self._groupable(x):
            return self._run_branches_grouped(x)
        if not (_BRANCH_STREAMS and x[0].is_cuda):
            return [branch(xi) for branch, xi in zip(self.branches, x)]
        main = torch.cuda.current_stream(x[0].device)
        side = _side_streams(x[0].device, self.num_branches - 1)
        outs = [None] * self.num_branches
        smap = _BRANCH_STREAM_MAP[:self.num_branches] if _BRANCH_STREAM_MAP else list(range(self.num_branches))
        used = sorted({k for k in smap if k > 0})
        for k in used:
            # x[k] carrying the mark of side stream k - 1 is the previous module's fused output k (or a transition's
            # output), produced on that very stream: nothing to wait for
            if not (_DEFER_JOIN and not _BRANCH_STREAM_MAP and getattr(x[k], '_dcl_stream', None) is side[k - 1]):
                side[k - 1].wait_stream(main)
        for i in range(self.num_branches - 1, -1, -1):         # smallest branch first: it has the most to gain
            if smap[i] == 0:
                continue
            s = side[smap[i] - 1]
            with torch.cuda.stream(s):
                _amax_record_stream(x[i], s)
                outs[i] = self.branches[i](x[i])
        for i in range(self.num_branches - 1, -1, -1):
            if smap[i] == 0:
                outs[i] = self.branches[i](x[i])
        for k in used:
            main.wait_stream(side[k - 1])
        for i in range(self.num_branches):
            if smap[i] > 0:
                _amax_record_stream(outs[i], main)
        return outs

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        x = self._run_branches(x)
        nrow = len(self.fuse_layers)
        if nrow >= 3 and x[0].is_cuda:
            # every branch output feeds every fuse row: hand each row its own alias, so that the nrow gradients of a
            # branch output are summed by one kernel instead of nrow - 1 autograd adds (ops.fan_out)
            if _FANOUT_ON_BRANCH_STREAM and _BRANCH_STREAMS and not _BRANCH_STREAM_MAP:
                # the alias node of branch j is recorded on branch j's stream: its backward (the sum of the rows'
                # gradients) then runs there, in parallel with the other branches' sums and in order with the branch's
                # own backward, instead of all four on the main stream
                side_f = _side_streams(x[0].device, self.num_branches - 1)
                al = [fan_out(x[0], nrow)]
                for j in range(1, self.num_branches):
                    with torch.cuda.stream(side_f[j - 1]):
                        al.append(fan_out(x[j], nrow))
            else:
                al = [fan_out(t, nrow) for t in x]
            xs = [[al[j][i] for j in range(self.num_branches)] for i in range(nrow)]
        else:
            xs = [x] * nrow
        if not (_BRANCH_STREAMS and x[0].is_cuda) or len(self.fuse_layers) == 1:
            return [self._fuse_row(i, row, xs[i]) for i, row in enumerate(self.fuse_layers)]
        # the fused outputs are independent of each other as well: output i on stream i
        main = torch.cuda.current_stream(x[0].device)
        side = _side_streams(x[0].device, len(self.fuse_layers) - 1)
        fused = [None] * len(self.fuse_layers)
        for i in range(len(self.fuse_layers) - 1, 0, -1):
            s = side[i - 1]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                for t in x:
                    _amax_record_stream(t, s)
                fused[i] = self._fuse_row(i, self.fuse_layers[i], xs[i])
        fused[0] = self._fuse_row(0, self.fuse_layers[0], xs[0])
        if _DEFER_JOIN and not getattr(self, 'join_output', True) and not _BRANCH_STREAM_MAP \
                and len(self.fuse_layers) == self.num_branches:
            # the next module of the stage runs branch i on the stream that produced fused[i]: no join here and no fork
            # there -- its branches start as soon as THEIR row is done instead of after the slowest one (the rows cost
            # between three up-sampling + add kernels and six stride-2 convolution + norm pairs)
            for i in range(1, len(self.fuse_layers)):
                fused[i]._dcl_stream = side[i - 1]
            return fused
        for i in range(1, len(self.fuse_layers)):
            main.wait_stream(side[i - 1])
            _amax_record_stream(fused[i], main)
        return fused

    def _fuse_row(self, i, row, x):
        """relu(sum_j f_ij(x_j)) (reference HRNet.py:270-285).  The identity term x_i opens the sum (so it costs no add
        kernel) and every other contribution adds into the running sum inside the kernel that produces it: the
        up-sampling kernel for coarser branches, the last norm of the stride-2 chain for finer ones; the closing ReLU
        rides on the last contribution.  (Summation order differs from the reference's j = 0, 1, ... by fp32
        round-off only.)"""
        if not x[0].is_cuda:                   # CPU: the reference's own order, op for op (tight parity tests)
            y = x[0] if i == 0 else row[0](x[0])
            for j in range(1, self.num_branches):
                if j == i:
                    y = y + x[j]
                elif j > i:
                    y = upsample_bilinear(row[j](x[j]), x[i].shape[-2:], self.align_corners, add=y)
                else:
                    chain = list(row[j])
                    t = x[j]
                    for k, step in enumerate(chain[:-1]):
                        t = step(t, consumer=chain[k + 1][0])
                    y = chain[-1](t, residual=y)
            return self.relu(y)
        # coarser branches (up-sampling kernels) first, finer ones (stride-2 chains) last: the closing ReLU then rides on a
        # norm kernel whenever the row has one -- its backward reads the packed sign mask, where an up-sampling with a
        # fused ReLU needs a threshold pass over the full-resolution gradient (rows 1.. of every module: 14 passes less)
        others = [j for j in range(self.num_branches) if j > i] + [j for j in range(self.num_branches) if j < i]
        y = x[i]
        for pos, j in enumerate(others):
            last = pos == len(others) - 1
            if j > i:
                y = upsample_bilinear(row[j](x[j]), x[i].shape[-2:], self.align_corners, add=y, relu=last)
            else:
                chain = list(row[j])            # stride-2 conv chain; its last norm absorbs "+ y" (and the ReLU)
                t = x[j]
                for k, step in enumerate(chain[:-1]):
                    t = step(t, consumer=chain[k + 1][0])       # (its norm + ReLU inside the next convolution when that one takes it)
                y = chain[-1](t, residual=y, relu=last)
        return y if others else self.relu(y)


class HighResolutionNet(nn.Module):
    def __init__(self, cfg, norm_layer=None, mixing_layer=False, use_as_backbone=False,
                 return_all_scales=False, align_corners=False, dataset='CITYSCAPES', experiment=1):
        super().__init__()
        self.norm_layer = norm_layer or nn.BatchNorm2d
        self.use_mxing_layer = mixing_layer
        self.dataset = dataset
        self.experiment = experiment
        self.use_as_backbone = use_as_backbone
        self.return_all_scales = return_all_scales
        self.align_corners = align_corners
        self.out_stride = 4
        self.projector_model = None
        if use_as_backbone and mixing_layer:
            self.num_classes = 0
        else:
            names = DATASETS_INFO[dataset].CLASS_INFO[experiment][1]
            self.num_classes = len(names) - 1 if 255 in names.keys() else len(names)

        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = self.norm_layer(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = self.norm_layer(64)
        self.relu = nn.ReLU(inplace=True)

        self.stage1_cfg = cfg['STAGE1']
        block = blocks_dict[self.stage1_cfg['BLOCK']]
        self.layer1 = _residual_chain(block, 64, self.stage1_cfg['NUM_CHANNELS'][0],
                                      self.stage1_cfg['NUM_BLOCKS'][0], self.norm_layer)
        pre = [block.expansion * self.stage1_cfg['NUM_CHANNELS'][0]]

        self.stage2_cfg = cfg['STAGE2']
        self.stage3_cfg = cfg['STAGE3']
        self.stage4_cfg = cfg['STAGE4']
        for idx, scfg in ((2, self.stage2_cfg), (3, self.stage3_cfg), (4, self.stage4_cfg)):
            blk = blocks_dict[scfg['BLOCK']]
            cur = [c * blk.expansion for c in scfg['NUM_CHANNELS']]
            setattr(self, f'transition{idx - 1}', self._transition(pre, cur))
            stage, pre = self._make_stage(scfg, cur)
            setattr(self, f'stage{idx}', stage)
        # stream continuity across the stage transitions as well (forward() / _enter_stage): the last module of stages
        # 2 and 3 leaves its outputs on their streams (marked), the transition and the next stage's first module pick them up
        for idx in (2, 3):
            getattr(self, f'stage{idx}')[-1].join_output = not _STAGE_CONTINUITY

    def _transition(self, pre, cur):
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                layers.append(_conv_bn(pre[i], c, 3, relu=True, norm=self.norm_layer) if c != pre[i] else None)
            else:   # new, coarser branch: stride-2 3x3 convs from the coarsest existing branch
                n_new = i + 1 - len(pre)
                steps = [_conv_bn(pre[-1], c if s == n_new - 1 else pre[-1], 3, 2, relu=True,
                                  norm=self.norm_layer) for s in range(n_new)]
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def _make_stage(self, scfg, num_inchannels, multi_scale_output=True):
        block = blocks_dict[scfg['BLOCK']]
        modules = []
        for m in range(scfg['NUM_MODULES']):
            ms_out = multi_scale_output or m < scfg['NUM_MODULES'] - 1
            modules.append(HighResolutionModule(scfg['NUM_BRANCHES'], block, scfg['NUM_BLOCKS'],
                                                num_inchannels, scfg['NUM_CHANNELS'], scfg['FUSE_METHOD'],
                                                ms_out, norm_layer=self.norm_layer))
            num_inchannels = modules[-1].get_num_inchannels()
        # between two modules of a stage output i stays on stream i (see HighResolutionModule.forward)
        for m, mod in enumerate(modules):
            mod.join_output = m == len(modules) - 1
        return nn.Sequential(*modules), num_inchannels

    @staticmethod
    def _enter_stage(transition, prev, n_prev):
        on_streams = (_DEFER_JOIN and _BRANCH_STREAMS and not _BRANCH_STREAM_MAP and n_prev > 1 and prev[0].is_cuda
                      and all(getattr(t, '_dcl_stream', None) is not None for t in prev[1:]))
        if not on_streams:
            out = []
            for i, t in enumerate(transition):
                src = prev[i] if i < n_prev else prev[-1]
                out.append(src if t is None else t(src))
            return out
        # prev[i] was left on stream i by the previous stage's last module (branch 0: the current stream): a transition
        # layer of an existing branch runs there, a new branch on the next free stream behind the branch it derives from;
        # the first module of the next stage continues on the same streams and joins after its branches
        main = torch.cuda.current_stream(prev[0].device)
        side = _side_streams(prev[0].device, len(transition) - 1)
        out = []
        for i, t in enumerate(transition):
            s = main if i == 0 else side[i - 1]
            if i < n_prev:
                if t is None:
                    out.append(prev[i])
                else:
                    with torch.cuda.stream(s):
                        out.append(t(prev[i]))
            else:
                src_stream = main if n_prev == 1 else side[n_prev - 2]
                s.wait_stream(src_stream)
                with torch.cuda.stream(s):
                    _amax_record_stream(prev[-1], s)
                    out.append(t(prev[-1]))
            if i > 0:
                out[-1]._dcl_stream = s
        return out

    def forward(self, x):
        x = self.conv1(x)
        # the stem's first norm + ReLU (403 MB at 12 x 512 x 1024) inside conv2's operand staging when conv2 takes it
        x = bn_act(self.bn1, x, defer=_defers(self.bn1, x, self.conv2))
        x = bn_act(self.bn2, self.conv2(x))
        x = self.layer1(x)
        y = self.stage2(self._enter_stage(self.transition1, [x], 1))
        y = self.stage3(self._enter_stage(self.transition2, y, self.stage2_cfg['NUM_BRANCHES']))
        y = self.stage4(self._enter_stage(self.transition3, y, self.stage3_cfg['NUM_BRANCHES']))
        assert self.use_as_backbone
        # (lazy_concat: the head convolution consumes the four maps themselves, ops.conv3x3_over_upsampled)
        cat = LazyConcat(list(y), self.align_corners) if getattr(self, 'lazy_concat', False) and y[0].is_cuda \
            else upsample_concat(list(y), self.align_corners)
        if self.return_all_scales:
            return cat, [y[0], y[1], y[2], y[3]]
        return cat


def _hrnet(arch, pretrained, progress, **kwargs):
    model = HighResolutionNet(MODEL_CONFIGS[arch], **kwargs)
    if pretrained:
        import os
        path = os.environ.get('HRNET_PRETRAINED', 'hrnetv2_w48_imagenet_pretrained.pth')
        if not os.path.isfile(path):
            raise FileNotFoundError(f'pretrained HRNet weights not found at {path} (no network on this box; '
                                    'set HRNET_PRETRAINED or use pretrained=False)')
        state = torch.load(path, map_location='cpu')
        missing = model.load_state_dict(state, strict=False)
        printlog(f'loaded pretrained {arch} from {path}; missing keys: {len(missing.missing_keys)}')
    return model


def hrnet18(pretrained=False, progress=True, **kwargs):
    return _hrnet('hrnet18', pretrained, progress, **kwargs)


def hrnet32(pretrained=False, progress=True, **kwargs):
    return _hrnet('hrnet32', pretrained, progress, **kwargs)


def hrnet48(pretrained=False, progress=True, **kwargs):
    return _hrnet('hrnet48', pretrained, progress, **kwargs)


_FACTORIES = {'hrnet18': hrnet18, 'hrnet32': hrnet32, 'hrnet48': hrnet48}


class HRNet(nn.Module):
    eligible_backbones = ['hrnet48', 'hrnet32', 'hrnet18']

    def __init__(self, config, experiment):
        super().__init__()
        self.config = config
        # the reference ignores config['backbone'] and always builds W48 (HRNet.py:567, 590);
        # honouring 'hrnet18'/'hrnet32' is an extension needed by BASELINE config 1
        name = config.get('backbone', 'hrnet48')
        self.backbone_name = name if name in _FACTORIES else 'hrnet48'
        self.out_stride = 4
        self.dataset = config['dataset']
        # fused BN(+add)(+ReLU) kernels (models/fused_bn.py); same parameters / state_dict as nn.BatchNorm2d
        self.norm = FusedBatchNorm2d if config.get('fused_bn', True) else nn.BatchNorm2d
        names = DATASETS_INFO[self.dataset].CLASS_INFO[experiment][1]
        self.num_classes = len(names) - 1 if 255 in names.keys() else len(names)
        self.align_corners = config['align_corners'] if 'align_corners' in config else True
        self.use_ms_projector = False
        self.projector_before_context = None
        self.backbone_cutoff = {'layer4': 'C5'}
        self.return_backbone_feats = False
        return_all_scales = 'ms_projector' in config
        if 'return_all_scales' in config:
            return_all_scales = config['return_all_scales']
            self.return_backbone_feats = True
            self.return_features = True

        self.backbone = _FACTORIES[self.backbone_name](
            self.config['pretrained'], mixing_layer=True, use_as_backbone=True,
            return_all_scales=return_all_scales, align_corners=self.align_corners,
            dataset=self.dataset, experiment=experiment, norm_layer=self.norm)
        self.backbone_out_channels = sum(self.backbone.stage4_cfg.NUM_CHANNELS)
        c = self.backbone_out_channels
        self.cls_head = nn.Sequential(
            nn.Conv2d(c, c, kernel_size=3, stride=1, padding=1),
            self.norm(c),
            nn.Conv2d(c, self.num_classes, kernel_size=1, stride=1, padding=0, bias=False))

        if 'projector' in config:
            self.return_features = True
            self.config['projector']['c_in'] = c
            self.projector_model = Projector(config=self.config['projector'])
        elif 'ms_projector' in config:
            self.return_features = True
            self.use_ms_projector = True
            mp = self.config['ms_projector']
            self.ms_projector_scales = mp['scales'] if 'scales' in mp else 4
            assert self.ms_projector_scales in [2, 3, 4], \
                f'HRNet scales must be in [2,3,4] instead got {self.ms_projector_scales}'
            mp['c_in'] = self.backbone.stage4_cfg.NUM_CHANNELS[:self.ms_projector_scales]
            self.projector_model = Projector(config=mp)
        else:
            self.projector_model = None
            self.return_features = False
        if 'return_all_scales' in config:
            self.return_features = True
        # 'direct' (default): the head's 720 -> 720 convolution on the same direct split-f16 kernels as the backbone
        # (32 ms for its three directions at batch 12); 'library': MIOpen
        self.head_conv = config.get('head_conv', 'direct')
        # head_split (default on with the direct head convolution): the 3x3 head convolution takes the four branch maps
        # instead of their up-sampled concatenation and moves the channel products of the two coarsest ones to low
        # resolution (models/ops.py conv3x3_over_upsampled); False = convolve the materialised 720-channel concatenation
        self.head_split = bool(config.get('head_split', True)) and self.head_conv == 'direct'
        self.backbone.lazy_concat = self.head_split and not self.return_backbone_feats and 'projector' not in config
        # 'f16x3': the backbone's 3x3 / stride-1 convolutions (BasicBlock, Bottleneck, transitions: ~80 % of the
        # FLOPs) on the direct split-f16 kernel (csrc/dcl_conv3x3.hip, fp32-equivalent); 'library': MIOpen
        self.branch_conv = config.get('branch_conv', 'f16x3')
        self._conv_packs = None
        if self.head_conv == 'direct':
            use_direct_conv3x3(self.cls_head)
        if self.branch_conv == 'f16x3':
            use_direct_conv3x3(self.backbone)
        # 1x1 convolutions (bottlenecks, fuse layers, projector, classifier): 'f16x3' (default) = the same direct
        # split-f16 kernels in their one-tap mode, all three directions; 'gemm' = plain batched fp32 library GEMMs (the
        # library's own weight gradient for a 1x1 convolution wraps an NHWC kernel in layout transposes); 'library'
        if self.projector_model is not None:
            # extension, default off: the heads' last 1x1 convolution is evaluated by the loss on the sampled pixels only
            self.projector_model.lazy = bool(config.get('lazy_projector', False))
            # (training on the GPU) embedding maps written pixel-major, handed out with channels-last strides: models/ops.py _Conv1x1ToNHWC
            self.projector_model.nhwc = bool(config.get('nhwc_projector', True))
        if config.get('fused_bn', True) and self.projector_model is not None:
            # the projector's norms as well (class switch: same parameters / buffers / state_dict keys); their outputs
            # and gradients then carry the absmax tags the projector's f16x3 1x1 convolutions need
            for m in self.projector_model.modules():
                if type(m) is nn.BatchNorm2d:
                    m.__class__ = FusedBatchNorm2d
        self.conv1x1 = config.get('conv1x1', 'f16x3' if config.get('gemm_conv1x1', True) else 'library')
        if self.conv1x1 == 'f16x3':
            use_direct_conv1x1(self)
        elif self.conv1x1 == 'gemm':
            use_gemm_conv1x1(self)
        if self.branch_conv == 'f16x3' or self.head_conv == 'direct' or self.conv1x1 == 'f16x3':
            self._conv_packs = ConvPackGroup(self)

    def _head(self, x):
        if isinstance(x, LazyConcat):
            conv = self.cls_head[0]
            t0 = x.ts[0]
            if (self.head_split and isinstance(conv, DirectConv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                    and t0.is_cuda and t0.dtype == torch.float32 and not torch.is_autocast_enabled()
                    and all(t.is_contiguous() for t in x.ts)):
                y = conv3x3_over_upsampled(x.ts, x.align_corners, conv.weight, conv.bias)
                return self._head_tail(y)
            x = x.materialize()
        return self._head_tail(self.cls_head[0](x))

    def _head_tail(self, z):
        """norm -> classifier behind the head convolution (reference models/HRNet.py:596-600: no activation in between): folded
        into one GEMM on z with rescaled weights when the norm is on its fused training path (models/ops_head.py)."""
        bn, cls = self.cls_head[1], self.cls_head[2]
        if len(self.cls_head) == 3 and head_norm_classifier_ok(z, bn, cls):
            return head_norm_classifier(z, bn, cls)
        return cls(bn(z))

    def forward(self, x):
        size = x.shape[-2:]
        if self._conv_packs is not None and x.is_cuda:
            self._conv_packs.refresh()          # all 3x3 weights re-packed by two launches per optimizer step
        feats = self.backbone(x)
        multi = self.use_ms_projector or self.return_backbone_feats
        logits = self._head(feats[0] if multi else feats)
        if self.config.get('lazy_logits', False) and self.training:
            # extension (default off = the reference's return value): the logits stay at 1/4 resolution; this repo's
            # LossWrapper / metrics apply up-sampling + cross-entropy / arg-max in fused kernels (models/ops.py)
            from .ops import UpsampledLogits
            logits = UpsampledLogits(logits, size, self.align_corners)
        else:
            logits = upsample_bilinear(logits, size, self.align_corners)
        if self.projector_model is not None:
            if self.use_ms_projector:
                proj = self.projector_model(feats[1][:self.ms_projector_scales])
            else:
                proj = self.projector_model(feats)
            return (logits, proj) if self.return_features else logits
        return (logits, feats) if self.return_features else logits
