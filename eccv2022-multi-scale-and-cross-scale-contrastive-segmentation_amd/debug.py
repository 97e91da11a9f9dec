"""The ONE place where tuning / A-B switches enter the package.

Every switch of the tuning tools (tools/*.py, bench.py A/B runs on one box) is a field of ``cfg`` below, read from
the environment ONCE at import; product modules read ``debug.cfg.<field>`` and never touch ``os.environ``.  All
defaults are the shipped configuration; nothing here changes results beyond fp32 round-off (kernel variants, stream
layout), and none of it is part of the reference's surface."""
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple


def _flag(name: str, default: bool = True) -> bool:
    v = os.environ.get(name)
    return default if v is None else v != '0'


def _int(name: str) -> Optional[int]:
    v = os.environ.get(name)
    return None if v is None or v == '' else int(v)


@dataclass
class DebugConfig:
    # ---- model graph (models/HRNet.py, models/ops.py, models/fused_bn.py)
    fuse_residual_grad: bool = field(default_factory=lambda: _flag('DCL_FUSE_RESIDUAL_GRAD'))   # GradToken path
    fold_head_norm: bool = field(default_factory=lambda: _flag('DCL_FOLD_HEAD_NORM'))           # the head's norm folded into its
    # classifier's weights (models/ops_head.py _HeadNormClassifier); 0 = the norm writes its 720-channel output (A/B runs)
    fuse_bn_apply: bool = field(default_factory=lambda: _flag('DCL_FUSE_BN_APPLY'))             # bn1 + ReLU of a residual block inside
    # conv2's operand staging (models/fused_bn.py PreAct, csrc/dcl_conv3x3_pre.hip); 0 = the norm writes its output (A/B runs)
    branch_streams: bool = field(default_factory=lambda: _flag('DCL_BRANCH_STREAMS'))           # one HIP stream per branch
    defer_join: bool = field(default_factory=lambda: _flag('DCL_DEFER_JOIN'))                   # no join between modules
    stage_continuity: bool = field(default_factory=lambda: _flag('DCL_STAGE_CONTINUITY'))       # ... nor between stages
    fanout_on_branch_stream: bool = field(default_factory=lambda: _flag('DCL_FANOUT_STREAM'))
    branch_stream_map: List[int] = field(default_factory=lambda: [
        int(v) for v in os.environ.get('DCL_BRANCH_STREAM_MAP', '').split(',') if v != ''])     # e.g. "0,1,1,0"
    fanout: bool = field(default_factory=lambda: _flag('DCL_FANOUT'))                           # one-kernel gradient sums
    upsample_tag: bool = field(default_factory=lambda: _flag('DCL_UPSAMPLE_TAG'))               # absmax tags of up-sampled maps
    small_cin_stem: bool = field(default_factory=lambda: _flag('DCL_SMALL_CIN_STEM'))            # fp32 kernel for the 3-channel stem conv
    packed_relu_mask: bool = field(default_factory=lambda: _flag('DCL_BN_MASK'))                # packed sign mask in the BN backward
    gemm_conv1x1: bool = field(default_factory=lambda: _flag('DCL_GEMM_CONV1X1'))               # wide 1x1 convolutions on dcl_gemm_f16x3
    lib_conv1x1_addend: bool = field(default_factory=lambda: _flag('DCL_LIB_CONV1X1_ADDEND'))   # big 1x1 data gradients: residual
    # gradient accumulated by the library GEMM (beta = 1)
    head_dx_splitk: int = field(default_factory=lambda: 1 if _int('DCL_HEAD_DX_SPLITK') is None else _int('DCL_HEAD_DX_SPLITK'))  # k-splits of the
    # head's coarse data-gradient GEMM [C_b x 6480] . [6480 x P] (0 = the library's plan)
    head_taps_image_major: bool = field(default_factory=lambda: _flag('DCL_HEAD_TAPS_IMAGE_MAJOR'))   # tap products [N][9 Co][h w], batched GEMMs
    gemm_head_taps: bool = field(default_factory=lambda: _flag('DCL_GEMM_HEAD'))                # the head's tap products on it
    head_overlap: int = field(default_factory=lambda: 2 if _int('DCL_HEAD_OVERLAP') is None else _int('DCL_HEAD_OVERLAP'))  # coarse half of the head's
    # backward on a side stream: 0 off, 1 on, 2 on with the fine part's weight gradient first (A/B: 96.4 / 96.0 / 95.6 ms)
    gemm_gemm_tile: Optional[int] = field(default_factory=lambda: _int('DCL_GEMM_TILE'))       # 1..4: force a workgroup tile
    fold_dropout2d: bool = field(default_factory=lambda: _flag('DCL_FOLD_DROPOUT2D'))           # Dropout2d -> conv1x1 as per-sample weights
    token_laterals: bool = field(default_factory=lambda: _flag('DCL_TOKEN_LATERALS'))           # UPerNet laterals read Swin outputs token-major
    gemm_ascale: bool = field(default_factory=lambda: _flag('DCL_GEMM_ASCALE'))                 # DropPath's factor as an operand scale of the backward GEMMs
    relu_then_bn: bool = field(default_factory=lambda: _flag('DCL_RELU_THEN_BN'))               # projector ReLU's backward inside the norm's
    head_split_min_scale: int = field(default_factory=lambda: 0 if _int('DCL_HEAD_SPLIT_MIN_SCALE') is None else _int('DCL_HEAD_SPLIT_MIN_SCALE'))
    # 0 = automatic (models/ops.py conv3x3_over_upsampled: >= 4x coarser maps, and 2x coarser ones with >= 256 channels, go through the
    # tap products of the head split); 4 = round 3's rule; 2 = every map at least 2x coarser
    fused_mlp: bool = field(default_factory=lambda: _flag('DCL_FUSED_MLP'))                     # Swin Mlp / residual sums in GEMM epilogues
    coalesced_sync_bn: bool = field(default_factory=lambda: _flag('DCL_SYNCBN_COALESCE'))       # stacked SyncBN exchanges
    mfma_mode: Optional[str] = field(default_factory=lambda: os.environ.get('DCL_MFMA'))        # 'f32' | 'f16x3' override
    keep_positives: bool = field(default_factory=lambda: _flag('DCL_KEEP_POSITIVES'))           # forward: one sweep + k_pos_finish
    sweep_streamk: Optional[int] = field(default_factory=lambda: _int('DCL_SWEEP_STREAMK'))     # 0 = column-split slabs
    sweep_slices: Optional[int] = field(default_factory=lambda: _int('DCL_SWEEP_SLICES'))       # 1 | 4 | 8 column slices of stream-K
    lib_path: Optional[str] = field(default_factory=lambda: os.environ.get('DCL_LIB_PATH'))     # probe builds of the library
    # ---- kernel variants set on the library at load (include/dcl_hip.h "tuning hook" entries)
    wgrad_variant: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_VARIANT'))
    wgrad_wg_target: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_TARGET'))    # workgroups a weight-gradient launch aims at
    wgrad_wave_mode: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_WAVE'))      # 0 = a workgroup per tile pair
    wgrad_strip_group: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_GROUP'))   # 0 = four row ranges of one strip
    up2_phases: Optional[int] = field(default_factory=lambda: _int('DCL_UP2_PHASES'))
    conv_interleave: Optional[int] = field(default_factory=lambda: _int('DCL_CONV_IL'))
    conv_min_workgroups: Optional[int] = field(default_factory=lambda: _int('DCL_CONV_MIN_WGS'))
    wgrad_tile: Optional[Tuple[int, int]] = field(default_factory=lambda: (
        tuple(int(v) for v in os.environ['DCL_WGRAD_TILE'].split(',')) if os.environ.get('DCL_WGRAD_TILE') else None))

    def apply_to_library(self, l) -> None:
        """Hand the kernel-variant switches to a freshly loaded libdcl_hip.so."""
        for val, fn in ((self.wgrad_variant, l.dcl_wgrad3x3_set_variant),
                        (self.wgrad_wave_mode, l.dcl_wgrad3x3_set_wave_mode),
                        (self.wgrad_wg_target, l.dcl_wgrad3x3_set_workgroup_target),
                        (self.wgrad_strip_group, l.dcl_wgrad3x3_set_strip_group),
                        (self.up2_phases, l.dcl_conv3x3_set_up2_phases),
                        (self.conv_interleave, l.dcl_conv3x3_set_interleave),
                        (self.conv_min_workgroups, l.dcl_conv3x3_set_min_workgroups),
                        (self.gemm_gemm_tile, l.dcl_gemm_set_tile)):
            if val is not None:
                fn(int(val))
        if self.wgrad_tile:
            l.dcl_wgrad3x3_set_tile(*self.wgrad_tile)
        if self.sweep_slices is not None:
            l.dcl_infonce_set_streamk_slices(int(self.sweep_slices))
        if self.sweep_streamk is not None and hasattr(l, 'dcl_infonce_set_streamk'):
            l.dcl_infonce_set_streamk(int(self.sweep_streamk))


cfg = DebugConfig()
