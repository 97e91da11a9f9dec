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
    fuse_bn_apply: bool = field(default_factory=lambda: _flag('DCL_FUSE_BN_APPLY'))             # bn1 + ReLU of a residual block inside
    # conv2's operand staging (models/fused_bn.py PreAct, csrc/dcl_conv3x3_pre.hip); 0 = the norm writes its output (A/B runs)
    branch_streams: bool = field(default_factory=lambda: _flag('DCL_BRANCH_STREAMS'))           # one HIP stream per branch
    defer_join: bool = field(default_factory=lambda: _flag('DCL_DEFER_JOIN'))                   # no join between modules
    merge_from: int = field(default_factory=lambda: 1 if _int('DCL_MERGE_FROM') is None else _int('DCL_MERGE_FROM'))   # first merged branch
    merge_branches: bool = field(default_factory=lambda: _flag('DCL_MERGE_BRANCHES', False))    # branches 1.. of an exchange module: one
    # launch per kernel stage and block depth (models/merged.py, csrc k_conv3x3_il_multi / k_bn_*_multi).  OFF by default: built,
    # bitwise-tested (tests/test_merged_branches.py) and measured -- the three coarse convolutions of a stage-4 depth take 130 us as
    # one launch against 213 us in a row (tools/probes/conv_multi_time.py), the serialised kernel time of a step drops by 6.9 ms and
    # its launches from 2 813 to ~1 900, and the step gets SLOWER, 89.5 / 89.7 against 87.6 ms (alternating runs on one box,
    # profiles/r05_ab_merge_branches.json): with one stream per branch those partial-chip kernels already run beside each other and
    # beside branch 0, and the merged schedule puts the three coarse weight gradients (each a full-chip launch) of a depth in a row on
    # ONE stream -- its dependency chain per module is 2.3x the longest per-branch chain
    stage_continuity: bool = field(default_factory=lambda: _flag('DCL_STAGE_CONTINUITY'))       # ... nor between stages
    fanout_on_branch_stream: bool = field(default_factory=lambda: _flag('DCL_FANOUT_STREAM'))
    branch_stream_map: List[int] = field(default_factory=lambda: [
        int(v) for v in os.environ.get('DCL_BRANCH_STREAM_MAP', '').split(',') if v != ''])     # e.g. "0,1,1,0"
    fuse_order: bool = field(default_factory=lambda: _flag('DCL_FUSE_ORDER'))                   # stride-2 chains last
    fanout: bool = field(default_factory=lambda: _flag('DCL_FANOUT'))                           # one-kernel gradient sums
    upsample_tag: bool = field(default_factory=lambda: _flag('DCL_UPSAMPLE_TAG'))               # absmax tags of up-sampled maps
    small_cin_stem: bool = field(default_factory=lambda: _flag('DCL_SMALL_CIN_STEM'))            # fp32 kernel for the 3-channel stem conv
    deterministic_stem_wgrad: bool = field(default_factory=lambda: _flag('DCL_DETERMINISTIC_STEM_WGRAD'))   # the stem's weight gradient
    # on the split-f16 kernel (input channels zero-padded to 16) instead of the library's atomic split-K kernel: the training step is
    # bitwise reproducible with it (tests/test_step_reproducible.py)
    packed_relu_mask: bool = field(default_factory=lambda: _flag('DCL_BN_MASK'))                # packed sign mask in the BN backward
    gemm_conv1x1: bool = field(default_factory=lambda: _flag('DCL_GEMM_CONV1X1'))               # wide 1x1 convolutions on dcl_gemm_f16x3
    lib_conv1x1_addend: bool = field(default_factory=lambda: _flag('DCL_LIB_CONV1X1_ADDEND'))   # big 1x1 data gradients: residual
    # gradient accumulated by the library GEMM (beta = 1)
    gemm_conv1x1_addend: bool = field(default_factory=lambda: _flag('DCL_GEMM_CONV1X1_ADDEND', False))  # ... with the residual
    # gradient as C += -- OFF: 467 us per launch on layer 1's 256-channel gradients against 372 for the tile kernel's fused addend
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
    side_stream_priority: Optional[int] = field(default_factory=lambda: _int('DCL_SIDE_PRIO'))  # HIP priority of the branch streams
    # ---- loss
    mfma_mode: Optional[str] = field(default_factory=lambda: os.environ.get('DCL_MFMA'))        # 'f32' | 'f16x3' override
    keep_positives: bool = field(default_factory=lambda: _flag('DCL_KEEP_POSITIVES'))           # forward: one sweep + k_pos_finish
    sweep_streamk: Optional[int] = field(default_factory=lambda: _int('DCL_SWEEP_STREAMK'))     # 0 = column-split slabs
    sweep_slices: Optional[int] = field(default_factory=lambda: _int('DCL_SWEEP_SLICES'))       # 1 | 4 | 8 column slices of stream-K
    lib_path: Optional[str] = field(default_factory=lambda: os.environ.get('DCL_LIB_PATH'))     # probe builds of the library
    # ---- kernel variants set on the library at load (include/dcl_hip.h "tuning hook" entries)
    wgrad_variant: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_VARIANT'))
    wgrad_stride2: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_S2'))
    wgrad_wg_target: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_TARGET'))    # workgroups a weight-gradient launch aims at
    wgrad_wave_mode: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_WAVE'))      # 0 = a workgroup per tile pair
    wgrad_strip_group: Optional[int] = field(default_factory=lambda: _int('DCL_WGRAD_GROUP'))   # 0 = four row ranges of one strip
    up2_phases: Optional[int] = field(default_factory=lambda: _int('DCL_UP2_PHASES'))
    tapup_bwd_form: Optional[int] = field(default_factory=lambda: _int('DCL_TAPUP_BWD'))         # 1 = first form of the tap-up backward
    conv_interleave: Optional[int] = field(default_factory=lambda: _int('DCL_CONV_IL'))
    conv_min_workgroups: Optional[int] = field(default_factory=lambda: _int('DCL_CONV_MIN_WGS'))
    upce_bwd_chunk: Optional[int] = field(default_factory=lambda: _int('DCL_UPCE_BWD_CHUNK'))
    upce_fwd_kib: Optional[int] = field(default_factory=lambda: _int('DCL_UPCE_FWD_KIB'))
    wgrad_tile: Optional[Tuple[int, int]] = field(default_factory=lambda: (
        tuple(int(v) for v in os.environ['DCL_WGRAD_TILE'].split(',')) if os.environ.get('DCL_WGRAD_TILE') else None))

    def apply_to_library(self, l) -> None:
        """Hand the kernel-variant switches to a freshly loaded libdcl_hip.so."""
        for val, fn in ((self.wgrad_variant, l.dcl_wgrad3x3_set_variant), (self.wgrad_stride2, l.dcl_wgrad3x3_set_stride2),
                        (self.wgrad_wave_mode, l.dcl_wgrad3x3_set_wave_mode),
                        (self.wgrad_wg_target, l.dcl_wgrad3x3_set_workgroup_target),
                        (self.wgrad_strip_group, l.dcl_wgrad3x3_set_strip_group),
                        (self.up2_phases, l.dcl_conv3x3_set_up2_phases),
                        (self.tapup_bwd_form, l.dcl_tapup_set_bwd_form),
                        (self.conv_interleave, l.dcl_conv3x3_set_interleave),
                        (self.conv_min_workgroups, l.dcl_conv3x3_set_min_workgroups),
                        (self.gemm_gemm_tile, l.dcl_gemm_set_tile),
                        (self.upce_bwd_chunk, l.dcl_upsample_ce_set_bwd_chunk),
                        (self.upce_fwd_kib, l.dcl_upsample_ce_set_fwd_lds)):
            if val is not None:
                fn(int(val))
        if self.wgrad_tile:
            l.dcl_wgrad3x3_set_tile(*self.wgrad_tile)
        if self.sweep_slices is not None:
            l.dcl_infonce_set_streamk_slices(int(self.sweep_slices))
        if self.sweep_streamk is not None and hasattr(l, 'dcl_infonce_set_streamk'):
            l.dcl_infonce_set_streamk(int(self.sweep_streamk))


cfg = DebugConfig()
