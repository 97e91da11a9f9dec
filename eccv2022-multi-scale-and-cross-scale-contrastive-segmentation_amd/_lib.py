"""ctypes binding of libdcl_hip.so (C ABI: include/dcl_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libdcl_hip.so")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

CP = 256          # DCL_CP
ROW_TILE = 128    # DCL_ROW_TILE
SEG = 256         # DCL_SEG
MAX_CLASSES = 255
MAX_SLABS = 64

_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float

# name -> argtypes (all return int except where noted); mirrors include/dcl_hip.h one to one
SIGNATURES = {
    "dcl_label_hist": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "dcl_rank_select": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "dcl_gather_normalize": [_vp, _i64, _i64, _i64, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "dcl_gather_raw": [_vp, _i64, _i64, _i64, _i, _vp, _vp, _i, _i, _vp, _vp],
    "dcl_scatter_raw": [_vp, _i64, _i64, _i64, _i, _vp, _vp, _i, _i, _vp, _vp],
    "dcl_infonce_fwd": [_vp, _i, _i, _vp, _i, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "dcl_infonce_zsweep": [_vp, _i, _i, _vp, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp],
    "dcl_infonce_possweep": [_vp, _i, _i, _vp, _i, _vp, _vp, _f, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "dcl_infonce_loss": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "dcl_infonce_zsweep_keep": [_vp, _i, _i, _vp, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "dcl_infonce_pos_finish": [_vp, _i, _i, _i, _vp, _vp, _f, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "dcl_infonce_prep_stats": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp],
    "dcl_infonce_bwd": [_vp, _i, _i, _vp, _i, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "dcl_infonce_bwd_streamk_workgroups": [_i, _i],
    "dcl_infonce_bwd_streamk_slabs": [_i, _i],
    "dcl_infonce_set_streamk_slices": [_i],
    "dcl_infonce_set_streamk": [_i],
    "dcl_infonce_set_streamk_timeout_ms": [_i],
    "dcl_infonce_bwd_streamk": [_vp, _i, _i, _vp, _i, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dcl_normalize_bwd_scatter": [ctypes.POINTER(_vp), _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp,
                                  _i64, _i64, _i64, _vp, _vp],
    "dcl_host_randperm_select": [_vp, _i64, _vp, _i, _i, _vp],
    "dcl_bn_num_slices": [_i, _i],
    "dcl_bn_stats": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "dcl_bn_finalize": [_vp, _i, ctypes.c_double, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "dcl_bn_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "dcl_bn_stats_finalize": [_vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dcl_bn_stats_part": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "dcl_bn_apply_fused": [_vp, _vp, _vp, ctypes.c_double, _f, _f, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                           _vp, _vp, _vp],
    "dcl_bn_apply_parts": [_vp, _vp, _vp, _i, ctypes.c_double, _f, _f, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                           _vp, _vp, _vp],
    "dcl_bn_stats_minmax_part": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "dcl_bn_finalize_pre": [_vp, _vp, _i, ctypes.c_double, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dcl_bn_stats_pre": [_vp, _i, _i, _i, _vp, _vp, _vp, ctypes.c_double, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dcl_head_norm_dz": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "dcl_conv3x3_pre_supported": [_i, _i, _i, _i, _i, _i],
    "dcl_conv3x3_pre_f16x3": [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "dcl_wgrad3x3_pre_supported": [_i, _i, _i, _i, _i, _i],
    "dcl_wgrad3x3_pre_f16x3": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "dcl_conv3x3_s2_smallcin": [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "dcl_wgrad3x3_s2_smallcin_workspace": [_i],
    "dcl_wgrad3x3_s2_smallcin": [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "dcl_bn_bwd_reduce_part": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "dcl_bn_bwd_apply_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _i, _i, _i, _i, _vp, _vp, _vp,
                               _vp, _vp, _vp],
    "dcl_bn_bwd_reduce": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "dcl_bn_bwd_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "dcl_upsample_bilinear_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "dcl_upsample_bilinear_bwd": [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "dcl_amax_sum2": [_vp, _i, _vp, _i, _vp, _vp],
    "dcl_tapup_fwd": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "dcl_tapup_bwd": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "dcl_tapup_bwd_amax": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "dcl_tapup_set_bwd_form": [_i],
    "dcl_tapup_supported": [_i, _i, _i, _i, _i, _i, _i],
    "dcl_gemm_supported": [_i, _i, _i, _i64, _i, _i64, _i],
    "dcl_gemm_workspace_floats": [_i, _i, _i, _i],
    "dcl_gemm_suggest_splitk": [_i, _i, _i, _i],
    "dcl_gemm_set_tile": [_i],
    "dcl_gemm_f16x3": [_vp, _i64, _i, _i64, _vp, _i64, _i, _i64, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i64, _i64, _i,
                       _vp, _i, _vp, _vp, _vp],
    "dcl_gemm_f16x3_ascaled": [_vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i64, _vp, _i, _vp, _vp, _vp, _i, _i,
                               _vp, _vp],
    "dcl_gemm_f16x3_ep": [_vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _i, _vp, _vp, _vp, _i,
                          _vp],
    "dcl_add_n": [_vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "dcl_upsample_bilinear_fwd_slice": [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "dcl_upsample_bilinear_bwd_slice": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "dcl_absmax": [_vp, _i64, _vp, _vp],
    "dcl_conv3x3_pack": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "dcl_absmax_multi": [_vp, _vp, _i, _vp],
    "dcl_conv3x3_pack_multi": [_vp, _vp, _i, _vp],
    "dcl_conv3x3_f16x3": [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "dcl_conv1x1_f16x3": [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "dcl_wgrad1x1_splits": [_i, _i, _i, _i, _i],
    "dcl_wgrad1x1_f16x3": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "dcl_wgrad3x3_splits": [_i, _i, _i, _i, _i, _i],
    "dcl_wgrad3x3_set_stride2": [_i],
    "dcl_wgrad3x3_set_splits": [_i],
    "dcl_wgrad3x3_set_workgroup_target": [_i],
    "dcl_wgrad3x3_set_wave_mode": [_i],
    "dcl_wgrad3x3_set_strip_group": [_i],
    "dcl_wgrad3x3_set_wave_band": [_i],
    "dcl_conv3x3_set_up2_phases": [_i],
    "dcl_conv3x3_set_min_workgroups": [_i],
    "dcl_conv3x3_set_interleave": [_i],
    "dcl_upsample_ce_set_bwd_chunk": [_i],
    "dcl_upsample_ce_set_fwd_lds": [_i],
    "dcl_wgrad3x3_set_tile": [_i, _i],
    "dcl_wgrad3x3_set_variant": [_i],
    "dcl_wgrad3x3_f16x3": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp],
    "dcl_confusion_matrix": [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp],
    "dcl_metrics_from_cm": [_vp, _i, _i, _vp, _vp],
    "dcl_confusion_matrix_pred": [_vp, _i64, _vp, _i, _i, _i, _vp, _vp, _vp],
    "dcl_winattn_npad": [_i, _i],
    "dcl_winattn_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "dcl_winattn_bwd_waves": [_i, _i, _i, _i],
    "dcl_winattn_set_mfma": [_i],
    "dcl_winattn_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "dcl_layernorm_supported": [_i],
    "dcl_layernorm_bwd_parts": [_i64, _i],
    "dcl_layernorm_fwd": [_vp, _vp, _vp, _i64, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "dcl_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "dcl_upsample_ce_fwd": [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "dcl_upsample_ce_bwd": [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "dcl_suggest_nsplit": [_i, _i],
    "dcl_version": [],
    "dcl_trace_kernels": [_i],
}


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into libdcl_hip.so (in-tree)."""
    out = subprocess.run(["make", "-j8", "-C", CSRC_DIR], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
        print(out.stderr)
    if out.returncode != 0:
        raise RuntimeError("building libdcl_hip.so failed:\n" + out.stderr[-4000:])
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found. The dense contrastive loss has no fallback path: build the HIP "
                f"library first (python -c 'import __graft_entry__ as g; g.build()' or make -C {CSRC_DIR}).")
        from .debug import cfg as _dbg0
        l = ctypes.CDLL(_dbg0.lib_path or LIB_PATH)      # (lib_path: probe builds, tools/probes/conv_bounds.sh)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        l.dcl_last_error.restype = ctypes.c_char_p
        l.dcl_last_kernel.restype = ctypes.c_char_p
        l.dcl_last_kernel.argtypes = []
        l.dcl_gemm_workspace_floats.restype = ctypes.c_int64
        l.dcl_last_error.argtypes = []
        from .debug import cfg as _dbg       # A/B switches of the tuning tools: one object, read once
        _dbg.apply_to_library(l)
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().dcl_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor for a ``void *`` argument (a plain int: ctypes converts it, and
    building a c_void_p object per argument costs more than the call on this launch-bound path); None -> NULL."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr(device=None):
    """The current HIP stream of ``device`` (default: the current device) as an int for a ``void *stream`` argument;
    torch's raw-stream query is ~10x cheaper than building a torch.cuda.Stream object per launch."""
    import torch
    if device is None or device.index is None:
        idx = torch.cuda.current_device()
    else:
        idx = device.index
    return torch._C._cuda_getCurrentRawStream(idx)
