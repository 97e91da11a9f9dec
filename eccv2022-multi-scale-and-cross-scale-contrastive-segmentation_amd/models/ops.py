"""Operator-level replacements inside the models where the library's default kernel is far from the hardware roofline on MI355X
(measured, see profiles/).  Since round 5 one module per operator family; this module re-exports all of them, so
``from .ops import X`` / ``ops.X`` keep working:

    ops_resize   bilinear up-sampling (+ add, + ReLU), concatenation without copies, one-kernel gradient sums (fan_out)
    ops_conv     direct f16x3 3x3 / 1x1 convolutions (forward, data gradient, weight gradient), DirectConv2d, weight packing
    ops_conv1x1  1x1 convolutions as batched split-f16 GEMMs (NCHW, token-major input, Dropout2d folded in, pixel-major output)
    ops_head     the head convolution over up-sampled maps without the up-sampled maps (tap products + tap gather)
    ops_linear   dcl_gemm_f16x3 and the token-major Linears (fused Mlp, residual epilogues, GELU)
    ops_swin     LayerNorm, window attention
    ops_logits   fused up-sampling + cross-entropy (UpsampledLogits)

Module switches (A/B runs, tests) live in the module that reads them (``ops_resize.HIP_UPSAMPLE``, ``ops_linear.FUSED_MLP``
...).  This namespace holds FUNCTIONS and CLASSES only; a switch is read through to its owner (``ops.FUSED_MLP`` is the live value) and
an assignment ``ops.FUSED_MLP = False`` is forwarded to the owning module, so the idiom of earlier tests and tools keeps working
instead of silently changing a dead copy."""
import sys
import types

from . import ops_common, ops_resize, ops_linear, ops_conv, ops_conv1x1, ops_head, ops_swin, ops_logits

_FAMILIES = (ops_common, ops_resize, ops_linear, ops_conv, ops_conv1x1, ops_head, ops_swin, ops_logits)


def _is_api(v):
    return isinstance(v, (types.FunctionType, type))


__all__ = []
for _m in _FAMILIES:
    for _k, _v in vars(_m).items():
        if _is_api(_v) and not _k.startswith("__") and getattr(_v, "__module__", "").startswith(__package__):
            globals()[_k] = _v
            if not _k.startswith("_"):
                __all__.append(_k)
del _m, _k, _v


def _owner(name):
    for m in _FAMILIES:
        if name in vars(m) and not _is_api(vars(m)[name]) and not isinstance(vars(m)[name], types.ModuleType):
            return m
    return None


def __getattr__(name):                      # switches: the owner's live value
    m = _owner(name)
    if m is None:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
    return getattr(m, name)


class _Facade(types.ModuleType):
    def __setattr__(self, name, value):
        m = _owner(name)
        if m is not None:
            setattr(m, name, value)         # a switch: set where it is read
        else:
            super().__setattr__(name, value)


sys.modules[__name__].__class__ = _Facade
