"""Absmax side-channel for the f16x3 convolution (csrc/dcl_conv3x3.hip).

The direct convolution scales each operand by a power of two derived ON THE DEVICE from the operand's absmax.
Producers that already stream the tensor (the fused BN kernels, forward and backward) emit 64 partial maxima into
a small zero-initialised buffer and tag their output with it; a consumer finds the tag through :func:`amax_of`
and otherwise falls back to one `dcl_absmax` pass.  A tag carries the tensor's ``_version`` and is ignored once
the tensor was modified in place (e.g. autograd's in-place gradient accumulation)."""
import ctypes

import torch

from .. import _lib

SLOTS = 64        # DCL_AMAX_SLOTS (include/dcl_hip.h): partial maxima a fused BN kernel emits

_POOL = {}
_POOL_FLOATS = 1 << 20


def zeros(n: int, device) -> torch.Tensor:
    """n zero floats from a pooled buffer (one fill kernel per 1 M floats instead of one per request);
    slices are never handed out twice, the buffer lives as long as any slice does."""
    n_al = (n + 3) & ~3
    # one pool per stream: the fill kernel of a pool buffer is ordered only against the stream that created it
    key = (device.type, device.index, _lib.stream_ptr(device) if device.type == "cuda" else 0)
    buf, off = _POOL.get(key, (None, 0))
    if buf is None or off + n_al > buf.numel():
        buf, off = torch.zeros(max(_POOL_FLOATS, n_al), dtype=torch.float32, device=device), 0
    _POOL[key] = (buf, off + n_al)
    return buf[off:off + n]


def tag(t: torch.Tensor, buf: torch.Tensor) -> torch.Tensor:
    t._dcl_amax = (t._version, buf)
    return t


def carry(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst (a view / reshape / contiguous copy of src: same values) inherits src's valid tag; returns dst."""
    if dst is not src:
        got = getattr(src, "_dcl_amax", None)
        if got is not None and got[0] == src._version and got[1].device == dst.device:
            dst._dcl_amax = (dst._version, got[1])
    return dst


def tag_of(t: torch.Tensor):
    """The valid absmax tag of t, or None (no fallback pass)."""
    got = getattr(t, "_dcl_amax", None)
    if got is not None and got[0] == t._version and got[1].device == t.device:
        return got[1]
    return None


def amax_of(t: torch.Tensor) -> torch.Tensor:
    """1-D float tensor whose maximum is max|t| (per-plane maxima from the producer, or one value)."""
    got = getattr(t, "_dcl_amax", None)
    if got is not None and got[0] == t._version and got[1].device == t.device:
        return got[1]
    buf = zeros(1, t.device)
    st = _lib.stream_ptr(t.device)
    _lib.check(_lib.lib().dcl_absmax(_lib.ptr(t), t.numel(), _lib.ptr(buf), st), "dcl_absmax")
    tag(t, buf)
    return buf


class PreAct:
    """Marks a tensor whose STORAGE holds the raw input z of a training-mode norm while its VALUE -- for the one consumer that
    understands the mark, ``DirectConv2d`` -- is ``relu(z * sc[c] + sh[c])``: the normalised tensor of conv1 -> bn1 -> relu ->
    conv2 (reference models/HRNet.py:77-93) is never written; conv2's forward and weight-gradient kernels apply the map while
    they stage their operand (csrc/dcl_conv3x3_pre.hip, k_wgrad3x3d PRE).  Only ``FusedBatchNorm2d(..., defer=True)`` creates one,
    and only after the consumer said it takes it (``DirectConv2d.fuses_input_norm``); anything else that is handed such a tensor
    must refuse it (``refuse_pre``).  ``amax``: the absmax slots of the mapped tensor."""
    __slots__ = ("sc", "sh", "amax", "version")

    def __init__(self, sc, sh, amax, version):
        self.sc, self.sh, self.amax, self.version = sc, sh, amax, version


def pre_of(t: torch.Tensor):
    """The valid PreAct mark of t, or None."""
    got = getattr(t, "_dcl_pre", None)
    if got is not None and got.version == t._version:
        return got
    return None


def refuse_pre(t: torch.Tensor, who: str):
    if getattr(t, "_dcl_pre", None) is not None:
        raise RuntimeError(f"{who}: the input is a deferred norm output (its storage holds the norm's INPUT); only a DirectConv2d "
                           "that answered fuses_input_norm() may consume it")


def record_stream(t: torch.Tensor, stream) -> torch.Tensor:
    """``t.record_stream(stream)`` for a tensor that is handed to another HIP stream, including the absmax buffer
    it is tagged with (separate storage, same lifetime hazard with the caching allocator)."""
    t.record_stream(stream)
    got = getattr(t, "_dcl_amax", None)
    if got is not None and got[1].is_cuda:
        got[1].record_stream(stream)
    return t
