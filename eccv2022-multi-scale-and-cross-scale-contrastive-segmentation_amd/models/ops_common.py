"""Shared helpers of the operator modules (models/ops_*.py)."""


def _stream(t):
    from .. import _lib
    return _lib.stream_ptr(t.device)
