"""Import shim for the *reference* checkout (this container only).

The reference (``/root/reference``) is pure Python but imports several third
party modules at package-import time that are absent from this image
(cv2, torchvision, timm, tensorboard, tsne_torch, h5py, easydict) and calls
``Tensor.cuda()`` unconditionally.  This shim registers inert stand-in modules
and makes ``.cuda()`` the identity so that the reference's loss / model code
can be *executed on CPU* to produce golden vectors.

It is test tooling: nothing under ``tools/`` is imported by the product, and
nothing here travels to the GPU box in a form that needs ``/root/reference``.
"""
import importlib.machinery
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


class _Dummy:
    """Stands in for any class/function pulled from a stubbed module."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


def _stub(name):
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
    mod.__path__ = []

    def _getattr(attr, _name=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Dummy

    mod.__getattr__ = _getattr
    sys.modules[name] = mod
    return mod


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _DropPath(torch.nn.Module):
    """timm.models.layers.DropPath semantics (per-sample stochastic depth)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep


def install():
    if getattr(install, "_done", False):
        return
    for name in [
        "cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
        "torchvision.models", "torchvision.models._utils", "torchvision.datasets",
        "torchvision.datasets.cityscapes",
        "timm", "timm.models", "tsne_torch", "tensorboard", "h5py",
        "torch.utils.tensorboard", "torch.utils.tensorboard.writer",
    ]:
        if name not in sys.modules:
            _stub(name)
    layers = _stub("timm.models.layers")
    layers.DropPath = _DropPath
    layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    ed = _stub("easydict")
    ed.EasyDict = _EasyDict
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True


def quiet():
    """Silence the reference's printlog/Logger chatter."""
    import logging
    logging.disable(logging.CRITICAL)
