"""The drop-in boundary exercised FROM THE REFERENCE SIDE: the reference's own
``managers.BaseManager`` / ``managers.HRNet_Manager`` modules are imported with ``models`` and ``losses`` resolving
to ``mscs_amd.models`` / ``mscs_amd.losses`` (the import swap of BaseManager.py:19-20), then the REFERENCE's
``load_model`` / ``load_loss`` (BaseManager.py:437-501, ``globals()[name]`` lookups) build this repo's classes from
the reference's shipped JSON config, and the REFERENCE's ``HRNetManager.forward_step`` (HRNet_Manager.py:18-54)
drives them.

Runs in this build container only (the reference checkout is not on the GPU box) and in a subprocess (the import
shim patches ``torch.Tensor.cuda`` and puts the reference on ``sys.path``)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")

SCRIPT = r"""
import json, os, sys, types
sys.path.insert(0, os.path.join(%(root)r, "tools"))
sys.path.insert(0, %(root)r)
import ref_shim
ref_shim.install(); ref_shim.quiet()
import torch
import mscs_amd
import mscs_amd.models, mscs_amd.losses
# the import swap of INTEGRATION.md section A: `from models import *` / `from losses import *` now see this repo
def swap(name, pkg):
    # module `name` = the public names of this repo's package; names of model families outside SURVEY section 8
    # (DeepLabv3, OCRNet -- imported by managers the hot path never uses) resolve to an inert placeholder
    mod = types.ModuleType(name)
    pub = [k for k in vars(pkg) if not k.startswith("_")]
    for k in pub:
        setattr(mod, k, getattr(pkg, k))
    mod.__all__ = [k for k in pub if not isinstance(getattr(pkg, k), types.ModuleType)]
    def missing(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return type(attr, (), {"_placeholder": True})
    mod.__getattr__ = missing
    sys.modules[name] = mod
swap("models", mscs_amd.models)
swap("losses", mscs_amd.losses)
import builtins
_print = builtins.print
builtins.print = lambda *a, **k: None
from managers.BaseManager import BaseManager as RefBase            # the REFERENCE's modules
from managers.HRNet_Manager import HRNetManager as RefHRNetManager
import managers.BaseManager as refmod
assert refmod.__file__.startswith("/root/reference/"), refmod.__file__
out = {}
out["globals"] = {n: refmod.__dict__[n].__module__ for n in
                  ("HRNet", "UPerNet", "LossWrapper", "DenseContrastiveLossV2_ms", "DenseContrastiveLossV2",
                   "TwoScaleLoss")}

def bare(cls, cfg_path, mode="training", patch=None):
    cfg = json.load(open(cfg_path))
    if patch:
        patch(cfg)
    m = object.__new__(cls)                  # construct-only: no log directory, no dataset, no process group
    m.config = cfg
    m.config["mode"] = mode
    m.experiment = cfg["data"]["experiment"]
    m.dataset = cfg["data"]["dataset"]
    m.config["graph"]["dataset"] = m.dataset                      # LoggingManager.py:43
    m.config["loss"].update({"dataset": m.dataset, "experiment": m.experiment})   # LoggingManager.py:63-65
    m.device = torch.device("cpu")
    m.parallel = False
    m.rank = 0
    m.epoch = 0
    m.empty_cache = False
    return m

def shrink(cfg):
    cfg["graph"]["pretrained"] = False

# 1. the shipped HRNet contrastive config through the reference's load_model / load_loss
m = bare(RefHRNetManager, "/root/reference/configs/CITYSCAPES/hrnet_contrastive_CTS.json", patch=shrink)
RefBase.load_model(m)
RefBase.load_loss(m)
out["hrnet_model"] = type(m.model).__module__ + "." + type(m.model).__name__
out["hrnet_loss"] = type(m.loss).__module__ + "." + type(m.loss).__name__
out["hrnet_terms"] = {k: type(v).__module__ for k, v in m.loss.loss_classes.items()}
out["hrnet_weights"] = dict(m.loss.loss_weightings)
out["hrnet_return_features"] = bool(m.return_features)
out["hrnet_params"] = sum(p.numel() for p in m.model.parameters())
dc = m.loss.loss_classes["DenseContrastiveLossV2_ms"]
out["hrnet_dc"] = {"scales": dc.scales, "weights": list(dc.weights), "cross": bool(dc.cross_scale_contrast),
                   "K": dc.DCV2_scale0.num_all_classes}

# 2. the shipped UPerNet + Swin-T config
def shrink_u(cfg):
    cfg["graph"]["pretrained"] = False
m2 = bare(RefHRNetManager, "/root/reference/configs/ADE20K/upnswin_contrastive_ADE20K.json", patch=shrink_u)
RefBase.load_model(m2)
RefBase.load_loss(m2)
out["upn_model"] = type(m2.model).__module__ + "." + type(m2.model).__name__
out["upn_terms"] = sorted(m2.loss.loss_classes)
out["upn_params"] = sum(p.numel() for p in m2.model.parameters())

# 3. the reference's forward_step driving this repo's model + LossWrapper (CE only: the contrastive loss itself
#    has no CPU path by design -- it raises, which is checked too)
def small(cfg):
    cfg["graph"]["pretrained"] = False
    cfg["graph"]["backbone"] = "hrnet18"
    cfg["loss"]["losses"] = {"CrossEntropyLoss": 1}
m3 = bare(RefHRNetManager, "/root/reference/configs/CITYSCAPES/hrnet_contrastive_CTS.json", patch=small)
RefBase.load_model(m3)
RefBase.load_loss(m3)
m3.model.train()
g = torch.Generator().manual_seed(0)
img = torch.randn(2, 3, 64, 64, generator=g)
lbl = torch.randint(0, 20, (2, 64, 64), generator=g)
ret = RefHRNetManager.forward_step(m3, img, lbl)
ret["loss"].backward()
out["step_keys"] = sorted(ret)
out["step_loss"] = float(ret["loss"])
out["step_output"] = list(ret["output"].shape)
out["step_feats"] = [list(f.shape) for f in ret["feats"]]
out["step_loss_vals"] = sorted(m3.loss.loss_vals)
out["step_grads"] = all(p.grad is not None for k, p in m3.model.named_parameters()
                        if not k.startswith("projector_model"))     # CE only: the projector gets no gradient

def with_dc(cfg):
    small(cfg)
    cfg["loss"]["losses"] = {"CrossEntropyLoss": 1, "DenseContrastiveLossV2_ms": 0.1}
m4 = bare(RefHRNetManager, "/root/reference/configs/CITYSCAPES/hrnet_contrastive_CTS.json", patch=with_dc)
RefBase.load_model(m4)
RefBase.load_loss(m4)
try:
    RefHRNetManager.forward_step(m4, img, lbl)
    out["cpu_dc"] = "ran"
except RuntimeError as e:
    out["cpu_dc"] = str(e)
_print("RESULT " + json.dumps(out))
"""


@pytest.mark.timeout(600)
def test_reference_manager_builds_and_steps_this_repos_classes():
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=env,
                       cwd=REF, timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    # every name the reference resolves through globals() is this repo's class
    assert all(v.startswith("mscs_amd.") or v.startswith("eccv2022") for v in out["globals"].values()), out["globals"]
    assert out["hrnet_model"].endswith("models.HRNet.HRNet")
    assert out["hrnet_loss"].endswith("losses.LossWrapper.LossWrapper")
    assert list(out["hrnet_terms"]) == ["CrossEntropyLoss", "DenseContrastiveLossV2_ms"]      # config order kept
    assert out["hrnet_terms"]["DenseContrastiveLossV2_ms"].endswith("losses.DenseContrastiveLossV2_ms")
    assert out["hrnet_weights"] == {"CrossEntropyLoss": 1, "DenseContrastiveLossV2_ms": 0.1}
    assert out["hrnet_return_features"] is True
    assert out["hrnet_params"] == 70391203 or abs(out["hrnet_params"] - 70.39e6) < 0.05e6
    assert out["hrnet_dc"] == {"scales": 4, "weights": [1, 0.7, 0.4, 0.1], "cross": True, "K": 20}
    assert out["upn_model"].endswith("models.UPerNet.UPerNet")
    assert "DenseContrastiveLossV2_ms" in out["upn_terms"]
    assert abs(out["upn_params"] - 62.4e6) < 0.5e6
    assert out["step_keys"] == ["feats", "interm_output", "loss", "output"]
    assert out["step_loss"] == out["step_loss"] and out["step_loss"] > 0
    assert out["step_output"] == [2, 19, 64, 64]
    assert out["step_feats"] == [[2, 256, 16, 16], [2, 256, 8, 8], [2, 256, 4, 4], [2, 256, 2, 2]]
    assert out["step_loss_vals"] == ["CrossEntropyLoss"]
    assert out["step_grads"] is True
    assert "no CPU fallback" in out["cpu_dc"]                                                    # fails loudly, by design
