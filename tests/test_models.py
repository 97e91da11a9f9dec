"""Model parity (CPU): state_dict manifests and eval-mode forward outputs of mscs_amd.models against
fixtures generated from the reference models with name-seeded weights (tools/gen_golden_models.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden_names

sys.path.insert(0, os.path.join(ROOT, "tools"))
from model_fill import fill_state_dict_, model_input  # noqa: E402

import mscs_amd  # noqa: F401,E402
from mscs_amd.utils import set_verbosity  # noqa: E402

set_verbosity(40)


def _flatten(out):
    res = []
    if isinstance(out, (list, tuple)):
        for o in out:
            res += _flatten(o)
    elif torch.is_tensor(out):
        res.append(out)
    return res


def _build(name, cfg, exp):
    from mscs_amd import models
    cls = models.HRNet if "hrnet" in name else models.UPerNet
    return cls(config=cfg, experiment=exp)


def _available(name):
    from mscs_amd import models
    return "hrnet" in name or hasattr(models, "UPerNet")


@pytest.mark.parametrize("name", golden_names(["G7_"]))
def test_model_matches_reference(name):
    if not _available(name):
        pytest.skip("model family not built yet")
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = json.loads(str(z["config_json"]))
    model = _build(name, cfg, int(z["experiment"]))
    manifest = json.loads(str(z["manifest_json"]))
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(mine.keys()) == list(manifest.keys()), "state_dict keys / order differ from the reference"
    assert mine == manifest
    fill_state_dict_(model)
    model.eval()
    with torch.no_grad():
        outs = _flatten(model(model_input(tuple(int(v) for v in z["input_shape"]))))
    assert len(outs) == int(z["n_outputs"])
    for i, o in enumerate(outs):
        assert list(o.shape) == list(z[f"out{i}_shape"])
        ref = z[f"out{i}_sample"]
        got = o.flatten()[::int(z[f"out{i}_step"])].numpy()
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got, ref, atol=tol, rtol=1e-4)
        np.testing.assert_allclose(o.double().abs().sum().item(), float(z[f"out{i}_abs_sum"]), rtol=1e-4)


def test_hrnet_train_mode_backward_runs():
    from mscs_amd.models import HRNet
    cfg = {'backbone': 'hrnet18', 'pretrained': False, 'dataset': 'CITYSCAPES', 'align_corners': True,
           'projector': {'mlp': [[1, -1, 1]], 'd': 64, 'use_bn': True}}
    m = HRNet(cfg, 1)
    assert m.backbone_out_channels == 18 + 36 + 72 + 144
    out, proj = m(torch.randn(2, 3, 64, 64))
    assert out.shape == (2, 19, 64, 64) and proj.shape == (2, 64, 16, 16)
    (out.mean() + proj.mean()).backward()
    assert all(p.grad is not None for p in m.parameters())


def test_lovasz_softmax_matches_reference():
    from mscs_amd.losses import LovaszSoftmax
    z = np.load(os.path.join(GOLDEN, "G10_lovasz.npz"))
    for name in ("default", "per_image", "all"):
        m = LovaszSoftmax(json.loads(str(z[name + "_cfg"])))
        x = torch.from_numpy(z["logits"]).requires_grad_(True)
        loss = m(x, torch.from_numpy(z["label"].astype(np.int64)))
        loss.backward()
        np.testing.assert_allclose(loss.item(), z[name + "_loss"], rtol=1e-5)
        np.testing.assert_allclose(x.grad.numpy(), z[name + "_grad"], atol=1e-6 * np.abs(z[name + "_grad"]).max() + 1e-9)
