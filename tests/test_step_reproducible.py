"""A whole training step is reproducible BITWISE from run to run.

HRNet-W48 + LossWrapper(CE + 0.1 DenseContrastiveLossV2_ms, cross-scale) through the manager (reference
managers/BaseManager.py:302-345 train_one_epoch's loop body: forward, loss, backward, SGD, schedule), every stream of the
model in use, seeded sampling in the loss: three managers built from the same seed take two optimizer steps on the same
resident batch; both losses and every parameter and buffer afterwards must be identical.  What this guards: cross-stream
races, kernels that depend on what runs beside them ("Packed FP32 beside MFMA"), atomics in any kernel
of the step (the library's weight gradient of the 3-channel stem was the last one: models/ops_conv.py)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_training_step_is_bitwise_reproducible():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    import mscs_amd  # noqa: F401
    from mscs_amd.managers import HRNetManager
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    dev = torch.device("cuda:0")
    H, W, B, S = 128, 256, 2, 3
    cfg = {
        "name": "repro", "mode": "training", "manager": "HRNet", "cuda": True, "seed": 0, "parallel": False,
        "graph": {"model": "HRNet", "backbone": "hrnet48", "sync_bn": True, "out_stride": 4, "pretrained": False,
                  "align_corners": True,
                  "ms_projector": {"mlp": [[1, -1, 1]], "scales": S, "d": 256, "use_bn": True, "before_context": True}},
        "data": {"dataset": "CITYSCAPES", "experiment": 1, "batch_size": B, "num_workers": 0, "synthetic": True,
                 "synthetic_length": 2 * B, "transform_values": {"crop_shape": [H, W]}},
        "loss": {"name": "LossWrapper", "losses": {"CrossEntropyLoss": 1, "DenseContrastiveLossV2_ms": 0.1},
                 "dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": S, "weights": [1.0, 0.7, 0.4],
                 "cross_scale_contrast": True, "min_views_per_class": 5, "max_views_per_class": 2500,
                 "max_features_total": 10000, "label_scaling_mode": "nn"},
        "train": {"learning_rate": 0.01, "lr_fct": "polynomial", "optim": "SGD", "lr_batchwise": True, "epochs": 4,
                  "momentum": 0.9, "weight_decay": 0.0005},
    }
    import copy
    gen = torch.Generator().manual_seed(3)
    img = torch.randn(B, 3, H, W, generator=gen).to(dev)
    lbl = torch.randint(0, 19, (B, H, W), generator=gen).to(dev)
    first = None
    for run in range(3):
        torch.manual_seed(0)
        mgr = HRNetManager(copy.deepcopy(cfg), autostart=False)
        mgr.setup()
        mgr.model.train()
        losses = []
        for _ in range(2):
            mgr.optimiser.zero_grad(set_to_none=True)
            ret = mgr.forward_step(img, lbl)
            ret["loss"].backward()
            mgr.optimiser.step()
            mgr.scheduler.step()
            losses.append(ret["loss"].detach().clone())
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in mgr.model.state_dict().items()}
        state.update({f"loss{i}": l for i, l in enumerate(losses)})
        assert all(torch.isfinite(l) for l in losses)
        if first is None:
            first = state
        else:
            bad = [k for k in first if not torch.equal(state[k], first[k])]
            assert not bad, f"run {run}: {len(bad)} of {len(first)} tensors differ from run 0, e.g. {bad[:5]}"
        del mgr
