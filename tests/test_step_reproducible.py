"""A whole training step is reproducible BITWISE from run to run.

Config 2: HRNet-W48 + LossWrapper(CE + 0.1 DenseContrastiveLossV2_ms, cross-scale); config 4: UPerNet + Swin-T with the same loss --
through the managers (reference managers/BaseManager.py:302-345 train_one_epoch's loop body: forward, loss, backward, optimizer,
schedule), every stream of the models in use, seeded sampling in the loss.  Three managers built from the same seed take two
optimizer steps on the same resident batch; both losses and every parameter and buffer afterwards must be identical.  What this
guards: cross-stream races, kernels whose results depend on what runs beside them ("Packed FP32 beside
MFMA"), atomics in any kernel of the step (the library's weight gradient of the 3-channel stem was the last one:
models/ops_conv.py).  The step configurations are bench.py's (the ones the benchmark times), at a small crop."""
import importlib.util
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv):
    spec = importlib.util.spec_from_file_location("bench_for_repro", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    keep = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return mod, mod.parse()
    finally:
        sys.argv = keep


@pytest.mark.gpu
@pytest.mark.parametrize("config,hw,labels", [(2, (128, 256), "iid"), (4, (512, 512), "blocky")])
def test_training_step_is_bitwise_reproducible(config, hw, labels):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    import mscs_amd  # noqa: F401
    from mscs_amd.managers import HRNetManager, OCRNetManager
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    bench, args = _bench(["--config", str(config), "--height", str(hw[0]), "--width", str(hw[1]), "--batch", "2", "--labels", labels])
    dev = torch.device("cuda:0")
    first = None
    for run in range(3):
        torch.manual_seed(0)
        mgr = (OCRNetManager if config in (4, 5) else HRNetManager)(bench.step_config(args, 1), autostart=False)
        mgr.setup()
        mgr.model.train()
        gen = torch.Generator().manual_seed(0)
        img = torch.randn(args.batch, 3, args.height, args.width, generator=gen).to(dev)
        lbl = bench.synth_labels(args, args.batch, args.height, args.width, gen).to(dev)
        losses = []
        for _ in range(2):
            mgr.optimiser.zero_grad(set_to_none=True)
            ret = mgr.forward_step(img, lbl)
            ret["loss"].backward()
            mgr.optimiser.step()
            mgr.scheduler.step()
            losses.append(ret["loss"].detach().clone())
        torch.cuda.synchronize()
        assert all(torch.isfinite(l) for l in losses)
        state = {k: v.detach().clone() for k, v in mgr.model.state_dict().items()}
        state.update({f"loss{i}": l for i, l in enumerate(losses)})
        if first is None:
            first = state
        else:
            bad = [k for k in first if not torch.equal(state[k], first[k])]
            assert not bad, f"run {run}: {len(bad)} of {len(first)} tensors differ from run 0, e.g. {bad[:5]}"
        del mgr


@pytest.mark.gpu
def test_training_step_with_deferred_norms_is_bitwise_the_step_that_writes_them():
    """Config 2 (HRNet-W48 + CE + multi-scale / cross-scale contrastive loss through the manager): two optimizer steps with the
    norms of `conv -> bn -> relu -> conv` deferred into the consuming convolution (the default: 108 BasicBlocks / Bottlenecks, the
    fuse layers' two-step chains, the stem; csrc/dcl_conv3x3_pre.hip) against the same steps with every norm writing its output
    (DCL_FUSE_BN_APPLY=0, the round-5 path): both losses, every parameter and every buffer bitwise equal -- the deferred path is an
    exact re-arrangement (same fma, same operand scales, same tiles), not an approximation, on the whole model as on its blocks
    (tests/test_pre_norm_conv.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    import mscs_amd  # noqa: F401
    from mscs_amd.debug import cfg
    from mscs_amd.managers import HRNetManager
    from mscs_amd.models import fused_bn
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    bench, args = _bench(["--config", "2", "--height", "128", "--width", "256", "--batch", "2"])
    dev = torch.device("cuda:0")
    states, deferred = [], []
    keep = cfg.fuse_bn_apply
    try:
        for fuse in (True, False):
            cfg.fuse_bn_apply = fuse
            n0 = fused_bn.DEFERRED["count"]
            torch.manual_seed(0)
            mgr = HRNetManager(bench.step_config(args, 1), autostart=False)
            mgr.setup()
            mgr.model.train()
            gen = torch.Generator().manual_seed(0)
            img = torch.randn(args.batch, 3, args.height, args.width, generator=gen).to(dev)
            lbl = bench.synth_labels(args, args.batch, args.height, args.width, gen).to(dev)
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
            states.append(state)
            deferred.append(fused_bn.DEFERRED["count"] - n0)
            del mgr
    finally:
        cfg.fuse_bn_apply = keep
    assert deferred[0] >= 2 * 120 and deferred[1] == 0, deferred        # 104 BasicBlocks + 4 Bottlenecks + chains + stem, two steps
    bad = [k for k in states[0] if not torch.equal(states[0][k], states[1][k])]
    assert not bad, f"{len(bad)} of {len(states[0])} tensors differ between the deferred and the written path, e.g. {bad[:5]}"
