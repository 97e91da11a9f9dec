"""Two ranks, either on two GPUs over RCCL (backend "nccl", one device per rank: what an 8-GPU node runs) when the box
has at least two devices, or on ONE GPU over gloo (CUDA tensors moved through the host; RCCL refuses two ranks on one
device): SyncBatchNorm semantics of FusedBatchNorm2d, the DDP training step of the manager with the HIP loss, and the
shared-negative-bank all-gather.  gloo's host-staged copies synchronise the issuing stream, so only the nccl variant
can expose a missing stream dependency around a collective; every collective of the package is issued on the CURRENT
stream of the calling code (the branch's stream inside HighResolutionModule), never on a hidden one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _backend():
    """("nccl", one device per rank) when the box has >= 2 GPUs, else ("gloo", every rank on cuda:0).  Counting devices
    does not initialise the GPU, so this is safe in the pytest parent before mp.spawn."""
    return "nccl" if torch.cuda.device_count() >= 2 else "gloo"


def _device_of(rank, backend):
    return rank if backend == "nccl" else 0


def _bn_worker(rank, world, port, out_dir, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import mscs_amd  # noqa: F401
    from mscs_amd.models.fused_bn import FusedBatchNorm2d
    dev = torch.device("cuda", _device_of(rank, backend))
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    shape = (4, 24, 12, 20)
    x = torch.randn(shape, generator=g) * 1.5 + 0.3
    r = torch.randn(shape, generator=g)
    gy = torch.randn(shape, generator=g)
    w = torch.rand(24, generator=g) + 0.5
    b = torch.rand(24, generator=g) - 0.5
    half = slice(2 * rank, 2 * rank + 2)
    bn = FusedBatchNorm2d(24).to(dev)
    bn.sync = True
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b)
    xi = x[half].to(dev).requires_grad_(True)
    ri = r[half].to(dev).requires_grad_(True)
    y = bn(xi, residual=ri, relu=True)
    y.backward(gy[half].to(dev))
    out = {"y": y.detach().cpu(), "dx": xi.grad.cpu(), "dr": ri.grad.cpu(), "dw": bn.weight.grad.cpu(),
           "db": bn.bias.grad.cpu(), "rm": bn.running_mean.cpu(), "rv": bn.running_var.cpu()}
    # the same layer twice more as the two members of ONE stacked exchange (models/fused_bn.bn_act_group), the second
    # member on a side stream: must give the single-layer results
    from mscs_amd.models import fused_bn
    side = torch.cuda.Stream(device=dev)
    bns = []
    for _ in range(2):
        b2 = FusedBatchNorm2d(24).to(dev)
        b2.sync = True
        with torch.no_grad():
            b2.weight.copy_(w); b2.bias.copy_(b)
        bns.append(b2)
    xs = [x[half].to(dev).requires_grad_(True) for _ in range(2)]
    rs = [r[half].to(dev).requires_grad_(True) for _ in range(2)]
    before = fused_bn.COLLECTIVES["count"]
    torch.cuda.synchronize()
    ys = fused_bn.bn_act_group(bns, xs, residuals=rs, relu=True, streams=[None, side])
    torch.cuda.current_stream().wait_stream(side)
    ((ys[0] * gy[half].to(dev)).sum() + (ys[1] * gy[half].to(dev)).sum()).backward()
    torch.cuda.synchronize()
    out["group_collectives"] = fused_bn.COLLECTIVES["count"] - before
    for k in range(2):
        out[f"g{k}"] = {"y": ys[k].detach().cpu(), "dx": xs[k].grad.cpu(), "dr": rs[k].grad.cpu(),
                        "dw": bns[k].weight.grad.cpu(), "rm": bns[k].running_mean.cpu()}
    torch.save(out, os.path.join(out_dir, f"bn{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_fused_bn_sync_two_ranks_one_gpu(tmp_path):
    port = _free_port()
    mp.spawn(_bn_worker, args=(2, port, str(tmp_path), _backend()), nprocs=2, join=True)
    g = torch.Generator().manual_seed(5)
    shape = (4, 24, 12, 20)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).double().requires_grad_(True)
    r = torch.randn(shape, generator=g).double().requires_grad_(True)
    gy = torch.randn(shape, generator=g).double()
    w = torch.rand(24, generator=g) + 0.5
    b = torch.rand(24, generator=g) - 0.5
    ref = torch.nn.BatchNorm2d(24).double()
    with torch.no_grad():
        ref.weight.copy_(w); ref.bias.copy_(b)
    y = torch.relu(ref(x) + r)
    y.backward(gy)
    outs = [torch.load(os.path.join(str(tmp_path), f"bn{q}.pt")) for q in range(2)]
    for q in range(2):
        half = slice(2 * q, 2 * q + 2)
        for key, want in (("y", y.detach()[half]), ("dx", x.grad[half]), ("dr", r.grad[half])):
            assert (outs[q][key].double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), key
        assert torch.allclose(outs[q]["rm"].double(), ref.running_mean, atol=1e-5)
        assert torch.allclose(outs[q]["rv"].double(), ref.running_var, atol=1e-5)
        assert outs[q]["group_collectives"] == 2            # one stacked exchange forward, one backward
        for k in range(2):                   # the grouped (stacked-exchange) form: identical to the single layer
            for key in ("y", "dx", "dr", "dw", "rm"):
                assert torch.equal(outs[q][f"g{k}"][key], outs[q][key]), (k, key)
    # weight / bias grads are per-rank partial sums (DDP averages them): they add up to the full-batch grads
    assert torch.allclose((outs[0]["dw"] + outs[1]["dw"]).double(), ref.weight.grad, atol=2e-4)
    assert torch.allclose((outs[0]["db"] + outs[1]["db"]).double(), ref.bias.grad, atol=2e-4)


def _train_worker(rank, world, port, out_dir, global_negatives, backbone="hrnet18", scales=2, backend="gloo", no_streamk_rank=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import mscs_amd  # noqa: F401
    from mscs_amd.managers import HRNetManager
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    if rank == no_streamk_rank:
        from mscs_amd import _lib
        _lib.lib().dcl_infonce_set_streamk(0)          # this rank's loss backward launches no stream-K kernel
    cfg = {"name": "t", "mode": "training", "manager": "HRNet", "cuda": True, "parallel": True,
           "gpu_device": [_device_of(q, backend) for q in range(world)], "seed": 3, "log_every_n_steps": 1000,
           "dist_backend": backend,
           "graph": {"model": "HRNet", "backbone": backbone, "sync_bn": True, "pretrained": False,
                     "align_corners": True,
                     "ms_projector": {"mlp": [[1, -1, 1]], "scales": scales, "d": 64, "use_bn": True}},
           "data": {"dataset": "CITYSCAPES", "experiment": 1, "batch_size": 4, "synthetic": True,
                    "synthetic_length": 8, "synthetic_mode": "blocky",
                    "transform_values": {"crop_shape": [64, 128]}},
           "loss": {"name": "LossWrapper", "losses": {"CrossEntropyLoss": 1, "DenseContrastiveLossV2_ms": 0.1},
                    "temperature": 0.1, "scales": scales, "weights": [1.0, 0.5, 0.3][:scales],
                    "cross_scale_contrast": True,
                    "min_views_per_class": 2, "max_features_total": 600, "global_negatives": global_negatives},
           "train": {"learning_rate": 0.01, "lr_fct": "polynomial", "optim": "SGD", "lr_batchwise": True,
                     "epochs": 1}}
    mgr = HRNetManager(cfg, autostart=False)
    mgr.world_size = mgr.n_gpus = world
    mgr._worker_setup(_device_of(rank, backend), rank)
    mgr.train_one_epoch()
    flat = torch.cat([p.detach().flatten() for p in mgr.model.parameters()]).cpu()
    from mscs_amd.models.ops import DirectConv2d
    direct = sum(isinstance(m, DirectConv2d) for m in mgr.model.modules())
    from mscs_amd.models import fused_bn
    torch.save({"params": flat, "metrics": mgr.metrics, "direct_convs": direct, "syncbn_collectives": fused_bn.COLLECTIVES["count"],
                "syncbn_host_waits": fused_bn.COLLECTIVES["host_waits"], "backend": backend, "steps": mgr.global_step,
                "segs": [len(t.segs) for t in mgr.loss.loss_classes["DenseContrastiveLossV2_ms"].last_state.terms]},
               os.path.join(out_dir, f"train{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("global_negatives", [False, True])
def test_ddp_training_step_two_ranks_one_gpu(tmp_path, global_negatives):
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path), global_negatives, "hrnet18", 2, _backend()), nprocs=2,
             join=True)
    a, b = [torch.load(os.path.join(str(tmp_path), f"train{q}.pt")) for q in range(2)]
    assert torch.equal(a["params"], b["params"]), "parameters diverged across ranks"
    assert np.isfinite(a["metrics"]["loss"]) and np.isfinite(b["metrics"]["loss"])
    assert a["segs"] == ([2, 2, 2] if global_negatives else [1, 1, 1])


@pytest.mark.timeout(600)
def test_ddp_step_when_one_rank_runs_the_loss_backward_without_streamk(tmp_path):
    """the all-rank exchange of the stream-K error word must be issued by EVERY rank in every backward pass, also by a
    rank whose backward launched no stream-K kernel (switched off on it) -- a collective that only some ranks issue pairs up with
    the peers' next gradient bucket and hangs or corrupts the gradients.  Rank 1 runs column-split, rank 0 stream-K: the epoch
    finishes and the parameters agree bit for bit across the ranks."""
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path), False, "hrnet18", 2, _backend(), 1), nprocs=2, join=True)
    a, b = [torch.load(os.path.join(str(tmp_path), f"train{q}.pt")) for q in range(2)]
    assert torch.equal(a["params"], b["params"]), "parameters diverged across ranks"
    assert np.isfinite(a["metrics"]["loss"]) and np.isfinite(b["metrics"]["loss"])


@pytest.mark.timeout(900)
def test_ddp_hrnet48_direct_kernels_and_branch_streams_two_ranks_one_gpu(tmp_path):
    """The W48 backbone under DDP + fused SyncBatchNorm with two ranks: unlike hrnet18 its channel counts (48 / 96 /
    192 / 384, head 720) take the direct f16x3 forward, data-gradient AND weight-gradient kernels, on one HIP stream
    per branch, with the blocking per-norm all-reduce between the statistics and the apply kernels -- the schedule
    that runs on an 8-GPU node (there over RCCL; gloo here because both ranks share the one GPU of this box).
    Shared negative bank on: the overlapped per-scale bank gathers run too."""
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path), True, "hrnet48", 3, _backend()), nprocs=2, join=True)
    a, b = [torch.load(os.path.join(str(tmp_path), f"train{q}.pt")) for q in range(2)]
    assert a["direct_convs"] > 250                                   # every 3x3 convolution of W48 is a DirectConv2d
    assert torch.equal(a["params"], b["params"]), "parameters diverged across ranks"
    assert np.isfinite(a["metrics"]["loss"]) and np.isfinite(b["metrics"]["loss"])
    assert a["segs"] == [2] * 5                                      # 3 intra + 2 cross terms, 2 segments each
    # SyncBatchNorm exchanges per step: 310 norms x 2 directions = 620 one by one; the branch chains' 208 norms share one
    # exchange per block depth (64 instead of 208 per direction)
    per_step = a["syncbn_collectives"] / max(1, a["steps"])
    assert per_step <= 340, per_step
    # every exchange is issued asynchronously and awaited on the STREAM right before its consumer (fused_bn._Exchange): on RCCL
    # (DCL_TEST_BACKEND=nccl, a node with two GPUs) the host never waits; gloo -- the stand-in on one-GPU boxes -- completes
    # every collective on the host
    assert a["syncbn_host_waits"] == (0 if a["backend"] == "nccl" else a["syncbn_collectives"])
