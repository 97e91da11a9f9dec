"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel runtime: the manager's spawned
workers wrap the model in DDP (+ SyncBatchNorm), shard the global batch, and end every step with
identical parameters on every rank; utils.distributed helpers behave as the reference's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mscs_amd  # noqa: F401


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cfg(parallel):
    return {"name": "t", "mode": "training", "manager": "HRNet", "cuda": False, "parallel": parallel,
            "gpu_device": [0, 1], "seed": 3, "log_every_n_steps": 1000,
            "graph": {"model": "HRNet", "backbone": "hrnet18", "sync_bn": True, "pretrained": False,
                      "align_corners": True},
            "data": {"dataset": "CITYSCAPES", "experiment": 1, "batch_size": 4, "synthetic": True,
                     "synthetic_length": 8, "synthetic_mode": "blocky",
                     "transform_values": {"crop_shape": [32, 32]}},
            "loss": {"name": "LossWrapper", "losses": {"CrossEntropyLoss": 1}},
            "train": {"learning_rate": 0.01, "lr_fct": "polynomial", "optim": "SGD", "lr_batchwise": True,
                      "epochs": 1}}


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(3)
    from mscs_amd.managers import HRNetManager
    from mscs_amd.utils import set_verbosity, get_rank, get_world_size, is_distributed, concat_all_gather, \
        reduce_tensor
    set_verbosity(40)
    mgr = HRNetManager(_cfg(True), autostart=False)
    mgr.world_size = mgr.n_gpus = world
    mgr._worker_setup(rank, rank)
    assert is_distributed() and get_rank() == rank and get_world_size() == world
    assert mgr.batch_size == 2                                         # global 4 -> 2 per rank
    assert isinstance(mgr.model, torch.nn.parallel.DistributedDataParallel)
    # SyncBatchNorm only exists for GPU modules; CPU ranks keep BatchNorm (buffers broadcast by DDP)
    assert not any(isinstance(m, torch.nn.SyncBatchNorm) for m in mgr.model.modules())
    mgr.train_one_epoch()
    flat = torch.cat([p.detach().flatten() for p in mgr.model.parameters()])
    gathered = concat_all_gather(flat[None])
    assert gathered.shape[0] == world
    assert torch.equal(gathered[0], gathered[1]), "parameters diverged across ranks"
    avg = reduce_tensor(torch.tensor([float(rank + 1)]))
    if rank == 0:
        assert abs(avg.item() - 1.5) < 1e-6
        torch.save({"loss": mgr.metrics["loss"], "steps": mgr.global_step}, os.path.join(out_dir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    assert res["steps"] == 2 and res["loss"] == res["loss"]              # 8 samples / (2 ranks * 2) = 2 steps, finite


def test_single_process_helpers():
    from mscs_amd.utils import get_rank, get_world_size, is_distributed, concat_all_gather, reduce_tensor
    assert not is_distributed() and get_rank() == 0 and get_world_size() == 1
    t = torch.arange(4.0)
    assert torch.equal(concat_all_gather(t), t) and torch.equal(reduce_tensor(t), t)


def _gather_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    import numpy as np
    from mscs_amd.losses.engine import StepState, _Scale, gather_peer_banks, class_layout, _npad
    from mscs_amd.losses.plan import build_host_plan
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(10 + rank)
    st = StepState()
    for s in range(2):
        counts = rs.randint(0, 60, size=(2, 6)).astype(np.int64)
        counts[:, -1] = 99
        counts[0, rank] = 30                          # at least one qualifying pair
        plan = build_host_plan(counts, 5, 2500, 200, native_rng=False)
        bank = torch.zeros(_npad(plan.N), 256)
        bank[:plan.N] = float(rank + 1) + 0.01 * s
        st.scales.append(_Scale(plan=plan, h=1, w=1, C=256, strides=(1, 1, 1), bank=bank))
    r, banks, banks_h, layouts = gather_peer_banks(st, max_features_total=200)
    assert banks_h is None                      # f32 mode: the f32 rows travel
    assert r == rank and len(banks) == world and len(layouts) == world
    for q in range(world):
        for s in range(2):
            assert banks[q][s].shape == (_npad(200), 256)
            assert torch.all(banks[q][s][0] == float(q + 1) + 0.01 * s)
    for s in range(2):
        np.testing.assert_array_equal(layouts[rank][s], class_layout(st.scales[s].plan))
    torch.save([[l.tolist() for l in lq] for lq in layouts], os.path.join(out_dir, f"lay{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_negative_bank_all_gather_two_ranks_gloo(tmp_path):
    """The exchange step of the shared-negative-bank extension: fixed-size padded bank gather + class
    layouts, identical on every rank."""
    port = _free_port()
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), "lay0.pt"))
    b = torch.load(os.path.join(str(tmp_path), "lay1.pt"))
    assert a == b
