"""Checkpoint compatibility (second half): the manager writes and reads the
reference's checkpoint dictionary (managers/LoggingManager.py:293-368), ``train()`` leaves best / last
artefacts, ``load_checkpoint`` / ``load_last`` config keys resume, and DDP's ``module.`` prefix is matched in
both directions (single-GPU save -> DDP model and back)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import mscs_amd  # noqa: F401
from mscs_amd.managers import HRNetManager
from mscs_amd.utils import set_verbosity

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_KEYS = {"global_step", "epoch", "model_state_dict", "optimiser_state_dict", "best_loss", "best_miou",
            "final_miou", "final_miou_step", "is_best", "scheduler_state_dict"}       # LoggingManager.py:301-313


def _cfg(tmp, epochs=2, **extra):
    cfg = {"name": "ck", "mode": "training", "manager": "HRNet", "cuda": False, "parallel": False,
           "gpu_device": [0], "seed": 3, "log_every_n_steps": 1000, "log_path": str(tmp), "run_id": "run0",
           "log_every_n_epochs": 1, "max_valid_imgs": 1,
           "graph": {"model": "HRNet", "backbone": "hrnet18", "sync_bn": False, "pretrained": False,
                     "align_corners": True},
           "data": {"dataset": "CITYSCAPES", "experiment": 1, "batch_size": 2, "synthetic": True,
                    "synthetic_length": 4, "synthetic_valid_length": 1, "synthetic_mode": "blocky",
                    "transform_values": {"crop_shape": [32, 32]}},
           "loss": {"name": "LossWrapper", "losses": {"CrossEntropyLoss": 1}},
           "train": {"learning_rate": 0.01, "lr_fct": "polynomial", "optim": "SGD", "lr_batchwise": True,
                     "epochs": epochs}}
    cfg.update(extra)
    return cfg


def _manager(cfg):
    set_verbosity(40)
    torch.set_num_threads(4)
    m = HRNetManager(cfg, autostart=False)
    m.setup()
    return m


def test_train_writes_reference_layout_and_resumes(tmp_path):
    m = _manager(_cfg(tmp_path))
    m.train()
    ck = tmp_path / "run0" / "chkpts"
    names = sorted(os.listdir(ck))
    assert "chkpt_best.pt" in names and "chkpt_epoch_001.pt" in names, names
    chk = torch.load(ck / "chkpt_epoch_001.pt", weights_only=False)
    assert set(chk) == REF_KEYS
    assert chk["epoch"] == 1 and chk["is_best"] is False and chk["global_step"] == m.global_step - 1
    assert list(chk["model_state_dict"]) == list(m.model.state_dict())
    # resume through the config keys (reference BaseManager.py:76-82): last checkpoint, one more epoch
    m2 = _manager(_cfg(tmp_path, epochs=3, load_checkpoint="run0", load_last=True, run_id="run1"))
    assert m2.start_epoch == 2 and m2.global_step == chk["global_step"] and m2.best_miou == chk["best_miou"]
    for (k, a), b in zip(m.model.state_dict().items(), m2.model.state_dict().values()):
        assert torch.equal(a, b), k
    mom = m2.optimiser.state_dict()["state"]
    assert len(mom) > 0 and all("momentum_buffer" in v for v in mom.values())
    assert m2.scheduler.last_epoch == m.scheduler.last_epoch
    m2.train()
    assert sorted(os.listdir(tmp_path / "run1" / "chkpts"))[-1] == "chkpt_epoch_002.pt"
    # 'best' is the default kind
    m3 = _manager(_cfg(tmp_path, load_checkpoint=str(tmp_path / "run0"), run_id="run2"))
    best = torch.load(ck / "chkpt_best.pt", weights_only=False)
    assert best["is_best"] is True and m3.start_epoch == best["epoch"] + 1


def test_prefix_mismatch_raises_instead_of_keeping_random_weights(tmp_path):
    m = _manager(_cfg(tmp_path))
    state = m.checkpoint_state()
    state["model_state_dict"] = {"encoder." + k: v for k, v in state["model_state_dict"].items()}
    path = str(tmp_path / "foreign.pt")
    torch.save(state, path)
    with pytest.raises(RuntimeError, match="shares no parameter name"):
        m.load_checkpoint(path)
    # partial overlap: reported, not silent
    state = m.checkpoint_state()
    sd = dict(state["model_state_dict"])
    dropped = [k for k in sd if k.startswith("cls_head.2")]
    for k in dropped:
        sd.pop(k)
    sd["stray.weight"] = torch.zeros(1)
    state["model_state_dict"] = sd
    torch.save(state, path)
    m.load_checkpoint(path)
    assert m.load_report.missing_keys == dropped and m.load_report.unexpected_keys == ["stray.weight"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_single_to_ddp_and_back(tmp_path):
    """A non-prefixed (single-GPU) checkpoint loads into the DDP-wrapped model and a DDP save loads into the bare
    model, every tensor bit-identical."""
    single = _manager(_cfg(tmp_path))
    with torch.no_grad():
        for i, p in enumerate(single.model.parameters()):
            p.add_(0.001 * (i % 7))
    p1 = single.save_checkpoint(path=str(tmp_path / "single.pt"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cfg = _cfg(tmp_path, parallel=True)
        ddp = HRNetManager(cfg, autostart=False)
        ddp.world_size = ddp.n_gpus = 1
        ddp._worker_setup(0, 0)
        assert isinstance(ddp.model, torch.nn.parallel.DistributedDataParallel)
        assert all(k.startswith("module.") for k in ddp.model.state_dict())
        ddp.load_checkpoint(p1)
        assert not ddp.load_report.missing_keys and not ddp.load_report.unexpected_keys
        for (k, a), b in zip(single.model.state_dict().items(), ddp.model.state_dict().values()):
            assert torch.equal(a, b), k
        with torch.no_grad():
            next(ddp.model.parameters()).mul_(2.0)
        p2 = ddp.save_checkpoint(path=str(tmp_path / "ddp.pt"))
        assert all(k.startswith("module.") for k in torch.load(p2, weights_only=False)["model_state_dict"])
    finally:
        dist.destroy_process_group()
    back = _manager(_cfg(tmp_path))
    back.load_checkpoint(p2)
    assert not back.load_report.missing_keys and not back.load_report.unexpected_keys
    for (k, a), b in zip(ddp.model.state_dict().items(), back.model.state_dict().values()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("fixture", ["G7_hrnet48_ms4", "G7_upernet_swinT_fpn"])
def test_reference_format_checkpoint_file_loads(tmp_path, fixture):
    """A .pt file laid out as the reference writes it -- key names and shapes taken from the manifest of the
    REFERENCE model's state_dict (tests/golden/G7_*.npz, written by tools/gen_golden_models.py), saved from a DDP
    run (``module.`` prefix, LoggingManager.py:303) -- loads with every key matched."""
    z = np.load(os.path.join(GOLD, fixture + ".npz"), allow_pickle=False)
    man = json.loads(str(z["manifest_json"]))
    graph = json.loads(str(z["config_json"]))
    graph["model"] = "HRNet" if "hrnet" in fixture else "UPerNet"
    gen = torch.Generator().manual_seed(5)
    sd = {"module." + k: (torch.randn(shape, generator=gen) if len(shape) else torch.tensor(3))
          for k, shape in man.items()}
    torch.save({"global_step": 99, "epoch": 7, "model_state_dict": sd, "optimiser_state_dict": {},
                "best_loss": 0.5, "best_miou": 0.25, "final_miou": 0.2, "final_miou_step": 98, "is_best": True,
                "scheduler_state_dict": None}, tmp_path / "chkpt_best.pt")
    cfg = _cfg(tmp_path, mode="inference")
    cfg["graph"] = dict(graph, dataset=graph.get("dataset", "CITYSCAPES"))
    cfg["data"]["dataset"] = cfg["graph"]["dataset"]
    cfg["data"]["experiment"] = int(z["experiment"])
    from mscs_amd.managers import BaseManager
    m = BaseManager(cfg, autostart=False)
    set_verbosity(40)
    m.device = torch.device("cpu")
    m.load_model()
    m.load_checkpoint(str(tmp_path / "chkpt_best.pt"))
    assert not m.load_report.missing_keys and not m.load_report.unexpected_keys
    got = m.model.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k[len("module."):]], v.to(got[k[len("module."):]].dtype)), k
