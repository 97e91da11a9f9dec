#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE on CPU (build container only).

Usage:  python tools/gen_golden.py [--out tests/golden]

The reference checkout (/root/reference) is imported through tools/ref_shim.py.
Every fixture stores the inputs (labels, features, config json, seed) and the
outputs the reference produced: the sampling plan of every scale (pair list, V,
sampled pixel indices -- captured by instrumenting ``sample_anchors_fast`` with
index-encoding probe features), per-scale / cross-scale losses, the total, and
the feature gradients from autograd.  Fixtures are data only; no reference source
is copied.  torch version is recorded because ``torch.randperm``'s CPU algorithm
is version dependent ("Bit-exact sampling").
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
ref_shim.quiet()
import builtins  # noqa: E402

_print = builtins.print
builtins.print = lambda *a, **k: None      # the reference prints from constructors
from losses import DenseContrastiveLossV2, DenseContrastiveLossV2_ms, LossWrapper  # noqa: E402
from losses import TwoScaleLoss  # noqa: E402
builtins.print = _print


def blocky_labels(gen, n, H, W, block, classes, p=None):
    """Random class per block x block cell -> label map [n, H, W] (uint8)."""
    hb, wb = H // block, W // block
    classes = torch.as_tensor(classes)
    if p is None:
        idx = torch.randint(0, len(classes), (n, hb, wb), generator=gen)
    else:
        idx = torch.multinomial(torch.tensor(p), n * hb * wb, replacement=True, generator=gen).view(n, hb, wb)
    lab = classes[idx]
    lab = lab.repeat_interleave(block, 1).repeat_interleave(block, 2)
    return lab.to(torch.uint8)


class PlanRecorder:
    """Wraps DCV2.sample_anchors_fast on given instances; records (features, labels) returned."""

    def __init__(self, modules):
        self.records = []
        self._orig = []
        for m in modules:
            orig = m.sample_anchors_fast
            self._orig.append((m, orig))

            def wrapped(dc, feats, _orig=orig):
                out = _orig(dc, feats)
                self.records.append((out[0].detach().clone(), out[1].detach().clone()))
                return out
            m.sample_anchors_fast = wrapped

    def restore(self):
        for m, orig in self._orig:
            m.sample_anchors_fast = orig


def probe_features(n, h, w):
    """Channel 0 = flat pixel index, channel 1 = image index (exact in fp32)."""
    f = torch.zeros(n, 2, h, w)
    f[:, 0] = torch.arange(h * w, dtype=torch.float32).view(1, h, w)
    f[:, 1] = torch.arange(n, dtype=torch.float32).view(n, 1, 1)
    return f


def run_ms(cfg, label, feats, seed, record_randperm=False):
    """Run DCV2_ms twice with the same seed: probe pass (plans) + real pass (losses/grads)."""
    builtins.print = lambda *a, **k: None
    mod = DenseContrastiveLossV2_ms(dict(cfg))
    builtins.print = _print
    for k in ("num_all_classes", "ignore_class"):
        if "_override_" + k in cfg:
            setattr(mod, k, cfg["_override_" + k])
            for s in range(mod.scales):
                setattr(getattr(mod, f"DCV2_scale{s}"), k, cfg["_override_" + k])
    subs = [getattr(mod, f"DCV2_scale{s}") for s in range(mod.scales)]
    lab = label.long()
    out = {}

    # pass A: plans
    rec = PlanRecorder(subs)
    torch.manual_seed(seed)
    probes = [probe_features(f.shape[0], f.shape[2], f.shape[3]) for f in feats]
    rp_log = []
    orig_rp = torch.randperm
    if record_randperm:
        def rp(n, *a, **k):
            p = orig_rp(n, *a, **k)
            rp_log.append((int(n), p[:8].tolist()))
            return p
        torch.randperm = rp
    try:
        mod(lab, probes)
    finally:
        torch.randperm = orig_rp
        rec.restore()
    for s, (sf, sl) in enumerate(rec.records):
        out[f"s{s}_pix"] = sf[:, 0, :].round().to(torch.int32).numpy()
        out[f"s{s}_pair_b"] = sf[:, 1, 0].round().to(torch.int32).numpy()
        out[f"s{s}_pair_k"] = sl.round().to(torch.int32).numpy()
        out[f"s{s}_V"] = np.int32(sf.shape[2])
        out[f"s{s}_log_this_step"] = np.bool_(subs[s].log_this_step)
    if record_randperm:
        out["randperm_n"] = np.array([n for n, _ in rp_log], dtype=np.int32)
        first = np.full((len(rp_log), 8), -1, dtype=np.int32)
        for i, (_, p) in enumerate(rp_log):
            first[i, :len(p)] = p
        out["randperm_first8"] = first

    # pass B: values
    for m in subs:
        m.log_this_step = False
    fs = [f.clone().requires_grad_(True) for f in feats]
    torch.manual_seed(seed)
    loss = mod(lab, fs)
    loss.backward()
    out["loss"] = np.float32(loss.item())
    out["ms_losses"] = np.array([x.item() for x in mod.ms_losses], dtype=np.float32)
    out["cs_losses"] = np.array([x.item() for x in mod.cs_losses], dtype=np.float32)
    for s, f in enumerate(fs):
        out[f"s{s}_grad"] = (f.grad if f.grad is not None else torch.zeros_like(f)).numpy()
    return out


def run_single(cfg, label, feat, seed):
    builtins.print = lambda *a, **k: None
    mod = DenseContrastiveLossV2(dict(cfg))
    builtins.print = _print
    for k in ("num_all_classes", "ignore_class"):
        if "_override_" + k in cfg:
            setattr(mod, k, cfg["_override_" + k])
    lab = label.long()
    out = {}
    rec = PlanRecorder([mod])
    torch.manual_seed(seed)
    mod(lab, probe_features(feat.shape[0], feat.shape[2], feat.shape[3]))
    rec.restore()
    sf, sl = rec.records[0]
    out["s0_pix"] = sf[:, 0, :].round().to(torch.int32).numpy()
    out["s0_pair_b"] = sf[:, 1, 0].round().to(torch.int32).numpy()
    out["s0_pair_k"] = sl.round().to(torch.int32).numpy()
    out["s0_V"] = np.int32(sf.shape[2])
    out["s0_log_this_step"] = np.bool_(mod.log_this_step)
    f = feat.clone().requires_grad_(True)
    torch.manual_seed(seed)
    res = mod(lab, f)
    loss = res[0] if isinstance(res, tuple) else res
    if isinstance(res, tuple):
        out["sampled_features"] = res[1].detach().numpy()
        out["sampled_labels"] = res[2].detach().numpy()
    loss.backward()
    out["loss"] = np.float32(loss.item())
    out["s0_grad"] = f.grad.numpy()
    return out


def save(path, cfg, label, feats, seed, out):
    d = dict(out)
    d["config_json"] = np.array(json.dumps(cfg))
    d["seed"] = np.int64(seed)
    d["label"] = label.numpy().astype(np.uint8)
    for s, f in enumerate(feats):
        d[f"feat{s}"] = f.numpy().astype(np.float32)
    d["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(path, **d)
    _print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)  loss={out.get('loss')}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    g = torch.Generator().manual_seed(1234)

    base = {"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "min_views_per_class": 5,
            "max_views_per_class": 2500, "max_features_total": 10000, "label_scaling_mode": "nn"}

    # ---- G1: single scale, 3 real classes + ignore (attribute override as DCV2.py:238-239)
    lab = torch.randint(0, 4, (2, 128, 128), generator=g).to(torch.uint8)
    feat = torch.randn(2, 16, 32, 32, generator=g)
    cfg = dict(base, _override_num_all_classes=4, _override_ignore_class=3)
    save(os.path.join(args.out, "G1_single_scale.npz"), cfg, lab, [feat], 0, run_single(cfg, lab, feat, 0))

    # ---- G1b: single scale returning the 4-tuple (cross_scale_contrast=True on DCV2)
    cfg = dict(base, cross_scale_contrast=True, _override_num_all_classes=4, _override_ignore_class=3)
    save(os.path.join(args.out, "G1b_single_scale_tuple.npz"), cfg, lab, [feat], 3,
         run_single(cfg, lab, feat, 3))

    # ---- G2: four scales + cross-scale, CITYSCAPES K=20, blocky labels + planted rare class
    def g2_inputs():
        lab = blocky_labels(g, 2, 128, 256, 32, [0, 1, 2, 5, 8, 13, 19], p=[.25, .2, .15, .15, .1, .1, .05])
        lab[0, 3:40:4, 5:60:4] = 11          # rare class: visible at stride 4 only (offset 3,5 -> not on the grid)
        lab[1, 0:64:8, 0:64:8] = 17          # rare class on the stride-8 grid: 64 px at s4/s8, 16 at s16, 4 at s32
        feats = [torch.randn(2, 32, 128 // s, 256 // s, generator=g) for s in (4, 8, 16, 32)]
        return lab, feats
    lab2, feats2 = g2_inputs()
    cfg2 = dict(base, scales=4, weights=[1, 0.7, 0.4, 0.1], cross_scale_contrast=True)
    out2 = run_ms(cfg2, lab2, feats2, 0, record_randperm=True)
    rp = {k: out2.pop(k) for k in ("randperm_n", "randperm_first8")}
    save(os.path.join(args.out, "G2_ms4_cross.npz"), cfg2, lab2, feats2, 0, out2)
    # ---- G8: RNG-order pin for the G2 run
    np.savez_compressed(os.path.join(args.out, "G8_rng_order.npz"), seed=np.int64(0),
                        torch_version=np.array(torch.__version__), **rp)
    _print("wrote G8_rng_order.npz", rp["randperm_n"].shape)

    # ---- G3: cap branches of _select_views_per_class
    lab3 = torch.randint(0, 20, (2, 64, 128), generator=g).to(torch.uint8)
    feats3 = [torch.randn(2, 8, 16, 32, generator=g)]
    for name, over in [("G3a_maxviews1", {"max_views_per_class": 1}),
                       ("G3b_maxviews7", {"max_views_per_class": 7}),
                       ("G3c_totalcap", {"max_features_total": 100}),
                       ("G3d_both", {"max_views_per_class": 7, "max_features_total": 90})]:
        cfg = dict(base, **over)
        save(os.path.join(args.out, name + ".npz"), cfg, lab3, feats3, 7, run_single(cfg, lab3, feats3[0], 7))

    # ---- G4: cross-scale rows with P_i = 0 (class 11 qualifies at scale 0 only)
    lab4 = blocky_labels(g, 2, 64, 128, 16, [0, 1, 2, 3])
    lab4[0, 0:32:4, 0:32:4] = 11            # 64 px at stride 4; stride 16 sees 2x2 = 4 px < 5 -> absent there
    feats4 = [torch.randn(2, 16, 16, 32, generator=g), torch.randn(2, 16, 4, 8, generator=g)]
    cfg4 = dict(base, scales=2, weights=[1.0, 0.5], cross_scale_contrast=True)
    save(os.path.join(args.out, "G4_cross_zero_pos.npz"), cfg4, lab4, feats4, 11, run_ms(cfg4, lab4, feats4, 11))

    # ---- G5: variants of G2: detach_deepest, w_high_low/mid, cross_scale_temperature quirk, S=2, S=3
    small = [f[:, :16].contiguous() for f in feats2]
    variants = {
        "G5a_detach": dict(cfg2, detach_deepest=True),
        "G5b_weights": dict(cfg2, w_high_low=0.3, w_high_mid=2.0),
        "G5c_cs_temp_quirk": dict(cfg2, temperature=0.2, cross_scale_temperature=0.7),
        "G5d_S2": dict(cfg2, scales=2, weights=[1.0, 0.25]),
        "G5e_S3": dict(cfg2, scales=3, weights=[1.0, 0.7, 0.4]),
        "G5f_S3_nocross": dict(cfg2, scales=3, weights=[1.0, 0.7, 0.4], cross_scale_contrast=False),
    }
    for name, cfg in variants.items():
        S = cfg["scales"]
        save(os.path.join(args.out, name + ".npz"), cfg, lab2, small[:S], 5, run_ms(cfg, lab2, small[:S], 5))

    # ---- G9: odd sizes (H, W not multiples of the stride; exercises ATen's nearest rule)
    lab9 = blocky_labels(g, 2, 96, 160, 8, [0, 1, 2, 3, 4])[:, :90, :150].contiguous()
    feats9 = [torch.randn(2, 8, 90 // 4, 150 // 4, generator=g), torch.randn(2, 8, 90 // 8, 150 // 8, generator=g)]
    # scale = W_label // W_feat: 150 // 37 = 4, 150 // 18 = 8 -> (90//4, 150//4) = (22, 37), (11, 18)
    cfg9 = dict(base, scales=2, weights=[1.0, 1.0], cross_scale_contrast=True)
    save(os.path.join(args.out, "G9_odd_sizes.npz"), cfg9, lab9, feats9, 2, run_ms(cfg9, lab9, feats9, 2))

    # ---- G6: LossWrapper aggregation (CE with Cityscapes weights + ignore, DCV2_ms) and TwoScaleLoss
    lab6 = blocky_labels(g, 2, 64, 128, 16, [0, 1, 2, 5, 19], p=[.3, .25, .2, .15, .1])
    feats6 = [torch.randn(2, 16, 16, 32, generator=g), torch.randn(2, 16, 8, 16, generator=g)]
    # logits are stored as fp16 (values exactly representable) to keep the fixture small
    logits = torch.randn(2, 19, 64, 128, generator=g).half().float()
    interm = torch.randn(2, 19, 64, 128, generator=g).half().float()
    lw_cfg = dict(base, scales=2, weights=[1.0, 0.5], cross_scale_contrast=True, name="LossWrapper",
                  losses={"CrossEntropyLoss": 1, "DenseContrastiveLossV2_ms": 0.1}, device="cpu")
    builtins.print = lambda *a, **k: None
    lw = LossWrapper(json.loads(json.dumps(lw_cfg)))
    builtins.print = _print
    lg = logits.clone().requires_grad_(True)
    fs = [f.clone().requires_grad_(True) for f in feats6]
    torch.manual_seed(21)
    total = lw(lg, lab6.long(), deep_features=fs)
    total.backward()
    d6 = {"total": np.float32(total.item()), "logits_f16": logits.half().numpy(),
          "logits_grad_sample": lg.grad[:, :, ::4, ::4].contiguous().numpy(),
          "logits_grad_abs_sum": np.float64(lg.grad.double().abs().sum().item())}
    for k, v in lw.loss_vals.items():
        d6["val__" + k] = np.float32(v.item() if torch.is_tensor(v) else v)
    for s, f in enumerate(fs):
        d6[f"s{s}_grad"] = f.grad.numpy()
    ts_cfg = dict(base, name="LossWrapper", device="cpu",
                  losses={"TwoScaleLoss": 1.0},
                  interm={"name": "CrossEntropyLoss", "weight": 0.4, "args": []},
                  final={"name": "CrossEntropyLoss", "weight": 1.0, "args": []})
    builtins.print = lambda *a, **k: None
    lw2 = LossWrapper(json.loads(json.dumps(ts_cfg)))
    builtins.print = _print
    tot2 = lw2(logits, lab6.long(), interm_prediction=interm)
    d6["twoscale_total"] = np.float32(tot2.item())
    d6["interm_f16"] = interm.half().numpy()
    d6["twoscale_config_json"] = np.array(json.dumps(ts_cfg))
    save(os.path.join(args.out, "G6_losswrapper.npz"), lw_cfg, lab6, feats6, 21, d6)


if __name__ == "__main__":
    main()
