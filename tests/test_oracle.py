"""The oracle (oracle/dcl_oracle.py + sampling_oracle.c) against golden vectors produced by
running the reference itself (tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, num_classes_for

MS_CASES = golden_names(["G2", "G4", "G5", "G9"])
SINGLE_CASES = golden_names(["G1_", "G1b_", "G3"])


def _cfg(orc, c, K):
    return orc.LossConfig(
        num_all_classes=K,
        temperature=c.get("temperature", 0.5),
        min_views_per_class=c.get("min_views_per_class", 5),
        max_views_per_class=c.get("max_views_per_class", 2500),
        max_features_total=c.get("max_features_total", 10000),
        scales=c.get("scales", 2),
        weights=c.get("weights"),
        cross_scale_contrast=c.get("cross_scale_contrast", False),
        has_cross_scale_temperature_key="cross_scale_temperature" in c,
        detach_deepest=c.get("detach_deepest", False),
        w_high_low=c.get("w_high_low", 1.0),
        w_high_mid=c.get("w_high_mid", 1.0),
    )


def _check_plan(g, s, plan):
    assert plan.V == int(g[f"s{s}_V"])
    np.testing.assert_array_equal(plan.pair_b, g[f"s{s}_pair_b"])
    np.testing.assert_array_equal(plan.pair_k, g[f"s{s}_pair_k"])
    np.testing.assert_array_equal(plan.pix, g[f"s{s}_pix"])          # bit-exact sampling
    assert plan.log_this_step == bool(g[f"s{s}_log_this_step"])


@pytest.mark.parametrize("name", SINGLE_CASES)
def test_single_scale_matches_reference(oracle, name):
    g = load_golden(name)
    c = g["config"]
    cfg = _cfg(oracle, c, num_classes_for(c))
    rng = oracle.MT19937(int(g["seed"]))
    loss, plan, dfeat = oracle.dcv2_single(g["label"].astype(np.int64), g["feat0"], cfg, rng=rng)
    _check_plan(g, 0, plan)
    np.testing.assert_allclose(loss, g["loss"], rtol=2e-6)
    scale = np.abs(g["s0_grad"]).max()
    np.testing.assert_allclose(dfeat, g["s0_grad"], atol=2e-6 * scale, rtol=1e-4)
    if "sampled_features" in g:        # DCV2 4-tuple layout [T, C, V]
        X = oracle.gather_bank(g["feat0"], plan)
        np.testing.assert_array_equal(X, g["sampled_features"])
        np.testing.assert_array_equal(plan.pair_k.astype(np.float32), g["sampled_labels"])


@pytest.mark.parametrize("name", MS_CASES)
def test_ms_matches_reference(oracle, name):
    g = load_golden(name)
    c = g["config"]
    cfg = _cfg(oracle, c, num_classes_for(c))
    feats = [g[f"feat{s}"] for s in range(cfg.scales)]
    rng = oracle.MT19937(int(g["seed"]))
    res = oracle.dcv2_ms(g["label"].astype(np.int64), feats, cfg, rng=rng)
    for s in range(cfg.scales):
        _check_plan(g, s, res.plans[s])
    np.testing.assert_allclose(res.ms_losses, g["ms_losses"], rtol=2e-6)
    assert len(res.cs_losses) == len(g["cs_losses"])
    np.testing.assert_allclose(res.cs_losses, g["cs_losses"], rtol=2e-6)
    np.testing.assert_allclose(res.loss, g["loss"], rtol=2e-6)
    for s in range(cfg.scales):
        ref = g[f"s{s}_grad"]
        scale = max(np.abs(ref).max(), 1e-30)
        np.testing.assert_allclose(res.grads[s], ref, atol=3e-6 * scale, rtol=1e-4)


def test_oracle_fp32_close_to_fp64(oracle):
    g = load_golden("G4_cross_zero_pos")
    c = g["config"]
    cfg = _cfg(oracle, c, 20)
    feats = [g[f"feat{s}"] for s in range(cfg.scales)]
    r64 = oracle.dcv2_ms(g["label"].astype(np.int64), feats, cfg, rng=oracle.MT19937(int(g["seed"])))
    r32 = oracle.dcv2_ms(g["label"].astype(np.int64), feats, cfg, rng=oracle.MT19937(int(g["seed"])),
                         dtype=np.float32)
    np.testing.assert_allclose(r32.loss, r64.loss, rtol=1e-5)


def test_randperm_restatement_matches_torch(oracle):
    """sampling_oracle.c's MT19937 + Fisher-Yates == torch.randperm on the CPU default generator."""
    for seed in (0, 1, 12345, 2**31 + 7):
        torch.manual_seed(seed)
        rng = oracle.MT19937(seed)
        for n in (1, 2, 5, 43, 1638, 0, 7, 32768):
            np.testing.assert_array_equal(rng.randperm(n), torch.randperm(n).numpy())


def test_rng_order_pin_G8(oracle):
    """Sequence of randperm lengths / heads consumed by the reference in the G2 run."""
    g8 = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "G8_rng_order.npz"))
    g2 = load_golden("G2_ms4_cross")
    rng = oracle.MT19937(int(g8["seed"]))
    lens = []
    for s in range(4):
        lbl = oracle.downsample_labels(g2["label"].astype(np.int64), 4 * 2 ** s)
        counts = oracle.class_counts(lbl, 20)
        for b, k in zip(g2[f"s{s}_pair_b"], g2[f"s{s}_pair_k"]):
            lens.append(int(counts[b, k]))
    np.testing.assert_array_equal(lens, g8["randperm_n"])
    for n, head in zip(g8["randperm_n"], g8["randperm_first8"]):
        p = rng.randperm(int(n))
        m = min(8, int(n))
        np.testing.assert_array_equal(p[:m], head[:m])


def test_randperm_callable_path_equals_rng_path(oracle):
    g = load_golden("G3b_maxviews7")
    lab = g["label"].astype(np.int64)
    a = oracle.make_plan(lab, 4, 20, 5, 7, 10000, rng=oracle.MT19937(3))
    torch.manual_seed(3)
    b = oracle.make_plan(lab, 4, 20, 5, 7, 10000, randperm=lambda n: torch.randperm(n).numpy())
    np.testing.assert_array_equal(a.pix, b.pix)


def test_no_pairs_raises(oracle):
    lab = np.full((1, 16, 16), 19, dtype=np.int64)     # only the (dropped) last class present
    with pytest.raises(RuntimeError):
        oracle.make_plan(lab, 4, 20, 5, 2500, 10000, rng=oracle.MT19937(0))


@pytest.mark.parametrize("name", ["G2_ms4_cross", "G5a_detach", "G5e_S3", "G4_cross_zero_pos"])
def test_eager_torch_restatement_matches_reference(name):
    """oracle/eager_torch.py (the eager-structure comparator used by bench.py) against the goldens."""
    from oracle import eager_torch
    g = load_golden(name)
    c = g["config"]
    S = c["scales"]
    feats = [torch.from_numpy(g[f"feat{s}"]).requires_grad_(True) for s in range(S)]
    label = torch.from_numpy(g["label"].astype(np.int64))
    torch.manual_seed(int(g["seed"]))
    total, ms, cs = eager_torch.dcv2_ms(
        label, feats, 20, c["temperature"], c["weights"], cross=c.get("cross_scale_contrast", False),
        cross_tau=0.1 if "cross_scale_temperature" in c else None,
        detach_deepest=c.get("detach_deepest", False), w_high_low=c.get("w_high_low", 1.0),
        w_high_mid=c.get("w_high_mid", 1.0))
    total.backward()
    np.testing.assert_allclose(total.item(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose([x.item() for x in ms], g["ms_losses"], rtol=1e-6)
    np.testing.assert_allclose([x.item() for x in cs], g["cs_losses"], rtol=1e-6)
    for s in range(S):
        got = feats[s].grad.numpy() if feats[s].grad is not None else np.zeros_like(g[f"s{s}_grad"])
        np.testing.assert_allclose(got, g[f"s{s}_grad"], atol=1e-6 * np.abs(g[f"s{s}_grad"]).max() + 1e-12)


def test_global_bank_emulation_reduces_to_reference_at_world_1(oracle):
    """dcv2_ms_global with one rank == the reference-pinned dcv2_ms (pin (1) of SURVEY section 8 row e)."""
    g = load_golden("G2_ms4_cross")
    c = g["config"]
    cfg = _cfg(oracle, c, 20)
    feats = [g[f"feat{s}"] for s in range(cfg.scales)]
    lab = g["label"].astype(np.int64)
    a = oracle.dcv2_ms(lab, feats, cfg, rng=oracle.MT19937(int(g["seed"])))
    b = oracle.dcv2_ms_global([lab], [feats], cfg, [int(g["seed"])], rank=0)
    np.testing.assert_allclose(b.loss, a.loss, rtol=1e-12)
    for x, y in zip(a.grads, b.grads):
        np.testing.assert_allclose(y, x, atol=1e-12 * max(1.0, np.abs(x).max()))
    np.testing.assert_allclose(b.loss, g["loss"], rtol=2e-6)
