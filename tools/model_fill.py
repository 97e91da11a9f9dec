"""Deterministic, name-seeded parameter fill shared by tools/gen_golden_models.py (reference side)
and tests/test_models.py (this repo's models): nothing large needs to be committed."""
import zlib

import torch


def fill_state_dict_(module: torch.nn.Module):
    sd = module.state_dict()
    for key in sorted(sd.keys()):
        t = sd[key]
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
        if key.endswith("num_batches_tracked"):
            t.fill_(7)
        elif key.endswith("running_var"):
            t.copy_(0.5 + torch.rand(t.shape, generator=g))
        elif key.endswith("running_mean"):
            t.copy_(0.1 * torch.randn(t.shape, generator=g))
        elif t.dim() <= 1 and key.endswith("weight"):          # norm scales
            t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
        elif t.dtype.is_floating_point:
            fan = max(1, t[0].numel()) if t.dim() > 1 else 1
            t.copy_(torch.randn(t.shape, generator=g) * (1.0 / fan ** 0.5 if t.dim() > 1 else 0.1))
    module.load_state_dict(sd)
    return module


def model_input(shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
