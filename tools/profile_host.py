"""Host-side cProfile of the loss (run on the GPU box): where do the non-kernel milliseconds go?"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd.losses import DenseContrastiveLossV2_ms
from mscs_amd.utils import set_verbosity

set_verbosity(40)
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
n, H, W = 12, 512, 1024
label = torch.randint(0, 20, (n, H, W), generator=gen).to(dev)
feats = [torch.randn(n, 256, H // (4 << s), W // (4 << s), generator=gen).to(dev).requires_grad_(True) for s in range(3)]
mod = DenseContrastiveLossV2_ms({"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 3,
                                 "weights": [1.0, 0.7, 0.4], "cross_scale_contrast": True})


def step():
    for f in feats:
        f.grad = None
    loss = mod(label, feats)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
# forward-only host time (no sync at the end)
t0 = time.perf_counter()
for _ in range(10):
    loss = mod(label, feats)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("fwd host ms (incl. its one sync)", (t1 - t0) * 100)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
st.sort_stats("tottime").print_stats(25)
