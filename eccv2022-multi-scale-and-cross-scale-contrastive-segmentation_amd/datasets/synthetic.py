"""Synthetic segmentation dataset emitting the reference datasets' item tuple
``(img f32 [3, H, W], lbl i32 [H, W], metadata)`` (datasets/Cityscapes.py:185-215) without touching
disk: there is no network / dataset on the benchmark box.  Labels are network ids in [0, K-1] where
K-1 is the ignore id when the experiment has one."""
import torch
from torch.utils.data import Dataset


class SyntheticSegmentation(Dataset):
    def __init__(self, length, size, num_all_classes, mode='iid', block=32, seed=0):
        self.length, self.size, self.K = int(length), tuple(size), int(num_all_classes)
        self.mode, self.block, self.seed = mode, block, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        H, W = self.size
        img = torch.randn(3, H, W, generator=g)
        if self.mode == 'iid':          # worst-case load for the contrastive loss (SURVEY.md row d)
            lbl = torch.randint(0, self.K, (H, W), generator=g, dtype=torch.int32)
        else:                           # blocky layout, closer to real label maps
            b = self.block
            small = torch.randint(0, self.K, ((H + b - 1) // b, (W + b - 1) // b), generator=g, dtype=torch.int32)
            lbl = small.repeat_interleave(b, 0).repeat_interleave(b, 1)[:H, :W].contiguous()
        return img, lbl, {'index': idx, 'target_size': [H, W]}
