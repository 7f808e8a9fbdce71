"""Synthetic stand-in for a DAVIS sequence (no dataset ships offline): frames with the layout, value
range and mean subtraction of ``DAVIS2016.__getitem__`` after ``ToTensor`` (reference:
src/dataloaders/davis_2016.py:101-134) - image [3,H,W] fp32 = noisy BGR frame with a brighter elliptical object, minus
the dataset mean; gt [1,H,W] in {0,1} (the ellipse, ~10 % foreground)."""
import torch
from torch.utils.data import Dataset

MEANVAL = (104.00699, 116.66877, 122.67892)  # src/dataloaders/davis_2016.py:28


def make_frame(h: int, w: int, seed: int = 1234, index: int = 0):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed + 7919 * index)
    noise = torch.rand((3, h, w), generator=g)
    yy = torch.arange(h, dtype=torch.float32).view(h, 1)
    xx = torch.arange(w, dtype=torch.float32).view(1, w)
    cy, cx = h * (0.45 + 0.02 * (index % 5)), w * (0.5 - 0.02 * (index % 7))
    gt = ((((yy - cy) / (h * 0.2)) ** 2 + ((xx - cx) / (w * 0.16)) ** 2) <= 1.0).float().unsqueeze(0)
    # brighter, lower-contrast object on a darker noisy background, then the dataset mean comes off
    img = gt * (150.0 + 100.0 * noise) + (1.0 - gt) * (150.0 * noise) - torch.tensor(MEANVAL).view(3, 1, 1)
    return img, gt


class SyntheticSequence(Dataset):
    """mode='train': one annotated frame (as DAVIS2016(train=True, seq_name=...) yields only frame 0);
    mode='test': ``n_frames`` frames of the same sequence."""

    def __init__(self, seq_name: str = 'synthetic', height: int = 480, width: int = 854, n_frames: int = 1,
                 seed: int = 1234):
        self.seq_name, self.h, self.w, self.n, self.seed = seq_name, height, width, n_frames, seed

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        img, gt = make_frame(self.h, self.w, self.seed, idx)
        return {'image': img, 'gt': gt, 'seq_name': self.seq_name, 'fname': '%05d' % idx}
