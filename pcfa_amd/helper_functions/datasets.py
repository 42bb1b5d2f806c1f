"""Image-pair sources for the attack loop.

The reference reads Sintel / KITTI-15 from disk (helper_functions/datasets.py:51-190,
ownutilities.prepare_dataloader:172-238); that IO is outside the hot path and the datasets are
not available here.  What the attack needs is a DataLoader yielding
(image1 [B,3,H,W] in [0,255], image2, flow_gt [B,2,H,W], valid) -- `SyntheticPairs` provides it
from a seed: a low-pass filtered random image and a copy shifted by
(3,-2) px with additive noise, so the unattacked flow is non-degenerate.
"""
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset

SHIFT_X, SHIFT_Y = 3, -2


def synthetic_pair(seed, height, width):
    g = torch.Generator().manual_seed(int(seed))
    raw = torch.randint(0, 256, (1, 3, height + 16, width + 16), generator=g).float()
    smooth = F.avg_pool2d(F.pad(raw, (4, 4, 4, 4), mode='reflect'), kernel_size=9, stride=1)
    # stretch the blurred image back to a useful dynamic range
    smooth = (smooth - smooth.mean()) * 6.0 + 127.5
    image1 = smooth[0, :, 8:8 + height, 8:8 + width]
    image2 = smooth[0, :, 8 - SHIFT_Y:8 - SHIFT_Y + height, 8 - SHIFT_X:8 - SHIFT_X + width]
    image2 = image2 + 2.0 * torch.randn(image2.shape, generator=g)
    flow = torch.empty(2, height, width)
    flow[0].fill_(float(SHIFT_X))
    flow[1].fill_(float(SHIFT_Y))
    return image1.clamp(0, 255).contiguous(), image2.clamp(0, 255).contiguous(), flow


class SyntheticPairs(Dataset):
    def __init__(self, pairs, height, width, seed0=0):
        self.pairs, self.height, self.width, self.seed0 = pairs, height, width, seed0

    def __len__(self):
        return self.pairs

    def has_groundtruth(self):
        return True

    def __getitem__(self, i):
        image1, image2, flow = synthetic_pair(self.seed0 + i, self.height, self.width)
        return image1, image2, flow, torch.ones(self.height, self.width)


class _Shard(Dataset):
    """Every world-th sample starting at rank (the universal attack's local slice of each global batch)."""

    def __init__(self, base, rank, world):
        self.base, self.rank, self.world = base, rank, world

    def __len__(self):
        return len(self.base) // self.world

    def __getitem__(self, i):
        return self.base[i * self.world + self.rank]


def prepare_dataloader(args, batch_size=1, shuffle=False, shard=None):
    """(DataLoader, has_gt) for args.dataset (ownutilities.py:172-238).

    `shuffle=True` uses a generator seeded identically on every rank, so all ranks walk the same
    global batch order (the reference's unseeded shuffle, attack_PCFA.py:347, is not reproducible).
    """
    if args.dataset != 'Synthetic':
        raise NotImplementedError(
            "Dataset %r: the Sintel/KITTI readers of the reference (helper_functions/datasets.py, frame_utils.py) "
            "are IO outside the accelerated path and are not part of this build; pass --dataset Synthetic or hand "
            "your own DataLoader to attack_l2(args, data_loader=..., has_gt=...)." % args.dataset)
    h, w = (int(v) for v in args.synthetic_size.lower().split("x"))
    n = 32 if args.small_run else args.synthetic_pairs
    ds = SyntheticPairs(n, h, w)
    has_gt = ds.has_groundtruth()
    if shard is not None and shard[1] > 1:
        ds = _Shard(ds, shard[0], shard[1])
    gen = torch.Generator().manual_seed(1234) if shuffle else None
    return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, generator=gen), has_gt
