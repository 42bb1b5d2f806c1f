"""Helpers shared by the test modules."""
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def t(a, device="cpu"):
    """numpy -> float32 torch tensor"""
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(device)


def rel_l2(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
