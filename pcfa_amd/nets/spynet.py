"""SpyNet (Ranjan & Black 2017) for the PCFA hot path.

Behavioural reference (cv-stuttgart/PCFA): models/SpyNet/SpyNet.py
    :19-54 Preprocess   :56-84 Basic   :86-102 Backward (warp)   :104-158 Network
SpyNet has no correlation operator: convolutions, average pooling, bilinear
upsampling and grid_sample only (all stock PyTorch-ROCm); it is the plumbing
configuration of BASELINE.json (config 0).

Unlike the reference, constructing the network does not read 60 weight files
from disk; `load_pretrained(dir)` does that on request with the reference's file
naming (SpyNet.py:75-81).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class Preprocess(nn.Module):
    def __init__(self, pre_normalization=None):
        super().__init__()
        self.pre_normalization = pre_normalization

    def forward(self, x):
        red, green, blue = x[:, 0:1], x[:, 1:2], x[:, 2:3]
        pn = self.pre_normalization
        if pn is not None:
            if hasattr(pn, 'mean') and hasattr(pn, 'std'):
                mean, std = pn.mean, pn.std
            else:
                flat = x.transpose(0, 1).contiguous().view(3, -1)
                mean, std = flat.mean(1), flat.std(1)
            red, green, blue = red * std[0] + mean[0], green * std[1] + mean[1], blue * std[2] + mean[2]
        red = (red - 0.485) / 0.229
        green = (green - 0.456) / 0.224
        blue = (blue - 0.406) / 0.225
        return torch.cat([red, green, blue], 1)


class Basic(nn.Module):
    def __init__(self, intLevel):
        super().__init__()
        self.intLevel = intLevel
        self.moduleBasic = nn.Sequential(
            nn.Conv2d(8, 32, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(32, 64, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(64, 32, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(32, 16, kernel_size=7, stride=1, padding=3), nn.ReLU(inplace=False),
            nn.Conv2d(16, 2, kernel_size=7, stride=1, padding=3))

    def forward(self, x):
        return self.moduleBasic(x)


def backward_warp(feat, flow):
    """Warp `feat` by `flow` with the grid clamped to [-1,1] (SpyNet.py:90-102)."""
    B, _, H, W = feat.shape
    hor = torch.linspace(-1.0, 1.0, W, device=feat.device).view(1, 1, 1, W).expand(B, 1, H, W)
    ver = torch.linspace(-1.0, 1.0, H, device=feat.device).view(1, 1, H, 1).expand(B, 1, H, W)
    grid = torch.cat([hor, ver], 1)
    flow = torch.cat([flow[:, 0:1] / ((W - 1.0) / 2.0), flow[:, 1:2] / ((H - 1.0) / 2.0)], 1)
    grid = (grid + flow).clamp(-1.0, 1.0).permute(0, 2, 3, 1)
    return F.grid_sample(input=feat, grid=grid, mode='bilinear', align_corners=False)


class Network(nn.Module):
    def __init__(self, nlevels=6, strmodel='F', pre_normalization=None, pretrained=False):
        super().__init__()
        self.nlevels = nlevels
        self.strmodel = strmodel
        self.pre_normalization = pre_normalization
        self.modulePreprocess = Preprocess(pre_normalization)
        self.moduleBasic = nn.ModuleList([Basic(l) for l in range(nlevels)])
        if not pretrained:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    if m.bias is not None:
                        nn.init.uniform_(m.bias)
                    nn.init.xavier_uniform_(m.weight)

    def load_pretrained(self, weights_dir):
        """weights_dir/modelL{level+1}_{strmodel}-{conv}-{weight|bias}.pth.tar (SpyNet.py:75-81)."""
        for lvl, basic in enumerate(self.moduleBasic):
            file_level = lvl
            if lvl == 5 and self.strmodel in ('3', '4'):
                file_level = 4  # chairs-trained variants ship no sixth level
            for i in range(5):
                stem = os.path.join(weights_dir, "modelL%d_%s-%d-" % (file_level + 1, self.strmodel, i + 1))
                basic.moduleBasic[i * 2].weight.data.copy_(torch.load(stem + "weight.pth.tar"))
                basic.moduleBasic[i * 2].bias.data.copy_(torch.load(stem + "bias.pth.tar"))

    def forward(self, first, second):
        firsts = [self.modulePreprocess(first)]
        seconds = [self.modulePreprocess(second)]
        for _ in range(self.nlevels - 1):
            firsts.insert(0, F.avg_pool2d(firsts[0], kernel_size=2, stride=2))
            seconds.insert(0, F.avg_pool2d(seconds[0], kernel_size=2, stride=2))

        flow = torch.zeros(firsts[0].size(0), 2, int(math.floor(firsts[0].size(2) / 2.0)),
                           int(math.floor(firsts[0].size(3) / 2.0)), device=first.device, dtype=first.dtype)
        all_flows = [None] * self.nlevels
        for lvl in range(len(firsts)):
            up = F.interpolate(flow, scale_factor=2, mode='bilinear', align_corners=False) * 2.0
            if up.size(2) != firsts[lvl].size(2):
                up = F.pad(up, [0, 0, 0, 1], 'replicate')
            if up.size(3) != firsts[lvl].size(3):
                up = F.pad(up, [0, 1, 0, 0], 'replicate')
            flow = self.moduleBasic[lvl](torch.cat([firsts[lvl], backward_warp(seconds[lvl], up), up], 1)) + up
            all_flows[self.nlevels - lvl - 1] = flow
        return all_flows if self.training else flow
