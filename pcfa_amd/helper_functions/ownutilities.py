"""Model adapter ("plugin API") of the PCFA hot path.

Mirror of the model half of the reference's helper_functions/ownutilities.py:
    InputPadder :21-62   import_and_load :64-169   preprocess_img :241-280
    postprocess_flow :283-299   compute_flow :302-343   model_takes_unit_input :347-360
Same names, argument meaning and error behaviour.  Deliberate differences:
  * postprocess_flow keeps the flow on its device (the reference's `.cpu()`, :297, forces a
    device sync + PCIe round trip on every forward);
  * RAFT / GMA are not wrapped in torch.nn.DataParallel (one process per GPU); the "module."
    prefix of the public checkpoints is stripped on load;
  * `weights=` keyword: "pretrained" (default, the reference's files under Paths.config("weights"))
    or "random:<seed>" for the synthetic benchmark / tests where no checkpoints exist.
"""
import json
import os
from argparse import Namespace

import torch
import torch.nn.functional as F

from .config_paths import Paths

RAFT_CONFIG = {"epsilon": 1e-8, "small": False, "mixed_precision": False, "alternate_correlation": False}
GMA_CONFIG = {"epsilon": 1e-8, "num_heads": 1, "small": False, "mixed_precision": True,
              "alternate_correlation": False, "position_only": False, "position_and_content": False}


class InputPadder:
    """Pads images such that dimensions are divisible by `divisor` (ownutilities.py:21-62)."""

    def __init__(self, dims, divisor=8, mode='sintel'):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // divisor) + 1) * divisor - self.ht) % divisor
        pad_wd = (((self.wd // divisor) + 1) * divisor - self.wd) % divisor
        if mode == 'sintel':
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode='replicate') for x in inputs]

    def get_dimensions(self):
        return self.ht, self.wd

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]


def _strip_module_prefix(state):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}


def _seed_from(weights):
    if isinstance(weights, str) and weights.startswith("random:"):
        return int(weights.split(":", 1)[1])
    return None


def build_network(net, weights="pretrained", device=torch.device("cpu")):
    """Construct flow network `net` and fill its parameters (checkpoint or seeded random)."""
    seed = _seed_from(weights)
    if seed is None and weights != "pretrained":
        raise ValueError("weights must be 'pretrained' or 'random:<seed>', got %r" % (weights,))
    wdir = Paths.config("weights")
    if seed is not None:
        rng_state = torch.get_rng_state()
        torch.manual_seed(seed)
    try:
        if net == 'RAFT':
            from ..nets.raft import RAFT
            model = RAFT(dict(RAFT_CONFIG))
            if seed is None:
                state = torch.load(os.path.join(wdir, 'raft-sintel.pth'), map_location=device)
                model.load_state_dict(_strip_module_prefix(state))
        elif net == 'GMA':
            from ..nets.gma import RAFTGMA
            model = RAFTGMA(Namespace(**GMA_CONFIG))
            if seed is None:
                state = torch.load(os.path.join(wdir, 'gma-sintel.pth'), map_location=device)
                model.load_state_dict(_strip_module_prefix(state))
            else:
                # the aggregation branch is multiplied by gamma, which initialises to 0 (gma.py:95)
                model.update_block.aggregator.gamma.data.fill_(0.5)
        elif net == 'PWCNet':
            from ..nets.pwcnet import PWCDCNet
            model = PWCDCNet()
            if seed is None:
                state = torch.load(os.path.join(wdir, 'pwc_net_chairs.pth.tar'), map_location=device)
                model.load_state_dict(state['state_dict'] if 'state_dict' in state.keys() else state)
        elif net == 'SpyNet':
            from ..nets.spynet import Network as SpyNet
            model = SpyNet(nlevels=6, pretrained=seed is None)
            if seed is None:
                model.load_pretrained(os.path.join(wdir, 'spynet_weights'))
            else:
                for p in model.parameters():
                    p.data.normal_(0.0, 0.02)
        elif net == 'FlowNet2':
            from ..nets.flownet2 import FlowNet2
            # configuration hard-coded by the reference: fp16=False, rgb_max=255, div_flow=20, batchNorm=False
            model = FlowNet2(rgb_max=255.0, div_flow=20.)
            if seed is None:
                state = torch.load(os.path.join(wdir, 'FlowNet2_checkpoint.pth.tar'), map_location=device)
                model.load_state_dict(state['state_dict'])
        else:
            raise RuntimeWarning('The network %s is not a valid model option for import_and_load(network). '
                                 'No model was loaded. Use "RAFT", "GMA", "FlowNetC", "PWCNet" or "SpyNet" instead.'
                                 % (net))
    finally:
        if seed is not None:
            torch.set_rng_state(rng_state)
    return model.to(device)


def import_and_load(net='RAFT', make_unit_input=False, variable_change=False, device=torch.device("cpu"),
                    make_scaled_input_model=False, **kwargs):
    """Import a model and load weights for it (ownutilities.py:64-169).  Extra keyword of this build: `config`, a frozen
    pcfa_amd.config.Config (default: config.DEFAULT) attached to every sub-module -- the build switches of the network."""
    from ..config import attach
    if make_unit_input or variable_change or make_scaled_input_model:
        from .own_models import ScaledInputModel
        model = ScaledInputModel(net, make_unit_input=make_unit_input, variable_change=variable_change,
                                 device=device, **kwargs)
        print("--> transforming model to 'make_unit_input'=%s, 'variable_change'=%s\n"
              % (str(make_unit_input), str(variable_change)))
        return attach(model, kwargs.get("config"))
    try:
        model = attach(build_network(net, weights=kwargs.get("weights", "pretrained"), device=device), kwargs.get("config"))
    except FileNotFoundError as e:
        print("\nLoading the model failed, because the checkpoint path was invalid. Are the checkpoints placed in "
              "%s? The full error that caused the loading failure is below:\n\n%s" % (Paths.config("weights"), e))
        raise SystemExit(1)
    print("--> flow network is set to %s" % net)
    return model


def preprocess_img(network, *images):
    """Pad (and for PWCNet/SpyNet/FlowNet rescale) the inputs (ownutilities.py:241-280)."""
    if network == 'RAFT' or network == "GMA":
        padder = InputPadder(images[0].shape)
        output = padder.pad(*images)
    elif network in ('PWCNet', 'SpyNet'):
        images = [(img / 255.) for img in images]
        padder = InputPadder(images[0].shape, divisor=64)
        output = padder.pad(*images)
    elif network[:7] == 'FlowNet':
        if not network[:8] == 'FlowNet2':
            images = [img / 255. for img in images]
        padder = InputPadder(images[0].shape, divisor=64)
        output = padder.pad(*images)
    else:
        padder = None
        output = images
    return padder, output


def postprocess_flow(network, padder, *flows):
    """Remove the padding (ownutilities.py:283-299); the flow stays on its device."""
    if padder is not None:
        return [padder.unpad(flow) for flow in flows]
    return flows


def compute_flow(model, network, x1, x2, test_mode=True, **kwargs):
    """Forward pass dispatch by network name (ownutilities.py:302-343)."""
    if network == "scaled_input_model":
        flow = model(x1, x2, test_mode=True, **kwargs)
    elif network == 'RAFT':
        _, flow = model(x1, x2, test_mode=test_mode, **kwargs)
    elif network == 'GMA':
        _, flow = model(x1, x2, iters=6, test_mode=test_mode, **kwargs)
    elif network[:7] == 'FlowNet':
        # [B, 3, 2, H, W]; FlowNet2 takes [0,255] images and removes the mean itself (FlowNet2.py:116-118)
        x = torch.stack((x1, x2), dim=-3)
        if not network[:8] == 'FlowNet2':
            rgb_mean = x.contiguous().view(x.size()[:2] + (-1,)).mean(dim=-1).view(x.size()[:2] + (1, 1, 1,)).detach()
            x = x - rgb_mean
        flow = model(x)
    else:  # PWCNet, SpyNet
        flow = model(x1, x2, **kwargs)
    return flow


def model_takes_unit_input(model):
    """True for networks fed with [0,1] images (ownutilities.py:347-360)."""
    return model in ["PWCNet", "SpyNet"]


def torchfloat_to_float64(torch_float):
    """ownutilities.py:507-518."""
    return float(torch_float.detach().cpu().numpy().astype('float64'))


def maximum_flow(flow):
    """Largest flow magnitude of a [.,2,H,W] / [2,H,W] field (ownutilities.py:486-504)."""
    f = flow.detach()
    if f.dim() == 4:
        mag = torch.sqrt(torch.sum(f * f, dim=1))
    else:
        mag = torch.sqrt(torch.sum(f * f, dim=0))
    return torchfloat_to_float64(torch.max(mag))
