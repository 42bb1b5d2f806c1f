"""Attack targets (reference helper_functions/targets.py:9-114)."""
import numpy as np
import torch
import torch.nn.functional as F


def zero_flow(flow):
    return torch.zeros_like(flow)


def neg_flow(flow):
    return -flow


def custom_target(flow, path_to_custom_target, device):
    """Load a flow field from a .npy ([H,W,2]) or Middlebury .flo file and crop / reflect-pad it to `flow`
    (targets.py:33-86; the reference reads more formats through frame_utils.read_gen)."""
    try:
        if path_to_custom_target.endswith(".npy"):
            data = np.load(path_to_custom_target)
        elif path_to_custom_target.endswith(".flo"):
            with open(path_to_custom_target, "rb") as f:
                magic = np.fromfile(f, np.float32, count=1)
                if magic.size != 1 or magic[0] != 202021.25:
                    raise AssertionError()
                w, h = (int(v) for v in np.fromfile(f, np.int32, count=2))
                data = np.fromfile(f, np.float32, count=2 * w * h).reshape(h, w, 2)
        else:
            raise AssertionError()
        if len(data) < 2:
            raise AssertionError()
        target = torch.from_numpy(np.array(data).astype(np.float32)).permute(2, 0, 1).float().to(device)
        ts, fs = target.size(), flow.size()
        if fs[-1] < ts[-1]:
            target = target[:, :, :fs[-1]]
        elif fs[-1] > ts[-1]:
            target = F.pad(target, (0, fs[-1] - ts[-1]), "reflect")
        if fs[-2] < ts[-2]:
            target = target[:, :fs[-2], :]
        elif fs[-2] > ts[-2]:
            target = F.pad(target, (0, 0, 0, fs[-2] - ts[-2]), "reflect")
        if len(fs) == 4:
            target = target.unsqueeze(0).repeat(fs[0], 1, 1, 1)
    except (AssertionError, OSError, ValueError):
        print("WARNING: The specified custom target file is not a valid flow file at %s" % path_to_custom_target)
        print("Please specify a valid flow file via --custom_target_path")
        print("\nExiting attack.")
        raise SystemExit(1)
    return target


def get_target(target_name, flow_pred_init, custom_target_path="", device=None):
    """targets.py:89-114."""
    if target_name == 'zero':
        return zero_flow(flow_pred_init)
    if target_name == 'neg_flow':
        return neg_flow(flow_pred_init)
    if target_name == 'custom':
        return custom_target(flow_pred_init, custom_target_path, device)
    raise ValueError('The specified target type "' + target_name + '" is not defined and cannot be used. '
                     'Select one of "zero", "neg_flow" or "custom". Aborting.')
