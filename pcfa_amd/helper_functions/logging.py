"""Metric helpers + a light metric sink for the attack loop.

The metric *math* mirrors the reference's helper_functions/logging.py:165-262 (thin wrappers
over losses.avg_epe / two_norm_avg); the mlflow sink (logging.py:67-111,343-354) is replaced by
an in-process recorder that can be dumped as JSON lines -- mlflow is out of scope
and its ~15 synchronous log_metric calls per step do not belong in a GPU hot loop.
Metric names are the reference's.
"""
import json
import os

import numpy as np
import torch

from . import losses, ownutilities


class MetricSink:
    """Collects (key, value, step) triples; `dump(path)` writes them as JSON lines."""

    def __init__(self):
        self.records = []
        self.params = {}

    def log_metric(self, key, value, step=None):
        if value is not None:
            self.records.append((key, float(value), step))

    def log_param(self, key, value):
        self.params[key] = value

    def dump(self, path):
        with open(path, "w") as f:
            f.write(json.dumps({"params": {k: str(v) for k, v in self.params.items()}}) + "\n")
            for k, v, s in self.records:
                f.write(json.dumps({"key": k, "value": v, "step": s}) + "\n")


SINK = MetricSink()


def log_metric(key, value, step=None):
    SINK.log_metric(key, value, step)


def log_param(key, value):
    SINK.log_param(key, value)


def log_metrics(step, *key_value_pairs):
    """logging.py:343-354."""
    for key, value in key_value_pairs:
        if value is not None:
            SINK.log_metric(key, value, step)


def calc_log_averages(numsteps, *key_value_pairs):
    """logging.py:357-372: log value/numsteps under `key`."""
    for key, value in key_value_pairs:
        SINK.log_metric(key, value / numsteps if numsteps else float("nan"))


def calc_metrics_adv(flow_pred, target, flow_pred_init):
    """AEE(adv, target), AEE(adv, init) -- logging.py:165-185."""
    return float(losses.avg_epe(flow_pred, target)), float(losses.avg_epe(flow_pred, flow_pred_init))


def calc_metrics_adv_gt(flow_pred, flow_gt):
    """logging.py:188-203."""
    return float(losses.avg_epe(flow_pred, flow_gt))


def calc_metrics_const(target, flow_pred_init):
    """logging.py:206-221."""
    return float(losses.avg_epe(target, flow_pred_init))


def calc_metrics_const_gt(target, flow_pred_init, flow_gt):
    """logging.py:224-246."""
    return float(losses.avg_epe(target, flow_gt)), float(losses.avg_epe(flow_pred_init, flow_gt))


def calc_delta_metrics(delta1, delta2, step=None):
    """logging.py:249-262."""
    l2_delta1 = ownutilities.torchfloat_to_float64(losses.two_norm_avg(delta1))
    l2_delta2 = ownutilities.torchfloat_to_float64(losses.two_norm_avg(delta2))
    l2_delta12 = ownutilities.torchfloat_to_float64(losses.two_norm_avg_delta(delta1, delta2))
    return l2_delta1, l2_delta2, l2_delta12


def create_subfolder(folder_path, name):
    path = os.path.join(folder_path, name)
    os.makedirs(path, exist_ok=True)
    return path


def save_tensor(tens, tensor_name, batch, output_folder, unregistered_artifacts=True):
    """`{batch:05d}_{name}.npy`, fp32 -- the artefact format evaluate_PCFA.py consumes (logging.py:265-286)."""
    if tens is None or output_folder is None:
        return None
    path = os.path.join(output_folder, "%05d_%s.npy" % (batch, tensor_name))
    np.save(path, tens.detach().cpu().numpy().astype(np.float32))
    return path
