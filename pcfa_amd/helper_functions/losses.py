"""PCFA loss terms (reference helper_functions/losses.py) on the fused HIP kernels.

Same function names and signatures as the reference module; the arithmetic runs
in pcfa_amd/csrc/attack_math.hip through :mod:`pcfa_amd.ops`.
"""
from .. import ops


def avg_epe(flow1, flow2):
    """Average endpoint error, losses.py:3-30 (as a metric; the differentiable use is loss_delta_constraint)."""
    return ops.get().avg_epe(flow1, flow2)


def avg_mse(flow1, flow2):
    """Mean squared error between two flow fields, losses.py:32-44 (differentiable)."""
    return ops.get().get_loss("mse", flow1, flow2)


def f_epe(pred, target):
    """losses.py:47-58."""
    return avg_epe(pred, target)


def f_mse(pred, target):
    """losses.py:61-73."""
    return avg_mse(pred, target)


def f_cosim(pred, target):
    """losses.py:76-88 -- bug-compatible: 1 - (p.t / sqrt(p.p)) * sqrt(t.t) (the reference multiplies by |t|)."""
    return ops.get().get_loss("cosim", pred, target)


def two_norm_avg_delta_squared(delta1, delta2):
    """losses.py:110-126."""
    return ops.get().two_norm_avg_delta_squared(delta1, delta2)


def relu_penalty(delta1, delta2, device=None, delta_bound=0.001):
    """relu(mean(delta^2) - delta_bound^2), losses.py:177-197."""
    return ops.get().relu_penalty(delta1, delta2, device, delta_bound)


def two_norm_avg_delta(delta1, delta2):
    """losses.py:91-107."""
    return ops.get().two_norm_avg_delta(delta1, delta2)


def two_norm_avg(x):
    """losses.py:129-142."""
    return ops.get().two_norm_avg(x)


def loss_delta_constraint(pred, target, delta1, delta2, device=None, delta_bound=0.001, mu=100., f_type="aee",
                          batch_sums=None):
    """similarity(pred, target) + mu * relu(mean(delta^2) - delta_bound^2), losses.py:200-230.

    `batch_sums` is not part of the reference signature: the multi-rank universal attack passes it with
    `--loss cosim`, whose three sums run over the global batch (pcfa_amd.sharding.BatchSums)."""
    if batch_sums is None:
        return ops.get().loss_delta_constraint(pred, target, delta1, delta2, device, delta_bound=delta_bound, mu=mu,
                                               f_type=f_type)
    return ops.get().loss_delta_constraint(pred, target, delta1, delta2, device, delta_bound=delta_bound, mu=mu,
                                           f_type=f_type, batch_sums=batch_sums)


def get_loss(f_type, pred, target):
    """Similarity term alone (losses.py:145-174), differentiable -- used by the I-FGSM baseline."""
    if f_type not in ("aee", "mse", "cosim"):
        raise NotImplementedError(
            "The requested loss type %s does not exist. Please choose one of 'aee', 'mse' or 'cosim'" % f_type)
    return ops.get().get_loss(f_type, pred, target)
