"""Operator table used by the networks and the attack loop.

The package ships exactly one implementation, :mod:`pcfa_amd.ops.hip` (HIP
kernels through the C-ABI; the operators live in the sibling modules corr, conv,
gru, gma, pwc, flownet, attack_math, with core / profiling underneath;
``pcfa_amd.hip_ops`` is an alias of the same module).  It raises on non-GPU tensors and when
libpcfa_hip.so is missing -- there is no CPU fallback.

`override_for_testing` exists so that the test-suite can drive the *host*
logic (attack schedule, sharding, model plumbing) on a GPU-less machine by
injecting the CPU oracle from outside the package; nothing in pcfa_amd calls it.
"""
import contextlib

from . import hip as _hip_ops

_active = _hip_ops


def get():
    return _active


@contextlib.contextmanager
def override_for_testing(ops_module):
    global _active
    prev = _active
    _active = ops_module
    try:
        yield ops_module
    finally:
        _active = prev
