"""Alias of :mod:`pcfa_amd.ops.hip` (the operator table), kept under the name rounds 1-3 used: `from pcfa_amd import
hip_ops` and `pcfa_amd.ops.get()` hand out the SAME module object."""
import sys

from .ops import hip as _impl

sys.modules[__name__] = _impl
