"""Per-launch timing helpers used by bench.py and tools/ (off by default): event brackets around C-ABI calls and HIP
events attached to the dispatch packet itself."""
import ctypes

import torch

from .. import _hip


class LaunchProfiler:
    """Optional per-launch timing: HIP events recorded on the launch stream around every C-ABI call.
    Used by bench.py to measure kernel durations live inside the timed region (off by default)."""

    def __init__(self, names=None):
        self.names = set(names) if names else None
        self.events = {}

    def wants(self, name):
        return self.names is None or name in self.names

    def summary(self):
        """name -> (mean microseconds, launches); synchronises the device."""
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.events.items():
            tot = sum(s.elapsed_time(e) for s, e in pairs)
            out[name] = (1e3 * tot / len(pairs), len(pairs))
        return out


class DispatchTimer:
    """Kernel durations from hipEvents attached to the dispatch packet itself (pcfa_timing_arm ->
    hipExtLaunchKernel).  Unlike an event bracket around a launch, which inserts two barrier packets
    (measured: +4..7 us per launch on MI355X), these events carry the packet's own begin/end timestamps --
    the same source rocprofv3's kernel trace reads.  Used by bench.py for the roofline figures.

    `plan` maps a C-ABI entry point to [(label, nth kernel it launches)]; see include/pcfa_hip.h for the
    launch order of the multi-kernel entry points."""

    DEFAULT_PLAN = {
        "pcfa_corr_lookup_fwd": [("corr_lookup_fwd", 0)],
        "pcfa_corr_lookup_bwd": [("corr_lookup_bwd", 0)],
        "pcfa_corr_pyramid_fwd": [("corr_pyramid_gemm_fwd", 0)],
        "pcfa_corr_pyramid_bwd": [("corr_pyramid_gemm_dfmap1", 0), ("corr_pyramid_gemm_df2ext", 2)],
        "pcfa_corr_pyramid_bwd_windows": [("corr_pyramid_gemm_dfmap1", 2), ("corr_pyramid_gemm_df2ext", 4)],
        "pcfa_corr_f2ext_fwd": [("corr_f2ext_fwd", 0)],
        "pcfa_spatial_corr_fwd": [("spatial_corr_fwd", 0)],
        "pcfa_spatial_corr_bwd": [("spatial_corr_bwd_in1", 0), ("spatial_corr_bwd_in2", 1)],
        "pcfa_flownet_corr_fwd": [("flownet_corr_fwd", 0)],
        "pcfa_flownet_corr_bwd": [("flownet_corr_bwd_in1", 0), ("flownet_corr_bwd_in2", 1)],
        "pcfa_resample2d_fwd": [("resample2d_fwd", 0)],
        "pcfa_resample2d_bwd": [("resample2d_bwd", 1)],  # kernel 0 clears grad_in1
        "pcfa_channelnorm_fwd": [("channelnorm_fwd", 0)],
        "pcfa_channelnorm_bwd": [("channelnorm_bwd", 0)],
        "pcfa_box_transform_fwd": [("box_transform_fwd", 0)],
        "pcfa_box_transform_bwd": [("box_transform_bwd", 0)],
        "pcfa_flow_loss_fwd": [("flow_loss_partial", 0)],
        "pcfa_gru_gates_fwd": [("gru_gates_fwd", 0)],
        "pcfa_gru_gates_bwd": [("gru_gates_bwd", 0)],
        "pcfa_gru_update_fwd": [("gru_update_fwd", 0)],
        "pcfa_gru_update_bwd": [("gru_update_bwd", 0)],
        "pcfa_conv_fewin_fwd": [("conv_fewin_fwd", 0)],
        "pcfa_pwc_warp_fwd": [("pwc_warp_fwd", 0)],
        "pcfa_pwc_warp_bwd": [("pwc_warp_bwd", 1)],
        "pcfa_pwc_warp_bwd_det": [("pwc_warp_bwd", 1)],
        "pcfa_conv3x3_fewout_fwd": [("conv3x3_fewout_fwd", 0)],
        "pcfa_conv3x3_fewout_bwd": [("conv3x3_fewout_bwd", 0)],
        "pcfa_instnorm_fwd": [("instnorm_stats_fwd", 0), ("instnorm_apply_fwd", 1)],
        "pcfa_instnorm_bwd": [("instnorm_stats_bwd", 0), ("instnorm_apply_bwd", 1)],
        "pcfa_add_relu_fwd": [("add_relu_fwd", 0)],
        "pcfa_bias_relu_fwd": [("bias_relu_fwd", 0)],
        "pcfa_relu_bwd": [("relu_bwd", 0)],
    }

    EVENT_FLAGS = 0x20000000  # hipEventDisableSystemFence

    def __init__(self, plan=None):
        self.plan = dict(self.DEFAULT_PLAN if plan is None else plan)
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.pairs = {}

    def new_pair(self, name):
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        for e in (e0, e1):
            # timing-only events: without hipEventDisableSystemFence the dispatch they ride on ends with a
            # SYSTEM-scope release (write-back of every dirty L2 line, also those of earlier kernels), which a
            # plain or graph-replayed launch does not pay -- rocprofv3 shows the same kernel 1.7 us longer with
            # default events attached (an r01 trace-splitting probe)
            err = self.hip.hipEventCreateWithFlags(ctypes.byref(e), self.EVENT_FLAGS)
            if err != 0:
                raise RuntimeError("hipEventCreateWithFlags failed: %d" % err)
        self.pairs.setdefault(name, []).append((e0, e1))
        return e0, e1

    def summary(self):
        """name -> (mean microseconds, launches); synchronises the device and releases the events."""
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.pairs.items():
            tot, n = 0.0, 0
            for e0, e1 in pairs:
                ms = ctypes.c_float()
                if self.hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1) == 0:  # else: never launched
                    tot += ms.value
                    n += 1
                self.hip.hipEventDestroy(e0)
                self.hip.hipEventDestroy(e1)
            if n:
                out[name] = (1e3 * tot / n, n)
        self.pairs = {}
        # a pair whose kernel never launched (an entry point that took a shorter path: the one-launch instance norm) makes
        # hipEventElapsedTime fail, and HIP keeps that as its sticky "last error": the next launch check would report it
        self.hip.hipGetLastError()
        return out
