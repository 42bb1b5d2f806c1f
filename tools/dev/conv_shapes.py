#!/usr/bin/env python3
"""Per-shape table of the hand-written convolution launches of one closure (eager): entry point, integer arguments,
calls, mean device time between two events around the launch, and the dense-equivalent TFLOP/s.
usage: conv_shapes.py [NET] [HxW]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import hip_ops  # noqa: E402

WATCH = ("pcfa_conv3x3_act_fwd", "pcfa_conv3x3_fwd", "pcfa_conv3x3_act_fwd_pair", "pcfa_conv3x3_masked_fwd",
         "pcfa_sepconv5_fwd_split", "pcfa_sepconv5_fwd_split_masked", "pcfa_sepconv5_fwd", "pcfa_conv3x3_fewout_fwd",
         "pcfa_conv3x3_fewout_bwd", "pcfa_conv_fewin_fwd")


def main():
    net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
    h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "436x1024").split("x"))
    st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0)
    for _ in range(2):
        st.optimizer.zero_grad()
        st._closure_body()
    torch.cuda.synchronize()
    log = []
    def spy(name, args, invoke):
        if name not in WATCH:
            return invoke(name, *args)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        invoke(name, *args)
        e.record()
        log.append((name, tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < 1 << 20), s, e))

    hip_ops.set_call_spy(spy)      # ops.core hook: the operator modules bind _call by name, patching the table does nothing
    try:
        st.optimizer.zero_grad()
        st._closure_body()
        torch.cuda.synchronize()
    finally:
        hip_ops.set_call_spy(None)
    assert log, "the spy saw no launch: WATCH names no entry point of this network"
    acc = collections.OrderedDict()
    for name, ints, s, e in log:
        a = acc.setdefault((name, ints), [0, 0.0])
        a[0] += 1
        a[1] += s.elapsed_time(e) * 1e3
    print("%-26s %-34s %5s %9s %9s" % ("entry point", "int args", "calls", "mean us", "total us"))
    tot = 0.0
    for (name, ints), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-26s %-34s %5d %9.1f %9.1f" % (name, ints, n, us / n, us))
        tot += us
    print("total %.1f us (event pairs include ~7 us of bracket overhead per launch)" % tot)


if __name__ == "__main__":
    main()
