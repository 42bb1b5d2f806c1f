#!/usr/bin/env python3
"""What happens between two closure replays of an attack step: device kernels launched by the optimiser (not by the
captured graphs), their time, and the idle gaps on the device.  usage: lbfgs_timeline.py [NET]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
    st = bench.AttackStepper(net, 436, 1024, torch.device("cuda", 0), seed=0)
    st.step()
    st.enable_graph()
    for _ in range(int(os.environ.get("STEPS", "11"))):    # history reaches 100 pairs after 10 steps
        st.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        st.step()
        torch.cuda.synchronize()
    evs = sorted((e for e in prof.events() if e.device_type == DeviceType.CUDA), key=lambda e: e.time_range.start)
    # closure kernels: identify graph replays by the long dense runs; simpler: the optimiser's kernels are the ones whose
    # names are in this set
    opt_names = ("lbfgs_", "reduce_kernel", "elementwise", "Memcpy", "copyBuffer", "vectorized", "fill", "dot", "abs")
    t0, t1 = evs[0].time_range.start, evs[-1].time_range.end
    busy = sum(e.time_range.elapsed_us() for e in evs)
    gaps = []
    for a, b in zip(evs, evs[1:]):
        g = b.time_range.start - a.time_range.end
        if g > 15:
            gaps.append((g, a.name[:50], b.name[:50]))
    print("step: %.1f ms wall on the device timeline, %.1f ms busy, %d kernels" % ((t1 - t0) / 1e3, busy / 1e3, len(evs)))
    print("gaps > 15 us: %d, total %.1f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e3))
    agg = collections.Counter()
    for g, a, b in gaps:
        agg[(a, b)] += g
    for (a, b), g in agg.most_common(12):
        print("  %8.1f us  after %-50s before %s" % (g, a, b))
    acc = collections.defaultdict(lambda: [0, 0.0])
    for e in evs:
        if any(k in e.name for k in ("lbfgs", "gram_")) or e.name.startswith("void at::native") or "Memcpy" in e.name:
            a = acc[e.name[:90]]
            a[0] += 1
            a[1] += e.time_range.elapsed_us()
    print("optimiser-side kernels (heuristic: lbfgs_*, at::native::*, memcpy) -- includes the closure's own torch kernels:")
    for k, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %5d x %8.1f us total  %s" % (n, us, k))


if __name__ == "__main__":
    main()
