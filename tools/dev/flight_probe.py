#!/usr/bin/env python3
"""Which of {two pairs in flight, solo on fresh graphs, solo on adopted graphs} give the same bits?  (r05 item 5 diagnosis)
usage: flight_probe.py [HxW] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import attack_PCFA  # noqa: E402


def chk(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())


def main():
    h, w = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "436x1024").split("x"))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    model = bench.load_model("RAFT", dev, True)
    seeds = (501, 502)

    def flight():
        model._pcfa_pair_graphs.clear() if hasattr(model, "_pcfa_pair_graphs") else None
        f = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("RAFT", h, w, dev, seeds[k], use_graph=True, model=model), 2, dev)
        last = f.run(steps)
        return [(chk(a.delta1), chk(a.flow_pred), tuple(last[k])) for k, a in enumerate(f.attacks)]

    def solo(seed, fresh):
        if fresh and hasattr(model, "_pcfa_pair_graphs"):
            model._pcfa_pair_graphs.clear()
        st = bench.AttackStepper("RAFT", h, w, dev, seed, use_graph=True, model=model)
        for _ in range(steps):
            last = st.step()
        return (chk(st.delta1), chk(st.flow_pred), tuple(last)), st.graphs_reused

    def interleaved(use_graph):
        """both lanes alive, stepped ALTERNATELY from this one thread (own streams + lanes): logical sharing shows here"""
        from pcfa_amd import ops
        model._pcfa_pair_graphs.clear() if hasattr(model, "_pcfa_pair_graphs") else None
        f = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("RAFT", h, w, dev, seeds[k], use_graph=use_graph, model=model), 2, dev)
        last = [None, None]
        for _ in range(steps):
            for k in range(2):
                with ops.core.lane(k), torch.cuda.stream(f.streams[k]):
                    last[k] = f.attacks[k].step()
        torch.cuda.synchronize()
        return [(chk(a.delta1), chk(a.flow_pred), tuple(last[k])) for k, a in enumerate(f.attacks)]

    def flight_eager():
        model._pcfa_pair_graphs.clear() if hasattr(model, "_pcfa_pair_graphs") else None
        f = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("RAFT", h, w, dev, seeds[k], use_graph=False, model=model), 2, dev)
        last = f.run(steps)
        return [(chk(a.delta1), chk(a.flow_pred), tuple(last[k])) for k, a in enumerate(f.attacks)]

    print("interleaved graph", interleaved(True))
    print("interleaved eager", interleaved(False))
    print("threads eager    ", flight_eager())
    print("flight A", flight())
    print("flight B", flight())
    for s in seeds:
        print("solo fresh  ", s, solo(s, True))
        print("solo fresh 2", s, solo(s, True))
        print("solo adopted", s, solo(s, False))


if __name__ == "__main__":
    main()
