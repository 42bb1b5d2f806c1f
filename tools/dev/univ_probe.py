import os, sys, torch
sys.path.insert(0, os.getcwd())
from tests import closure_util
from tests.util import load_golden, rel_l2
from pcfa_amd import attack_PCFA
g = load_golden("universal_raft")
for fused in ("1", "0"):
    for graph in ("1", "0"):
        from pcfa_amd.nets import raft as raft_net
        raft_net.FUSED_LOOKUP = fused == "1"
        os.environ["PCFA_HIP_GRAPH"] = graph
        closure_util._MODELS.clear()
        args, loader = closure_util.universal_case(g)
        res = attack_PCFA.attack_l2_universal(args, data_loader=loader, has_gt=False)
        r1 = rel_l2(res["delta1"].cpu(), torch.from_numpy(g["delta1_b1_t8"]))
        r2 = rel_l2(res["delta2"].cpu(), torch.from_numpy(g["delta2_b1_t8"]))
        n8 = rel_l2(torch.from_numpy(g["delta1_b1_t3"]), torch.from_numpy(g["delta1_b1_t8"]))
        print("fused", fused, "graph", graph, "delta1 %.4f delta2 %.4f (ref self-noise %.4f)" % (r1, r2, n8),
              [round(h["l2_delta-avg"], 6) for h in res["history"]], flush=True)
