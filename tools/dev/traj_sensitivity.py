#!/usr/bin/env python3
"""How much does the 3-step RAFT trajectory at 436x1024 move under a rounding-level change elsewhere in the network
(conv3x3 algorithm policy, PCFA_CONV3X3_ALGO), with the library stride-2 layers?  argv: s2 on|off"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import schedule_parity  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

raft.CONV_S2 = sys.argv[1] == "on"
r = schedule_parity.run("RAFT", 436, 1024, 6, torch.device("cuda", 0))
print("conv_s2 %s algo %s:" % (sys.argv[1], os.environ.get("PCFA_CONV3X3_ALGO", "policy")),
      [round(s["aee_adv_init"], 4) for s in r["per_step"]], [round(s["aee_adv_tgt"], 3) for s in r["per_step"]])
