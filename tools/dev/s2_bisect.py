#!/usr/bin/env python3
"""3-step RAFT attack at 436x1024 on the GPU with the stride-2 kernels switched on one at a time; per-step metrics next
to the CPU port's from profiles/r03_schedule_parity_20steps.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import schedule_parity  # noqa: E402
from pcfa_amd import hip_ops  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

ref = json.load(open(os.path.join(ROOT, "profiles", "r03_schedule_parity_20steps.json")))
print("port  ", [round(s["aee_adv_init"], 5) for s in ref["port_a"]["per_step"][:3]], [round(s["aee_adv_tgt"], 5) for s in ref["port_a"]["per_step"][:3]])
print("gpu r3", [round(s["aee_adv_init"], 5) for s in ref["gpu"]["per_step"][:3]], [round(s["aee_adv_tgt"], 5) for s in ref["gpu"]["per_step"][:3]])
for name, (cs2, bwd, ds) in (("all off", (False, False, False)), ("fwd only", (True, False, False)),
                             ("fwd+bwd", (True, True, False)), ("all on", (True, True, True))):
    raft.CONV_S2, hip_ops.CONV_S2_BWD, raft.FUSED_DOWNSAMPLE = cs2, bwd, ds
    r = schedule_parity.run("RAFT", 436, 1024, 3, torch.device("cuda", 0))
    print("%-8s" % name, [round(s["aee_adv_init"], 5) for s in r["per_step"]], [round(s["aee_adv_tgt"], 5) for s in r["per_step"]], flush=True)
