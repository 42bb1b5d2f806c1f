#!/usr/bin/env python3
"""20-step PWC-Net attack (375x1242, joint, clipping) on the GPU with this round's PWC changes switched off one at a
time; best-iterate metrics against the CPU port's (profiles/r03_schedule_parity_pwcnet_20steps_det_warp.json).
argv: s2=0|1 dil=0|1  (PCFA_CONV3X3_ALGO from the environment)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import schedule_parity  # noqa: E402
import dataclasses  # noqa: E402

from pcfa_amd import config  # noqa: E402

kv = dict(a.split("=") for a in sys.argv[1:])
conf = dataclasses.replace(config.DEFAULT, conv_s2=kv.get("s2", "1") == "1",
                           dilated_as_subgrids=config.DEFAULT.dilated_as_subgrids if kv.get("dil", "1") == "1" else ())
ref = json.load(open(os.path.join(ROOT, "profiles", "r03_schedule_parity_pwcnet_20steps_det_warp.json")))
seed = int(kv.get("seed", "0"))
r = schedule_parity.run("PWCNet", 375, 1242, 20, torch.device("cuda", 0), box="clipping", joint=True, seed=seed, config=conf)
print("seed %d " % seed, end="")
print("s2=%s dil=%s algo=%s: tgt_min %.4f (port %.4f / %.4f)  init_at_min %.4f (port %.4f)  first steps %s" % (
    kv.get("s2", "1"), kv.get("dil", "1"), os.environ.get("PCFA_CONV3X3_ALGO", "policy"), r["aee_adv_tgt_min"],
    ref["port_a"]["aee_adv_tgt_min"], ref["port_b"]["aee_adv_tgt_min"], r["aee_adv_init_at_min"],
    ref["port_a"]["aee_adv_init_at_min"], [round(s["aee_adv_tgt"], 3) for s in r["per_step"][:6]]), flush=True)
