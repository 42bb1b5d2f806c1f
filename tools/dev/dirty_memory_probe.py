#!/usr/bin/env python3
"""Do results depend on what the allocator's recycled blocks held?  Poison the caching allocator (blocks of many sizes
filled with POISON, then freed: later torch.empty calls get them back), then run a solo attack and the same pair in a
two-lane flight and compare with a reference run made BEFORE the poisoning.
usage: dirty_memory_probe.py [nan|rand|stale|vramnan|vramrand]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import attack_PCFA  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "nan"
dev = torch.device("cuda", 0)
model = bench.load_model("RAFT", dev, True)
H, W, STEPS = 128, 160, 2


def solo(seed):
    if hasattr(model, "_pcfa_pair_graphs"):
        model._pcfa_pair_graphs.clear()
    st = bench.AttackStepper("RAFT", H, W, dev, seed, use_graph=True, model=model)
    for _ in range(STEPS):
        last = st.step()
    return tuple(last), st.delta1.clone(), st.flow_pred.clone()


def poison_vram():
    """Fill (nearly) all free VRAM with the pattern and hand it BACK TO THE DRIVER (empty_cache): what hipMalloc returns
    afterwards -- the segments of a hipGraph's private pool, for one -- holds the pattern unless the driver scrubs."""
    torch.cuda.synchronize()
    free, _ = torch.cuda.mem_get_info(dev)
    blocks, chunk = [], 4 << 30
    n = int(free * 0.92) // chunk
    for _ in range(n):
        t = torch.empty(chunk // 4, device=dev)
        if mode.endswith("nan"):
            t.fill_(float("nan"))
        else:
            t.normal_(0.0, 3.0)
        blocks.append(t)
    torch.cuda.synchronize()
    print("poisoned %d GiB of VRAM" % (n * 4), flush=True)
    del blocks
    torch.cuda.empty_cache()
    probe = torch.empty(1 << 28, device=dev)   # 1 GiB straight from the driver: does it come back dirty?
    print("fresh 1 GiB block: %.1f %% of its words are non-zero" % (100.0 * float((probe.view(torch.int32) != 0).float().mean())),
          flush=True)
    del probe
    torch.cuda.empty_cache()


def poison():
    if mode.startswith("vram"):
        return poison_vram()
    torch.cuda.synchronize()
    blocks = []
    g = torch.Generator(device=dev).manual_seed(1)
    for mb in (0.004, 0.03, 0.25, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        for _ in range(6):
            n = int(mb * 262144)
            t = torch.empty(n, device=dev)
            if mode == "nan":
                t.fill_(float("nan"))
            else:
                t.copy_(torch.randn(n, device=dev, generator=g) * 3.0)
            blocks.append(t)
    torch.cuda.synchronize()
    del blocks


ref11 = solo(11)
ref12 = solo(12)
print("reference", ref11[0], ref12[0], flush=True)
if mode != "stale":
    model._pcfa_pair_graphs.clear()
    poison()
a = solo(11)
print("solo 11 after poisoning:", a[0], "equal" if a[0] == ref11[0] and torch.equal(a[1], ref11[1]) else "DIFFERENT", flush=True)
model._pcfa_pair_graphs.clear()
if mode != "stale":
    poison()
flight = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("RAFT", H, W, dev, (11, 12)[k], use_graph=True, model=model), 2, dev)
last = flight.run(STEPS)
for k, ref in enumerate((ref11, ref12)):
    same = tuple(last[k]) == ref[0] and torch.equal(flight.attacks[k].delta1, ref[1])
    print("flight lane %d:" % k, tuple(last[k]), "equal" if same else "DIFFERENT", flush=True)
