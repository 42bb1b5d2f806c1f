#!/usr/bin/env python3
"""Does a closure give the same bits while ANOTHER stream keeps the chip busy?  A background thread launches LDS-hogging
(pcfa_poison_lds: 160 KB per workgroup) and matrix-pipe-hogging (pcfa_calib_mfma_f32) kernels on its own stream while the
captured closure is replayed N times; every replay's loss and gradients are compared with a quiet replay.  A kernel with a
timing-dependent defect (a missing wait or barrier that lock-step execution hides) shows up here.
usage: noise_stress.py [NET] [HxW] [N] [joint]"""
import ctypes
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import _hip  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128x160").split("x"))
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30
joint = len(sys.argv) > 4
dev = torch.device("cuda", 0)
lib = _hip.load()
st = bench.AttackStepper(net, h, w, dev, 3, boxconstraint="clipping" if joint else "change_of_variables", joint=joint,
                         use_graph=True)
assert st.graphed is not None


def evaluate():
    loss = st.closure()
    torch.cuda.current_stream().synchronize()
    return loss.detach().clone(), [p.grad.detach().clone() for p in st.optimizer._params]


ref = evaluate()
again = evaluate()
print("quiet replay twice identical:", torch.equal(ref[0], again[0]) and all(torch.equal(a, b) for a, b in zip(ref[1], again[1])),
      flush=True)
stop = threading.Event()
launched = [0]


def noise():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(dev)
    scratch = torch.empty(1 << 20, device=dev)
    with torch.cuda.stream(s):
        k = 0
        while not stop.is_set():
            h_ = ctypes.c_void_p(s.cuda_stream)
            if k % 3 == 0:
                lib.pcfa_poison_lds(0x7fc00000, h_)
            else:
                lib.pcfa_calib_mfma_f32(ctypes.c_void_p(scratch.data_ptr()), 256 * (1 + k % 4), 40, h_)
            k += 1
            launched[0] = k
            if k % 6 == 0:
                s.synchronize()
        s.synchronize()


t = threading.Thread(target=noise)
t.start()
bad = 0
for it in range(N):
    got = evaluate()
    same = torch.equal(ref[0], got[0]) and all(torch.equal(a, b) for a, b in zip(ref[1], got[1]))
    if not same:
        bad += 1
        print("replay %d under noise DIFFERS: loss %r vs %r, max |dgrad| %g" %
              (it, float(got[0]), float(ref[0]), max(float((a - b).abs().max()) for a, b in zip(ref[1], got[1]))), flush=True)
stop.set()
t.join()
print("%s %dx%d: %d replays under noise (%d noise launches), %d differed" % (net, h, w, N, launched[0], bad), flush=True)
