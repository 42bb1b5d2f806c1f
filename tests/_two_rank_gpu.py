"""Helper of tests/test_gpu_parity.py: ONE rank of a 2-rank job that shares the box's single GPU (gloo collectives;
RCCL refuses two ranks on one device).  Started through pcfa_amd.launch.spawn_ranks, writes <out>_r<rank>.npy.

    mode cosim:  one universal-attack closure (attack_PCFA.py:469-490) with --loss cosim (losses.py:76-88) on this rank's
                 half of a 2-pair batch, through UniversalAttack (HIP kernels) -> [grad delta1 | grad delta2 | loss]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.distributed.init_process_group("gloo")
    from tests import closure_util
    from pcfa_amd import attack_PCFA, sharding
    from pcfa_amd.helper_functions import datasets
    dev = torch.device("cuda:0")
    assert mode == "cosim"
    args = closure_util.cli_args(net="RAFT", universal_perturbation=True, boxconstraint="clipping", loss="cosim",
                                 target="custom", custom_target_path=sys.argv[3])
    pairs = [datasets.synthetic_pair(i, 128, 160) for i in range(2)]
    mine = pairs if world == 1 else pairs[rank:rank + 1]
    im1 = torch.stack([p[0] for p in mine])
    im2 = torch.stack([p[1] for p in mine])
    gen = torch.Generator().manual_seed(3)
    d1 = 0.01 * torch.randn(3, 128, 160, generator=gen)
    d2 = 0.01 * torch.randn(3, 128, 160, generator=gen)
    model = closure_util.load_model("RAFT", False, dev)
    ua = attack_PCFA.UniversalAttack(model, d1, d2, dev, 5e5, args)
    assert (ua.batch_sums is not None) == (world > 1) and (world == 1 or not ua.use_graph)
    with torch.no_grad():
        ua.nw_delta1.copy_(d1)
        ua.nw_delta2.copy_(d2)
    ua.begin_batch(im1, im2)
    loss = float(ua.closure())
    if world > 1:
        assert ua.batch_sums.collectives == 1 and ua.reducer.collectives == 1
    np.save("%s_r%d.npy" % (out, rank), np.concatenate([p.grad.detach().cpu().numpy().ravel() for p in ua.params]
                                                        + [np.array([loss], np.float32)]))
    sharding.shutdown()


if __name__ == "__main__":
    main()
