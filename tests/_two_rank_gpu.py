"""Helper of tests/test_gpu_parity.py: ONE rank of a 2-rank job that shares the box's single GPU (gloo collectives;
RCCL refuses two ranks on one device).  Started through pcfa_amd.launch.spawn_ranks, writes <out>_r<rank>.npy.

    mode cosim:  one universal-attack closure (attack_PCFA.py:469-490) with --loss cosim (losses.py:76-88) on this rank's
                 half of a 2-pair batch, through UniversalAttack (HIP kernels) -> [grad delta1 | grad delta2 | loss]
    mode rccl1:  every collective pcfa_amd.sharding issues, through the "nccl" backend (= RCCL) with a communicator of ONE
                 rank -- the most this one-GPU box can execute of the path the 8-GPU run takes: library load, communicator
                 set-up, the all-reduce / all-gather launches on device buffers, their ordering against a side stream
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def rccl_single_rank(out):
    import json
    os.environ.update(RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    from pcfa_amd import sharding
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    assert sharding.is_dist() and sharding.world_size() == 1
    sharding.barrier()
    rec = {"max": sharding.max_scalar(3.25, dev), "all": sharding.all_scalars(1.5, dev),
           "means": list(sharding.mean_scalars((2.0, -4.0), dev)),
           "rows": sharding.gather_rows([(1.0, 2.0), (3.0, 4.0)], 2, dev)}
    # the universal closure's collective: gradients + loss packed behind work on a side stream, reduced, re-pointed
    gen = torch.Generator().manual_seed(0)
    params = [torch.zeros(3, 64, 80, device=dev, requires_grad=True) for _ in range(2)]
    grads = [torch.randn(3, 64, 80, generator=gen) for _ in range(2)]
    red = sharding.FlatReducer(params)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for p, g in zip(params, grads):
            p.grad = g.to(dev, non_blocking=False) * 1.0
        red.pack(torch.tensor(7.5, device=dev))
        loss = red.reduce()
        got = [p.grad.clone() for p in params]
    side.synchronize()
    rec["loss"] = float(loss)
    rec["grads_equal"] = all(torch.equal(a.cpu(), b) for a, b in zip(got, grads))
    rec["collectives"] = red.collectives
    sums = torch.tensor([1.0, 2.0, 3.0], device=dev)
    rec["batch_sums_world"] = sharding.BatchSums()(sums)
    rec["sums"] = sums.tolist()
    rec["backend"] = torch.distributed.get_backend()
    sharding.shutdown()
    with open(out, "w") as f:
        json.dump(rec, f)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    if mode == "rccl1":
        return rccl_single_rank(out)
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
