"""Perturbation-Constrained Flow Attack driver -- the hot loop of the reference's attack_PCFA.py.

    extract_deltas / extract_deltas_joint   attack_PCFA.py:20-37   (fused HIP, pcfa_amd.ops)
    pcfa_attack                             attack_PCFA.py:40-294
    attack_l2_universal                     attack_PCFA.py:297-566
    attack_l2                               attack_PCFA.py:570-701

Schedule: the averaged-L2 bound is an exact penalty minimised by
L-BFGS(max_iter=10, no line search: pcfa_amd.lbfgs.LBFGS = torch.optim.LBFGS semantics on HIP kernels); one `--steps` iteration = one LBFGS.step = 10
closure evaluations (forward + loss + backward) followed by one re-prediction forward, and the
reported result is the best iterate with ||delta|| <= bound.  Kept: the optimiser's algorithm and host
decisions, the closure arithmetic and the best-iterate rule.  Dropped because they cannot change a result:
  * the `loss.backward()` before every LBFGS.step (attack_PCFA.py:173) -- the closure's
    zero_grad() discards its gradient before anything reads it;
  * the autograd graph of the re-prediction forward (only metrics read it) -> torch.no_grad();
  * torch.autograd.set_detect_anomaly, the per-forward `.cpu()` round trip, mlflow.

Multi-GPU (one process per GPU, torch.distributed over RCCL): independent pairs are sharded
round-robin over ranks with no collective in the data path (`attack_l2`); the universal attack
is data parallel over the batch with ONE all-reduce of d(loss)/d(delta) per closure
(`attack_l2_universal`), see pcfa_amd/sharding.py.
"""
import os
import time

import numpy as np
import torch

from . import ops, sharding
from .helper_functions import datasets, logging, losses, ownutilities, parsing_file, targets
from .helper_functions.config_paths import Conf

EPS_BOX = 1e-7  # attack_PCFA.py:608


def extract_deltas(nw_input1, nw_input2, image1, image2, boxconstraint, eps_box=0.):
    """delta_i = box(nw_input_i) - image_i (attack_PCFA.py:20-29)."""
    return ops.get().extract_deltas(nw_input1, nw_input2, image1, image2, boxconstraint, eps_box=eps_box)


def extract_deltas_joint(nw_delta, images_max, images_min):
    """Two-sided clamp of a shared perturbation against both frames (attack_PCFA.py:32-37)."""
    return ops.get().extract_deltas_joint(nw_delta, images_max, images_min)


def default_mu(args):
    """attack_PCFA.py:578-584."""
    optim_mu = args.mu
    if optim_mu == -1.:
        optim_mu = 2500. / args.delta_bound
        if args.target not in ['zero']:
            optim_mu = 1.5 * optim_mu
    return optim_mu


def select_device():
    """attack_PCFA.py:631-634."""
    if Conf.config('useCPU') or not torch.cuda.is_available():
        return torch.device("cpu")
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.device("cuda", sharding.local_rank())
    return torch.device("cuda")


def _should_save(batch, args):
    return ((batch % args.save_frequency == 0 and not args.small_save) or (args.small_save and batch < 32)) \
        and not args.no_save


def pcfa_attack(model, image1, image2, flow, batch, distortion_folder, eps_box, device, has_gt, optim_mu, args,
                statistics_in_every_step=True):
    """Attack one image pair; returns the reference's 12-tuple (attack_PCFA.py:40-294)."""
    curr_step = batch * args.steps
    aee_gt = aee_gt_tgt = aee_adv_gt = 0.
    aee_adv_tgt = aee_adv_pred = 0.
    l2_delta1 = l2_delta2 = l2_delta12 = 0.

    image1, image2 = image1.to(device), image2.to(device)
    flow = flow.to(device)
    if not ownutilities.model_takes_unit_input(args.net):
        image1 = image1 / 255.
        image2 = image2 / 255.
    padder, [image1, image2] = ownutilities.preprocess_img(args.net, image1, image2)
    image1.requires_grad = False
    image2.requires_grad = False
    images_max = torch.max(image1, image2).detach()
    images_min = torch.min(image1, image2).detach()

    delta1 = torch.zeros_like(image1)
    delta2 = torch.zeros_like(image2)
    nw_delta = None
    cov = args.boxconstraint in ['change_of_variables']

    if args.joint_perturbation:
        if cov:
            raise ValueError("Training a --joint_perturbation with --boxconstraint=change_of_variables is not "
                             "defined. Please use --boxconstraint=clipping.")
        nw_delta = delta1
        nw_delta.requires_grad = True
        nw_input1, nw_input2 = image1, image2
        optimizer = ops.get().LBFGS([nw_delta], max_iter=10)
        fwd_kwargs = {"delta1": nw_delta}
    else:
        if cov:
            nw_input1 = torch.atanh(2. * (1. - eps_box) * (image1 + delta1) - (1 - eps_box))
            nw_input2 = torch.atanh(2. * (1. - eps_box) * (image2 + delta2) - (1 - eps_box))
        else:
            nw_input1 = image1 + delta1
            nw_input2 = image2 + delta2
        nw_input1.requires_grad = True
        nw_input2.requires_grad = True
        optimizer = ops.get().LBFGS([nw_input1, nw_input2], max_iter=10)
        fwd_kwargs = {}

    def predict():
        out = ownutilities.compute_flow(model, "scaled_input_model", nw_input1, nw_input2, test_mode=True,
                                        **fwd_kwargs)
        [out] = ownutilities.postprocess_flow(args.net, padder, out)
        return out

    def current_deltas():
        if args.joint_perturbation:
            return extract_deltas_joint(nw_delta, images_max, images_min)
        return extract_deltas(nw_input1, nw_input2, image1, image2, args.boxconstraint, eps_box=eps_box)

    with torch.no_grad():
        flow_pred_init = predict().detach().clone()
    target = targets.get_target(args.target, flow_pred_init, custom_target_path=args.custom_target_path,
                                device=device).to(device)
    target.requires_grad = False

    aee_tgt = logging.calc_metrics_const(target, flow_pred_init)
    if has_gt:
        aee_gt_tgt, aee_gt = logging.calc_metrics_const_gt(target, flow_pred_init, flow)
    else:
        aee_gt_tgt, aee_gt = None, None
    logging.log_metrics(curr_step, ("aee_pred-tgt", aee_tgt), ("aee_gt-tgt", aee_gt_tgt), ("aee_pred-gt", aee_gt))
    logging.log_metric(key="optim_mu", value=optim_mu, step=curr_step)

    model.zero_grad()
    optimizer.zero_grad()

    delta_below_threshold = False
    delta12_min_val = float('inf')
    aee_adv_tgt_min_val = float('inf')
    aee_adv_pred_min_val = 0.
    delta1_min = delta2_min = flow_pred_min = None
    flow_pred = flow_pred_init

    def closure_body():
        flow_closure = predict()
        d1, d2 = current_deltas()
        loss_closure = losses.loss_delta_constraint(flow_closure, target, d1, d2, device,
                                                    delta_bound=args.delta_bound, mu=optim_mu, f_type=args.loss)
        loss_closure.backward()
        return loss_closure

    # On the GPU the closure (static shapes, variables updated in place by L-BFGS) is captured once per pair
    # into a hipGraph and replayed: same kernels in the same order, no per-launch host work (PCFA_HIP_GRAPH=0 disables).
    graphed = repredict = None
    if torch.device(device).type == "cuda" and os.environ.get("PCFA_HIP_GRAPH", "1") == "1" and args.steps > 0:
        from .graphed import GraphedClosure, GraphedForward
        graphed = GraphedClosure(closure_body, optimizer.param_groups[0]["params"])
        repredict = GraphedForward(lambda: (current_deltas(), predict()), device)

    def closure():
        if graphed is not None:
            return graphed()
        optimizer.zero_grad()
        return closure_body()

    for steps in range(args.steps):
        curr_step = batch * args.steps + steps
        logging.log_metrics(curr_step, ("batch", batch), ("steps", steps), ("epoch", 0))

        optimizer.step(closure)

        if repredict is not None:
            (delta1, delta2), flow_pred = repredict()
        else:
            with torch.no_grad():
                delta1, delta2 = current_deltas()
                flow_pred = predict()

        aee_adv_tgt, aee_adv_pred = logging.calc_metrics_adv(flow_pred, target, flow_pred_init)
        aee_adv_gt = logging.calc_metrics_adv_gt(flow_pred, flow) if has_gt else None
        logging.log_metrics(curr_step, ("aee_predadv-tgt", aee_adv_tgt), ("aee_pred-predadv", aee_adv_pred),
                            ("aee_predadv-gt", aee_adv_gt))
        l2_delta1, l2_delta2, l2_delta12 = logging.calc_delta_metrics(delta1, delta2, curr_step)
        logging.log_metrics(curr_step, ("l2_delta1", l2_delta1), ("l2_delta2", l2_delta2),
                            ("l2_delta-avg", l2_delta12))

        # best-iterate rule, attack_PCFA.py:226-243
        update_minima = False
        if not delta_below_threshold:
            if l2_delta12 < delta12_min_val or (l2_delta12 == delta12_min_val and aee_adv_tgt < aee_adv_tgt_min_val):
                update_minima = True
                if l2_delta12 <= args.delta_bound:
                    delta_below_threshold = True
        elif l2_delta12 <= args.delta_bound and aee_adv_tgt < aee_adv_tgt_min_val:
            update_minima = True
        if update_minima:
            delta12_min_val = l2_delta12
            aee_adv_tgt_min_val = aee_adv_tgt
            aee_adv_pred_min_val = aee_adv_pred
            delta1_min = delta1.detach().clone()
            delta2_min = delta2.detach().clone()
            flow_pred_min = flow_pred.detach().clone()
        logging.log_metrics(curr_step, ("aee_pred-tgt_min", aee_adv_tgt_min_val),
                            ("l2_delta-avg_min", delta12_min_val), ("aee_pred-predadv_min", aee_adv_pred_min_val))

    if distortion_folder is not None and _should_save(batch, args):
        for tens, name in ((delta1, "delta1_final"), (delta2, "delta2_final"), (delta1_min, "delta1_best"),
                           (delta2_min, "delta2_best"), (image1, "image1"), (image2, "image2"), (target, "target"),
                           (flow_pred, "flow_pred_final"), (flow_pred_min, "flow_pred_best"),
                           (flow_pred_init, "flow_pred_init")):
            logging.save_tensor(tens, name, batch, distortion_folder)
        if has_gt:
            logging.save_tensor(flow, "flow_gt", batch, distortion_folder)

    return (aee_gt, aee_tgt, aee_gt_tgt, aee_adv_gt, aee_adv_tgt, aee_adv_pred, l2_delta1, l2_delta2, l2_delta12,
            aee_adv_tgt_min_val, aee_adv_pred_min_val, delta12_min_val)


def _load_model(args, device, variable_change):
    model_takes_unit_input = ownutilities.model_takes_unit_input(args.net)
    kwargs = {"weights": getattr(args, "weights", "pretrained")}
    if variable_change:
        kwargs["eps_box"] = EPS_BOX
    model = ownutilities.import_and_load(args.net, make_unit_input=not model_takes_unit_input,
                                         variable_change=variable_change, make_scaled_input_model=True,
                                         device=device, **kwargs)
    model.eval()
    for param in model.parameters():
        param.requires_grad = False
    return model


def _output_folder(args, tag):
    if args.no_save:
        return None
    stamp = time.strftime("%Y-%m-%d_%H:%M:%S")
    kind = "%s_PCFA_%s_%s" % (args.net, "cd" if args.joint_perturbation else "dd",
                              "u" if args.universal_perturbation else "-")
    folder = os.path.join(args.output_folder, kind, stamp + tag)
    return logging.create_subfolder(folder, "patches")


def attack_l2(args, data_loader=None, has_gt=None):
    """PCFA on every pair of a dataset, one perturbation (pair) per image pair (attack_PCFA.py:570-701).

    With torch.distributed initialised, pair i is attacked on rank i % world_size and rank 0
    receives every pair's result tuple; there is no collective inside the attack.
    Returns the dict of averaged metrics (also logged under the reference's metric names).
    """
    optim_mu = default_mu(args)
    rank, world = sharding.rank(), sharding.world_size()
    distortion_folder = _output_folder(args, "_r%d" % rank if world > 1 else "")
    if data_loader is None:
        data_loader, has_gt = datasets.prepare_dataloader(args, batch_size=1, shuffle=False)
    device = select_device()
    cov = args.boxconstraint in ['change_of_variables']
    model = _load_model(args, device, variable_change=cov)

    local = []
    for batch, (image1, image2, flow, _) in enumerate(data_loader):
        if batch % world != rank:
            continue
        res = pcfa_attack(model, image1, image2, flow, batch, distortion_folder, EPS_BOX, device, has_gt, optim_mu,
                          args)
        local.append((batch,) + tuple(float('nan') if v is None else float(v) for v in res))

    rows = sharding.gather_rows(local, width=13, device=device)
    if rank != 0:
        return None
    rows = sorted(rows, key=lambda r: r[0])
    tests = len(rows)
    names = ("aee_avg_pred-gt", "aee_avg_pred-tgt", "aee_avg_gt-tgt", "aee_avg_predadv-gt", "aee_avg_predadv-tgt",
             "aee_avg_pred-predadv", "l2_avg_delta1", "l2_avg_delta2", "l2_avg_delta12", "aee_avg_predadv-tgt_min",
             "aee_avg_pred-predadv_min", "l2_avg_delta12_min")
    sums = {n: float(np.nansum([r[i + 1] for r in rows])) for i, n in enumerate(names)}
    logging.calc_log_averages(tests, *sums.items())
    result = {n: (v / tests if tests else float('nan')) for n, v in sums.items()}
    result["pairs"] = tests
    print("\nFinished attacking with PCFA. The best achieved values are")
    print("\tAEE(f_adv, f_init)=%f" % result["aee_avg_pred-predadv_min"])
    print("\tAEE(f_adv, f_targ)=%f" % result["aee_avg_predadv-tgt_min"])
    print("\tL2(perturbation)  =%f" % result["l2_avg_delta12_min"])
    return result


def attack_l2_universal(args, data_loader=None, has_gt=None):
    """One perturbation for a whole dataset (attack_PCFA.py:297-566), data parallel over the batch.

    Every rank holds batch_size/world_size pairs of each global batch, a replica of delta and of
    the L-BFGS state.  Per closure: local forward/backward, then one all-reduce(AVG) of
    d(loss)/d(delta) and of the scalar loss.  The penalty depends on delta only, so it is identical
    on every rank and its average is itself; the similarity term is a mean over the batch, so the
    average of the local means equals the reference's single-process mean over the global batch.
    """
    optim_mu = default_mu(args)
    rank, world = sharding.rank(), sharding.world_size()
    if world > 1 and args.loss == "cosim":
        raise NotImplementedError("cosim is a ratio of batch sums and does not decompose over ranks")
    distortion_folder = _output_folder(args, "") if rank == 0 else None
    device = select_device()
    if data_loader is None:
        if args.batch_size % world != 0:
            raise ValueError("--batch_size %d must be divisible by the %d ranks" % (args.batch_size, world))
        data_loader, has_gt = datasets.prepare_dataloader(args, batch_size=args.batch_size // world, shuffle=True,
                                                          shard=(rank, world))
    model = _load_model(args, device, variable_change=False)  # universal = clipping only (attack_PCFA.py:365)
    model_takes_unit_input = ownutilities.model_takes_unit_input(args.net)

    image1_init, image2_init, _, _ = next(iter(data_loader))
    _, [image1_init, image2_init] = ownutilities.preprocess_img(args.net, image1_init, image2_init)
    nw_delta1 = torch.zeros_like(image1_init[0, :, :, :]).to(device)
    nw_delta2 = torch.zeros_like(image2_init[0, :, :, :]).to(device)
    nw_delta1.requires_grad = True
    if args.joint_perturbation:
        params = [nw_delta1]
    else:
        nw_delta2.requires_grad = True
        params = [nw_delta1, nw_delta2]
    optimizer = ops.get().LBFGS(params, max_iter=10)

    def deltas():
        return (nw_delta1, nw_delta1) if args.joint_perturbation else (nw_delta1, nw_delta2)

    history = []
    batch_ctr = -1
    for epoch in range(args.epochs):
        for batch, (image1, image2, flow, _) in enumerate(data_loader):
            batch_ctr += 1
            curr_step = batch_ctr * args.steps
            image1, image2 = image1.to(device), image2.to(device)
            if has_gt:
                flow = flow.to(device)
            if not model_takes_unit_input:
                image1 = image1 / 255.
                image2 = image2 / 255.
            padder, [image1, image2] = ownutilities.preprocess_img(args.net, image1, image2)

            def predict(perturbed=True):
                kw = {}
                if perturbed:
                    kw = {"delta1": nw_delta1} if args.joint_perturbation else {"delta1": nw_delta1,
                                                                                  "delta2": nw_delta2}
                out = ownutilities.compute_flow(model, "scaled_input_model", image1, image2, test_mode=True, **kw)
                [out] = ownutilities.postprocess_flow(args.net, padder, out)
                return out

            with torch.no_grad():
                flow_pred_init = predict(perturbed=False).detach().clone()
            target = targets.get_target(args.target, flow_pred_init, custom_target_path=args.custom_target_path,
                                        device=device).to(device)
            aee_tgt = sharding.mean_scalar(logging.calc_metrics_const(target, flow_pred_init), device)
            logging.log_metrics(curr_step, ("aee_pred-tgt", aee_tgt))
            model.zero_grad()
            optimizer.zero_grad()

            def closure():
                optimizer.zero_grad()
                d1, d2 = deltas()
                loss_closure = losses.loss_delta_constraint(predict(), target, d1, d2, device,
                                                            delta_bound=args.delta_bound, mu=optim_mu,
                                                            f_type=args.loss)
                loss_closure.backward()
                return sharding.allreduce_closure(params, loss_closure)

            for steps in range(args.steps):
                curr_step = batch_ctr * args.steps + steps
                logging.log_metrics(curr_step, ("steps", steps), ("batch", batch), ("epoch", epoch))
                optimizer.step(closure)
                with torch.no_grad():
                    flow_pred = predict()
                d1, d2 = deltas()
                aee_adv_tgt, aee_adv_pred = logging.calc_metrics_adv(flow_pred, target, flow_pred_init)
                aee_adv_tgt = sharding.mean_scalar(aee_adv_tgt, device)
                aee_adv_pred = sharding.mean_scalar(aee_adv_pred, device)
                l2_delta1, l2_delta2, l2_delta12 = logging.calc_delta_metrics(d1.detach(), d2.detach(), curr_step)
                logging.log_metrics(curr_step, ("aee_predadv-tgt", aee_adv_tgt), ("aee_pred-predadv", aee_adv_pred),
                                    ("l2_delta1", l2_delta1), ("l2_delta2", l2_delta2), ("l2_delta-avg", l2_delta12))
                history.append({"epoch": epoch, "batch": batch, "step": steps, "aee_predadv-tgt": aee_adv_tgt,
                                "aee_pred-predadv": aee_adv_pred, "l2_delta-avg": l2_delta12})
            if distortion_folder is not None and _should_save(batch_ctr, args):
                logging.save_tensor(nw_delta1, "delta1_b" + str(batch_ctr), batch_ctr, distortion_folder)
                logging.save_tensor(deltas()[1], "delta2_b" + str(batch_ctr), batch_ctr, distortion_folder)
        if distortion_folder is not None:
            # `NNNNN_delta1_e{E}.npy`: the pattern evaluate_PCFA.py:42-43 looks for
            logging.save_tensor(nw_delta1, "delta1_e" + str(epoch), batch_ctr, distortion_folder)
            if not args.joint_perturbation:
                logging.save_tensor(nw_delta2, "delta2_e" + str(epoch), batch_ctr, distortion_folder)
    return {"delta1": nw_delta1.detach(), "delta2": deltas()[1].detach(), "history": history}


def main(argv=None):
    parser = parsing_file.create_parser(stage='training', attack_type='pcfa')
    args = parser.parse_args(argv)
    print(args)
    sharding.init_from_env()
    try:
        if args.universal_perturbation:
            return attack_l2_universal(args)
        return attack_l2(args)
    finally:
        sharding.shutdown()


if __name__ == '__main__':
    main()
