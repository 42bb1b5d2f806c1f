"""Iterated FGSM baseline on the same adapter and kernels (reference attack_FGSM.py:21-56,59-308).

One iteration = 1 forward + 1 backward + a signed-gradient step of size --epsilon, images clipped to [0,1];
`--joint_perturbation` averages the two image gradients before taking the sign."""
import torch

from . import sharding
from .attack_PCFA import _load_model, select_device
from .helper_functions import datasets, logging, losses, ownutilities, parsing_file, targets


def fgsm_attack_step(image1, image2, epsilon, image1_grad, image2_grad, image_min=0., image_max=1., clipping=True,
                     common_perturb=False):
    """attack_FGSM.py:21-56."""
    if not common_perturb:
        s1, s2 = image1_grad.sign(), image2_grad.sign()
    else:
        s1 = s2 = (0.5 * (image1_grad + image2_grad)).sign()
    p1 = image1 - epsilon * s1
    p2 = image2 - epsilon * s2
    if clipping:
        p1 = torch.clamp(p1, image_min, image_max)
        p2 = torch.clamp(p2, image_min, image_max)
    return p1, p2


def fgsm_attack(model, image1, image2, flow, device, has_gt, args):
    """The per-pair loop of attack_FGSM.py:150-247; returns the final metrics as a dict."""
    image1, image2, flow = image1.to(device), image2.to(device), flow.to(device)
    if not ownutilities.model_takes_unit_input(args.net):
        image1, image2 = image1 / 255., image2 / 255.
    padder, [image1, image2] = ownutilities.preprocess_img(args.net, image1, image2)
    nw1 = image1.clone().detach().requires_grad_(True)
    nw2 = image2.clone().detach().requires_grad_(True)

    def predict(a, b):
        out = ownutilities.compute_flow(model, "scaled_input_model", a, b, test_mode=True)
        [out] = ownutilities.postprocess_flow(args.net, padder, out)
        return out

    flow_pred = predict(nw1, nw2)
    flow_pred_init = flow_pred.detach().clone()
    target = targets.get_target(args.target, flow_pred_init, custom_target_path=args.custom_target_path,
                                device=device).to(device)
    res = {"aee_pred-tgt": logging.calc_metrics_const(target, flow_pred_init)}
    delta1 = delta2 = torch.zeros_like(image1)
    for _ in range(args.steps):
        loss = losses.get_loss(args.loss, flow_pred, target)
        model.zero_grad()
        loss.backward()
        nw1, nw2 = fgsm_attack_step(nw1, nw2, args.epsilon, nw1.grad.data, nw2.grad.data, clipping=True,
                                    common_perturb=args.joint_perturbation)
        delta1 = torch.clamp(nw1, 0., 1.) - image1
        delta2 = torch.clamp(nw2, 0., 1.) - image2
        nw1 = nw1.detach().requires_grad_(True)
        nw2 = nw2.detach().requires_grad_(True)
        flow_pred = predict(nw1, nw2)
    aee_adv_tgt, aee_adv_pred = logging.calc_metrics_adv(flow_pred, target, flow_pred_init)
    l2_1, l2_2, l2_12 = logging.calc_delta_metrics(delta1.detach(), delta2.detach())
    res.update({"aee_predadv-tgt": aee_adv_tgt, "aee_pred-predadv": aee_adv_pred, "l2_delta1": l2_1,
                "l2_delta2": l2_2, "l2_delta-avg": l2_12})
    if has_gt:
        res["aee_predadv-gt"] = logging.calc_metrics_adv_gt(flow_pred, flow)
    return res


def attack(args, data_loader=None, has_gt=None):
    """attack_FGSM.py:59-308: every pair of the dataset, pairs sharded over ranks like attack_PCFA.attack_l2."""
    rank, world = sharding.rank(), sharding.world_size()
    if data_loader is None:
        data_loader, has_gt = datasets.prepare_dataloader(args, batch_size=1, shuffle=False)
    device = select_device()
    model = _load_model(args, device, variable_change=False)
    keys = ("aee_pred-tgt", "aee_predadv-tgt", "aee_pred-predadv", "l2_delta1", "l2_delta2", "l2_delta-avg")
    rows = []
    for batch, (image1, image2, flow, _) in enumerate(data_loader):
        if batch % world != rank:
            continue
        r = fgsm_attack(model, image1, image2, flow, device, has_gt, args)
        rows.append((batch,) + tuple(r[k] for k in keys))
    rows = sharding.gather_rows(rows, width=len(keys) + 1, device=device)
    if rank != 0:
        return None
    n = len(rows)
    out = {k: sum(r[i + 1] for r in rows) / max(n, 1) for i, k in enumerate(keys)}
    out["pairs"] = n
    logging.calc_log_averages(1, *[("avg_" + k, v) for k, v in out.items()])
    return out


def main(argv=None):
    args = parsing_file.create_parser(stage='training', attack_type='fgsm').parse_args(argv)
    sharding.init_from_env()
    try:
        return attack(args)
    finally:
        sharding.shutdown()


if __name__ == '__main__':
    main()
