"""Forward-only evaluation of trained (universal) perturbations, also across networks (reference
evaluate_PCFA.py:21-58,60-79,86-299).  Reads the `.npy` artefacts attack_l2_universal writes
(`patches/NNNNN_delta1_e{E}.npy`, logging.save_tensor) or a single `.npy` file."""
import os
import re

import numpy as np
import torch

from . import sharding
from .attack_PCFA import EPS_BOX, _load_model, select_device
from .helper_functions import datasets, logging, losses, ownutilities, parsing_file


def extract_epoch_patchlist(path):
    """evaluate_PCFA.py:21-58: (epochs, delta1 files, delta2 files)."""
    if os.path.isfile(path):
        if os.path.splitext(path)[1] != ".npy":
            raise ValueError("Invalid extension for perturbation file, please use a .npy file instead of %s" % path)
        return 1, [path], []
    base = os.path.join(path, "patches")
    p1, p2 = re.compile("[0-9]{5}_delta1_e[0-9]*.npy"), re.compile("[0-9]{5}_delta2_e[0-9]*.npy")
    d1 = sorted(os.path.join(base, f) for f in os.listdir(base) if p1.match(f))
    d2 = sorted(os.path.join(base, f) for f in os.listdir(base) if p2.match(f))
    if not d1:
        raise ValueError("no NNNNN_delta1_e*.npy files under %s" % base)
    epochs = int(d1[-1].split("_")[-1].split(".")[0][1:]) + 1
    return epochs, d1, d2


def convert_perturbationsizes(delta, image, network_training, network_eval, dataset=None):
    """Re-pad a perturbation trained on one padding family (div 8: RAFT/GMA, div 64: PWCNet/SpyNet/FlowNet2) for a
    network of the other family (evaluate_PCFA.py:60-79)."""
    fnet, raft, unit = ["PWCNet", "SpyNet", "FlowNet2"], ["RAFT", "GMA"], ["PWCNet", "SpyNet"]
    if (network_training in fnet and network_eval in fnet) or (network_training in raft and network_eval in raft):
        return delta
    padder_train, _ = ownutilities.preprocess_img(network_training, image.detach().clone())
    unpadded = torch.unsqueeze(padder_train.unpad(delta), 0)
    _, [repadded] = ownutilities.preprocess_img(network_eval, unpadded.detach().clone())
    if network_eval in unit:  # preprocess_img divided by 255 but delta already lives in [0,1]
        repadded = repadded * 255.
    return repadded


def eval_l2_universal(args, data_loader=None, has_gt=None):
    """Per epoch: AEE(f(image+delta), f(image)) averaged over the dataset (evaluate_PCFA.py:86-299)."""
    if args.origin_net is None:
        raise ValueError("args.origin_net is not allowed to be empty. Please state which network was used to train "
                         "the perturbations via the --origin_net argument.")
    epochs, delta1_paths, delta2_paths = extract_epoch_patchlist(args.perturbation_sourcefolder)
    if data_loader is None:
        data_loader, has_gt = datasets.prepare_dataloader(args, batch_size=args.batch_size, shuffle=False)
    image1_init, image2_init, _, _ = next(iter(data_loader))
    device = select_device()
    cov = args.boxconstraint in ['change_of_variables']
    model = _load_model(args, device, variable_change=cov)
    results = []
    for epoch in range(epochs):
        delta1 = convert_perturbationsizes(torch.from_numpy(np.load(delta1_paths[epoch])), image1_init,
                                           args.origin_net, args.net)
        if args.universal_perturbation and not delta2_paths:
            delta2 = delta1
        else:
            delta2 = convert_perturbationsizes(torch.from_numpy(np.load(delta2_paths[epoch])), image2_init,
                                               args.origin_net, args.net)
        delta1, delta2 = delta1.to(device).detach(), delta2.to(device).detach()
        total, images = 0., 0
        with torch.no_grad():
            for image1, image2, _flow, _ in data_loader:
                image1, image2 = image1.to(device), image2.to(device)
                if not ownutilities.model_takes_unit_input(args.net):
                    image1, image2 = image1 / 255., image2 / 255.
                padder, [image1, image2] = ownutilities.preprocess_img(args.net, image1, image2)
                f0 = ownutilities.compute_flow(model, "scaled_input_model", image1, image2, test_mode=True)
                kw = {"delta1": delta1} if args.joint_perturbation else {"delta1": delta1, "delta2": delta2}
                f1 = ownutilities.compute_flow(model, "scaled_input_model", image1, image2, test_mode=True, **kw)
                [f0, f1] = ownutilities.postprocess_flow(args.net, padder, f0, f1)
                for i in range(image1.size(0)):
                    total += ownutilities.torchfloat_to_float64(losses.avg_epe(f1[i:i + 1], f0[i:i + 1]))
                    images += 1
        l2 = logging.calc_delta_metrics(delta1, delta2)
        results.append({"epoch": epoch, "epoch_aee_pred-predadv": total / max(images, 1), "l2_delta1": l2[0],
                        "l2_delta2": l2[1], "l2_delta-avg": l2[2], "images": images})
        logging.log_metrics(epoch, *[(k, v) for k, v in results[-1].items() if k != "epoch"])
        print("Finished attacking epoch %d\n\tAEE(f_adv, f_init)=%f\n\tL2(perturbation)  =%f\n"
              % (epoch, results[-1]["epoch_aee_pred-predadv"], l2[2]))
    return results


def main(argv=None):
    args = parsing_file.create_parser(stage='evaluation', attack_type='pcfa').parse_args(argv)
    sharding.init_from_env()
    try:
        return eval_l2_universal(args)
    finally:
        sharding.shutdown()


if __name__ == '__main__':
    main()
