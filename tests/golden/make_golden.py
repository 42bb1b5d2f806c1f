"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run in the build container only (needs /root/reference, which is read-only and is never copied):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does (SURVEY.md section 8c stub recipe):
  * stubs the packages the reference imports but this image lacks (cv2, mlflow, torchvision, png),
    restores the removed alias np.float, chdirs to the reference root (it opens relative paths);
  * gives every network the SAME seeded random weights as pcfa_amd (no checkpoints exist here):
    the state_dict of `pcfa_amd...build_network(net, "random:<seed>")` is loaded into the reference's
    modules -- which also proves parameter-name compatibility;
  * PWCNet: uses the reference's own C++ sampler built by oracle/build_ref.py, and neutralises the
    unconditional `.cuda()` of models/PWCNet/PWCNet.py:194;
  * writes inputs + reference outputs as compressed .npz files (data only).

Fixtures:
  corr_block_*.npz        CorrBlock build + lookup, forward and backward   (models/raft/corr.py)
  spatial_corr_*.npz      spatial_correlation_sample forward/backward       (reference C++ sampler)
  attack_math.npz         losses / extract_deltas / ScaledInputModel prologue
  losses_names.npz        avg_mse, f_mse, f_cosim, two_norm_avg_delta_squared, relu_penalty + gradients (losses.py)
  closure_<net>.npz       one PCFA closure: flow, loss, d loss / d nw_input  (128x160 or 128x192)
  trajectory_raft.npz     5-step pcfa_attack, at 8 and at 3 CPU threads (noise floor)
  universal_raft.npz      attack_l2_universal (attack_PCFA.py:297-566): RAFT 128x160, 2 batches of 2 pairs, 2 steps each,
                          deltas after each batch / the epoch + the metric stream, at 8 and at 3 CPU threads
  closure_flownet2.npz    one PCFA closure of FlowNet2 (128x192).  WIRING ONLY: the reference's Python model code
                          (models/FlowNet/FlowNet2.py, FlowNetC/S/SD/Fusion.py, submodules.py and the three
                          autograd Functions correlation.py / resample2d.py / channelnorm.py) runs for real, but its
                          CUDA-only extension modules correlation_cuda / resample2d_cuda / channelnorm_cuda cannot be
                          built here (no CUDA), so those three module names are bound to the oracle restatement
                          (oracle/ops.py).  The fixture therefore pins layer wiring, parameter names, pre-/post-
                          processing and autograd plumbing -- NOT the arithmetic of the three operators, which stays
                          "parity unpinned" (cross-checked against the pinned C++ sampler instead).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED = 1234
METRIC_LOG = None  # list of (key, value, step) while a golden run wants the reference's mlflow metrics recorded


# ----------------------------------------------------------------------------- environment
def install_stubs():
    sys.dont_write_bytecode = True
    np.float = float  # helper_functions/ownutilities.py:518

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    noop = lambda *a, **k: None
    cv2 = stub("cv2", setNumThreads=noop)
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=noop)

    def log_metric(key=None, value=None, step=None, **kw):  # the reference binds this name at import time
        if METRIC_LOG is not None:
            METRIC_LOG.append((key, float(value), step))

    import contextlib
    ml = stub("mlflow", log_metric=log_metric, log_param=noop, log_artifact=noop, log_artifacts=noop,
              set_experiment=noop, start_run=lambda *a, **k: contextlib.nullcontext(), set_tracking_uri=noop,
              create_experiment=noop, get_experiment_by_name=lambda name: types.SimpleNamespace(experiment_id=0))
    ml.exceptions = stub("mlflow.exceptions", MlflowException=Exception)
    tv = stub("torchvision")
    tv.datasets = stub("torchvision.datasets")
    tv.transforms = stub("torchvision.transforms")
    stub("png")
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    os.chdir(REF)
    # reference C++ sampler + its python wrapper package
    from oracle import build_ref
    build_ref.build()
    assert build_ref.load_module() is not None, "reference sampler not built"
    sys.path.insert(0, os.path.join(REF, "models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module"))
    torch.Tensor.cuda = lambda self, *a, **k: self  # models/PWCNet/PWCNet.py:194


def install_flownet_extension_bindings():
    """Bind the names of the reference's three CUDA-only extension modules to the oracle (see module docstring:
    wiring-only fixture).  Signatures: correlation_cuda.cc:10-16,89-96; resample2d_cuda.cc:6-24;
    channelnorm_cuda.cc:6-25 -- caller-allocated outputs that the callee resizes / fills."""
    from oracle import ops as O

    def corr_fwd(i1, i2, rbot1, rbot2, output, pad, k, md, s1, s2, mult):
        res = O.flownet_corr_forward(i1, i2, pad, k, md, s1, s2)
        output.resize_(res.shape).copy_(res)
        return 1

    def corr_bwd(i1, i2, rbot1, rbot2, gout, g1, g2, pad, k, md, s1, s2, mult):
        a, b = O.flownet_corr_backward(i1, i2, gout, pad, k, md, s1, s2)
        g1.resize_(a.shape).copy_(a)
        g2.resize_(b.shape).copy_(b)
        return 1

    def rs_fwd(i1, i2, output, kernel_size, bilinear):
        output.copy_(O.resample2d_forward(i1, i2, kernel_size, bilinear))
        return 1

    def rs_bwd(i1, i2, gout, g1, g2, kernel_size, bilinear):
        a, b = O.resample2d_backward(i1, i2, gout)
        g1.copy_(a)
        g2.copy_(b)
        return 1

    def cn_fwd(i1, output, norm_deg):
        output.copy_(O.channelnorm_forward(i1))
        return 1

    def cn_bwd(i1, output, gout, g1, norm_deg):
        g1.copy_(O.channelnorm_backward(i1, output, gout))
        return 1

    for name, fwd, bwd in (("correlation_cuda", corr_fwd, corr_bwd), ("resample2d_cuda", rs_fwd, rs_bwd),
                           ("channelnorm_cuda", cn_fwd, cn_bwd)):
        m = types.ModuleType(name)
        m.forward, m.backward = fwd, bwd
        sys.modules[name] = m
    # correlation.py:23 wraps the call in `torch.cuda.device_of(tensor)`, a no-op context for CPU tensors


def product_state(net):
    from pcfa_amd.helper_functions import ownutilities as own
    model = own.build_network(net, weights="random:%d" % WEIGHT_SEED)
    return model.state_dict()


class patched_torch_load:
    """torch.load -> the seeded state dict (keyed the way each reference loader expects it)."""

    def __init__(self, net):
        self.net = net
        self.state = product_state(net)

    def __call__(self, path, *a, **k):
        path = str(path)
        if self.net in ("RAFT", "GMA"):
            return {"module." + k_: v for k_, v in self.state.items()}
        if self.net in ("PWCNet", "FlowNet2"):
            return {"state_dict": self.state}
        # SpyNet: .../modelL{level+1}_F-{conv+1}-{weight|bias}.pth.tar   (models/SpyNet/SpyNet.py:77-81)
        base = os.path.basename(path)
        level = int(base.split("modelL")[1].split("_")[0]) - 1
        conv = int(base.split("-")[1]) - 1
        kind = "weight" if "weight" in base else "bias"
        return self.state["moduleBasic.%d.moduleBasic.%d.%s" % (level, conv * 2, kind)]

    def __enter__(self):
        self.orig = torch.load
        torch.load = self
        return self

    def __exit__(self, *exc):
        torch.load = self.orig


def load_reference_model(net, variable_change, eps_box=1e-7):
    from helper_functions import ownutilities
    with patched_torch_load(net):
        unit = ownutilities.model_takes_unit_input(net)
        kw = {"eps_box": eps_box} if variable_change else {}
        model = ownutilities.import_and_load(net, make_unit_input=not unit, variable_change=variable_change,
                                             make_scaled_input_model=True, device=torch.device("cpu"), **kw)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    return model


def test_images(seed, h, w):
    """Integer-valued [0,255] image pair (stored as uint8)."""
    from pcfa_amd.helper_functions.datasets import synthetic_pair
    i1, i2, _ = synthetic_pair(seed, h, w)
    return i1.round().clamp(0, 255)[None], i2.round().clamp(0, 255)[None]


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------- fixtures
def golden_corr_block():
    from models.raft.corr import CorrBlock
    cases = {"a": dict(B=1, D=64, H=16, W=20, seed=11), "b": dict(B=2, D=32, H=17, W=21, seed=12)}
    for tag, c in cases.items():
        g = torch.Generator().manual_seed(c["seed"])
        f1 = torch.randn(c["B"], c["D"], c["H"], c["W"], generator=g)
        f2 = torch.randn(c["B"], c["D"], c["H"], c["W"], generator=g)
        ys, xs = torch.meshgrid(torch.arange(c["H"]), torch.arange(c["W"]), indexing="ij")
        grid = torch.stack([xs, ys], 0).float()[None].repeat(c["B"], 1, 1, 1)
        coords0 = grid.clone()                                               # integer coordinates (iteration 0)
        coords1 = grid + 3.0 * torch.randn(grid.shape, generator=g)          # generic
        coords2 = grid + 30.0 * torch.randn(grid.shape, generator=g)         # mostly out of the image
        gout = torch.randn(c["B"], 324, c["H"], c["W"], generator=g)
        f1.requires_grad_(True)
        f2.requires_grad_(True)
        blk = CorrBlock(f1, f2, num_levels=4, radius=4)
        outs = [blk(cc) for cc in (coords0, coords1, coords2)]
        # backward through two of the lookups at once (the pyramid gradient accumulates)
        loss = (outs[1] * gout).sum() + (outs[2] * gout.flip(1)).sum()
        loss.backward()
        extra = {"pyr0": blk.corr_pyramid[0]} if tag == "a" else {}  # level 0 of case b alone is 1 MB
        save("corr_block_" + tag, fmap1=f1, fmap2=f2, coords0=coords0, coords1=coords1, coords2=coords2,
             grad_out=gout, out0=outs[0], out1=outs[1], out2=outs[2], dfmap1=f1.grad, dfmap2=f2.grad,
             pyr1=blk.corr_pyramid[1], pyr2=blk.corr_pyramid[2], pyr3=blk.corr_pyramid[3], **extra)


def golden_spatial_corr():
    from spatial_correlation_sampler import spatial_correlation_sample
    cases = {
        # the PWC-Net call (PWCNet.py:48-52) on scaled-down level shapes, incl. ragged sizes
        "pwc_a": dict(seed=21, B=1, C=24, H=12, W=20, kw=dict(kernel_size=1, patch_size=9, stride=1)),
        "pwc_b": dict(seed=22, B=2, C=10, H=9, W=35, kw=dict(kernel_size=1, patch_size=9, stride=1)),
        "pwc_c": dict(seed=23, B=1, C=7, H=3, W=5, kw=dict(kernel_size=1, patch_size=9, stride=1)),
        # generic parameters (FlowNetC-like dilation_patch, and a strided/padded 3x3 kernel)
        "gen_a": dict(seed=24, B=1, C=6, H=10, W=12, kw=dict(kernel_size=1, patch_size=5, stride=1, dilation_patch=2)),
        "gen_b": dict(seed=25, B=2, C=4, H=11, W=9, kw=dict(kernel_size=3, patch_size=3, stride=2, padding=1, dilation=1,
                                                  dilation_patch=1)),
    }
    for tag, c in cases.items():
        g = torch.Generator().manual_seed(c["seed"])
        a = torch.randn(c["B"], c["C"], c["H"], c["W"], generator=g, requires_grad=True)
        b = torch.randn(c["B"], c["C"], c["H"], c["W"], generator=g, requires_grad=True)
        out = spatial_correlation_sample(a, b, **c["kw"])
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        kw = c["kw"]
        params = np.array([kw.get("kernel_size", 1), kw.get("patch_size", 1), kw.get("stride", 1),
                           kw.get("padding", 0), kw.get("dilation", 1), kw.get("dilation_patch", 1)])
        save("spatial_corr_" + tag, in1=a, in2=b, params=params, out=out, grad_out=gout, gin1=a.grad, gin2=b.grad)


def golden_attack_math():
    import attack_PCFA
    from helper_functions import losses
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 11, 13
    pred = 4 * torch.randn(B, 2, H, W, generator=g)
    target = 4 * torch.randn(B, 2, H, W, generator=g)
    img1 = torch.rand(1, 3, H + 5, W + 3, generator=g)
    img2 = torch.rand(1, 3, H + 5, W + 3, generator=g)
    out = dict(pred=pred, target=target, image1=img1, image2=img2)
    eps = 1e-7
    # change-of-variables inputs around the images, clipping inputs that leave [0,1]
    w1 = torch.atanh(2. * (1. - eps) * img1 - (1 - eps)) + 0.05 * torch.randn(img1.shape, generator=g)
    w2 = torch.atanh(2. * (1. - eps) * img2 - (1 - eps)) + 0.05 * torch.randn(img2.shape, generator=g)
    c1 = img1 + 0.2 * torch.randn(img1.shape, generator=g)
    c2 = img2 + 0.2 * torch.randn(img2.shape, generator=g)
    nd = 0.3 * torch.randn(img1.shape, generator=g)
    out.update(w1=w1, w2=w2, c1=c1, c2=c2, nw_delta=nd)
    for box, (a, b) in (("change_of_variables", (w1, w2)), ("clipping", (c1, c2))):
        a = a.clone().requires_grad_(True)
        b = b.clone().requires_grad_(True)
        d1, d2 = attack_PCFA.extract_deltas(a, b, img1, img2, box, eps_box=eps)
        gd = torch.randn(d1.shape, generator=torch.Generator().manual_seed(9))
        ((d1 * gd).sum() + (d2 * gd.flip(-1)).sum()).backward()
        out.update({"delta1_" + box: d1, "delta2_" + box: d2, "gdelta": gd, "gw1_" + box: a.grad,
                    "gw2_" + box: b.grad})
        for f_type in ("aee", "mse", "cosim"):
            for mu, bound in ((5e5, 0.005), (5e5, 10.0)):  # penalty active / inactive
                p = pred.clone().requires_grad_(True)
                dd1 = d1.detach().clone().requires_grad_(True)
                dd2 = d2.detach().clone().requires_grad_(True)
                loss = losses.loss_delta_constraint(p, target, dd1, dd2, torch.device("cpu"), delta_bound=bound,
                                                    mu=mu, f_type=f_type)
                loss.backward()
                key = "%s_%s_%g" % (box, f_type, bound)
                out.update({"loss_" + key: loss, "gpred_" + key: p.grad, "gd1_" + key: dd1.grad,
                            "gd2_" + key: dd2.grad})
    imax, imin = torch.max(img1, img2), torch.min(img1, img2)
    ndv = nd.clone().requires_grad_(True)
    dj, dj2 = attack_PCFA.extract_deltas_joint(ndv, imax, imin)
    assert dj is dj2
    pj = pred.clone().requires_grad_(True)
    lj = losses.loss_delta_constraint(pj, target, dj, dj2, torch.device("cpu"), delta_bound=0.005, mu=5e5,
                                      f_type="aee")
    lj.backward()
    out.update(delta_joint=dj, loss_joint=lj, gnd_joint=ndv.grad, gpred_joint=pj.grad)
    out.update(aee=losses.avg_epe(pred, target), aee3=losses.avg_epe(pred[0], target[0]),
               l2_1=losses.two_norm_avg(d1), l2_12=losses.two_norm_avg_delta(d1, d2))
    save("attack_math", **out)


def golden_losses():
    """The remaining public names of helper_functions/losses.py (:32-88,110-126,177-197) with their gradients:
    avg_mse / f_mse, f_cosim, two_norm_avg_delta_squared, relu_penalty (bound inactive, active)."""
    from helper_functions import losses
    g = torch.Generator().manual_seed(6)
    pred = (3 * torch.randn(2, 2, 9, 14, generator=g)).requires_grad_(True)
    target = 3 * torch.randn(2, 2, 9, 14, generator=g)
    d1 = (0.02 * torch.randn(1, 3, 12, 16, generator=g)).requires_grad_(True)
    d2 = (0.01 * torch.randn(1, 3, 12, 16, generator=g)).requires_grad_(True)
    out = dict(pred=pred, target=target, delta1=d1, delta2=d2)

    def grad_of(fn, *leaves):
        for l in leaves:
            l.grad = None
        v = fn()
        v.backward()
        return v, [l.grad.clone() for l in leaves]

    v, (gp,) = grad_of(lambda: losses.avg_mse(pred, target), pred)
    out.update(avg_mse=v, g_avg_mse=gp, f_mse=losses.f_mse(pred, target))
    v, (gp,) = grad_of(lambda: losses.f_cosim(pred, target), pred)
    out.update(f_cosim=v, g_f_cosim=gp)
    v, (g1, g2) = grad_of(lambda: losses.two_norm_avg_delta_squared(d1, d2), d1, d2)
    out.update(msq=v, g_msq_1=g1, g_msq_2=g2)
    for tag, bound in (("active", 0.005), ("inactive", 0.5)):
        v, (g1, g2) = grad_of(lambda: losses.relu_penalty(d1, d2, torch.device("cpu"), bound), d1, d2)
        out.update({"penalty_" + tag: v, "g_penalty_%s_1" % tag: g1, "g_penalty_%s_2" % tag: g2})
    save("losses_names", **out)


def closure_case(net, h, w, boxconstraint, joint, target_name, loss_name, seed):
    """One closure evaluation exactly as attack_PCFA.py:175-192 performs it."""
    import attack_PCFA
    from helper_functions import losses, ownutilities, targets
    cov = boxconstraint == "change_of_variables"
    eps = 1e-7
    model = load_reference_model(net, cov, eps)
    im1, im2 = test_images(seed, h, w)
    a, b = im1.clone(), im2.clone()
    if not ownutilities.model_takes_unit_input(net):
        a, b = a / 255., b / 255.
    padder, [a, b] = ownutilities.preprocess_img(net, a, b)
    mu = 2500. / 0.005 * (1.0 if target_name == "zero" else 1.5)
    g = torch.Generator().manual_seed(seed + 100)
    if joint:
        nw_delta = (0.01 * torch.randn(a.shape, generator=g)).requires_grad_(True)
        imax, imin = torch.max(a, b), torch.min(a, b)
        n1, n2 = a, b
        fwd = dict(delta1=nw_delta)
        leaves = [nw_delta]
    else:
        if cov:
            n1 = torch.atanh(2. * (1. - eps) * a - (1 - eps))
            n2 = torch.atanh(2. * (1. - eps) * b - (1 - eps))
        else:
            n1, n2 = a.clone(), b.clone()
        n1 = (n1 + 0.02 * torch.randn(a.shape, generator=g)).requires_grad_(True)
        n2 = (n2 + 0.02 * torch.randn(a.shape, generator=g)).requires_grad_(True)
        fwd = {}
        leaves = [n1, n2]
    with torch.no_grad():  # unperturbed prediction defines the target (attack_PCFA.py:118-131)
        if joint or not cov:
            c1, c2 = a, b
        else:
            c1 = torch.atanh(2. * (1. - eps) * a - (1 - eps))
            c2 = torch.atanh(2. * (1. - eps) * b - (1 - eps))
        f0 = ownutilities.compute_flow(model, "scaled_input_model", c1, c2, test_mode=True)
        [f0] = ownutilities.postprocess_flow(net, padder, f0)
    target = targets.get_target(target_name, f0, device=torch.device("cpu"))
    flow = ownutilities.compute_flow(model, "scaled_input_model", n1, n2, test_mode=True, **fwd)
    [flow] = ownutilities.postprocess_flow(net, padder, flow)
    if joint:
        d1, d2 = attack_PCFA.extract_deltas_joint(nw_delta, imax, imin)
    else:
        d1, d2 = attack_PCFA.extract_deltas(n1, n2, a, b, boxconstraint, eps_box=eps)
    loss = losses.loss_delta_constraint(flow, target, d1, d2, torch.device("cpu"), delta_bound=0.005, mu=mu,
                                        f_type=loss_name)
    loss.backward()
    out = dict(image1=im1.to(torch.uint8), image2=im2.to(torch.uint8), flow_init=f0, target=target, flow=flow,
               loss=loss, mu=np.float64(mu), meta=np.array([h, w, seed, int(cov), int(joint)]))
    for i, leaf in enumerate(leaves):
        out["leaf%d" % i] = leaf.detach()
        out["grad%d" % i] = leaf.grad
    return out


def golden_closures():
    save("closure_raft", **closure_case("RAFT", 128, 160, "change_of_variables", False, "zero", "aee", 1))
    save("closure_gma", **closure_case("GMA", 128, 160, "change_of_variables", False, "neg_flow", "aee", 2))
    save("closure_pwcnet", **closure_case("PWCNet", 120, 180, "clipping", True, "zero", "aee", 3))
    save("closure_spynet", **closure_case("SpyNet", 100, 150, "change_of_variables", False, "zero", "mse", 4))


def golden_flownet2():
    install_flownet_extension_bindings()
    save("closure_flownet2", **closure_case("FlowNet2", 128, 192, "change_of_variables", False, "zero", "aee", 5))


def golden_trajectory():
    import attack_PCFA
    from argparse import Namespace
    args = Namespace(net="RAFT", steps=5, joint_perturbation=False, boxconstraint="change_of_variables",
                     delta_bound=0.005, target="zero", custom_target_path="", loss="aee", save_frequency=1,
                     small_save=False, no_save=True, unregistered_artifacts=True)
    im1, im2 = test_images(7, 128, 160)
    res = {}
    for threads in (8, 3):
        torch.set_num_threads(threads)
        model = load_reference_model("RAFT", True, 1e-7)
        torch.autograd.set_detect_anomaly(False)
        r = attack_PCFA.pcfa_attack(model, im1.clone(), im2.clone(), torch.zeros(1, 2, 128, 160), 0, None, 1e-7,
                                    torch.device("cpu"), False, 2500. / 0.005, args)
        res["threads%d" % threads] = np.array([np.nan if v is None else float(v) for v in r], dtype=np.float64)
        print("threads", threads, res["threads%d" % threads])
    torch.set_num_threads(8)
    save("trajectory_raft", image1=im1.to(torch.uint8), image2=im2.to(torch.uint8), **res)


def golden_universal():
    """The reference's own attack_l2_universal (attack_PCFA.py:297-566) on a fixed synthetic loader: RAFT 128x160,
    clipping (universal mode allows nothing else, :365), zero target, AEE, 4 pairs in 2 batches of 2, one epoch,
    2 L-BFGS steps per batch (44 closure evaluations, one optimiser for the whole run).  The loader is a list, so
    the reference's unseeded shuffle (:347) has nothing to permute.  Recorded: the .npy artefacts the reference
    writes (per-batch and per-epoch deltas) and the metric stream it sends to mlflow, at 8 and at 3 CPU threads
    (the reference's own noise floor for this schedule)."""
    global METRIC_LOG
    import glob
    import shutil
    import tempfile
    import attack_PCFA
    from argparse import Namespace
    from helper_functions import logging as rlog, ownutilities
    H, W, NB, BS, STEPS = 128, 160, 2, 2, 2
    pairs = [test_images(20 + i, H, W) for i in range(NB * BS)]
    im1 = torch.cat([p[0] for p in pairs])
    im2 = torch.cat([p[1] for p in pairs])
    loader = [(im1[b * BS:(b + 1) * BS].clone(), im2[b * BS:(b + 1) * BS].clone(),
               torch.zeros(BS, 2, H, W), torch.ones(BS, H, W)) for b in range(NB)]
    ownutilities.prepare_dataloader = lambda *a, **k: (loader, False)
    rlog.save_image = lambda *a, **k: None      # PNG dumps (flow_library / pypng): outside the path
    rlog.save_flow = lambda *a, **k: None
    res = {}
    for threads in (8, 3):
        torch.set_num_threads(threads)
        out = tempfile.mkdtemp(prefix="pcfa_universal_golden_")
        args = Namespace(net="RAFT", steps=STEPS, joint_perturbation=False, universal_perturbation=True,
                         boxconstraint="clipping", delta_bound=0.005, mu=-1., target="zero", custom_target_path="",
                         loss="aee", save_frequency=1, small_save=False, no_save=False, unregistered_artifacts=True,
                         output_folder=out, dataset="Sintel", dataset_stage="training", dstype="final",
                         batch_size=BS, epochs=1, small_run=False)
        METRIC_LOG = []
        with patched_torch_load("RAFT"):
            attack_PCFA.attack_l2_universal(args)
        log, METRIC_LOG = METRIC_LOG, None
        torch.autograd.set_detect_anomaly(False)
        tag = "_t%d" % threads
        for name in ("delta1_b0", "delta2_b0", "delta1_b1", "delta2_b1", "delta1_e0", "delta2_e0"):
            [f] = glob.glob(os.path.join(out, "**", "*_%s.npy" % name), recursive=True)
            res[name + tag] = np.load(f)
        for i in "12":  # the epoch artefact is the perturbation after the last batch: stored once
            assert np.array_equal(res.pop("delta%s_e0" % i + tag), res["delta%s_b1" % i + tag])
        for key in ("aee_pred-tgt", "aee_predadv-tgt", "aee_pred-predadv", "l2_delta1", "l2_delta2", "l2_delta-avg"):
            res[key + tag] = np.array([v for k, v, _ in log if k == key], dtype=np.float64)
        print("threads", threads, {k: res[k + tag] for k in ("aee_predadv-tgt", "aee_pred-predadv", "l2_delta-avg")})
        shutil.rmtree(out)
    torch.set_num_threads(8)
    save("universal_raft", image1=im1.to(torch.uint8), image2=im2.to(torch.uint8),
         meta=np.array([H, W, NB, BS, STEPS]), **res)


if __name__ == "__main__":
    install_stubs()
    torch.manual_seed(0)
    which = sys.argv[1:] or ["corr", "scorr", "math", "losses", "closures", "trajectory", "flownet2", "universal"]
    if "corr" in which:
        golden_corr_block()
    if "scorr" in which:
        golden_spatial_corr()
    if "math" in which:
        golden_attack_math()
    if "losses" in which:
        golden_losses()
    if "closures" in which:
        golden_closures()
    if "trajectory" in which:
        golden_trajectory()
    if "flownet2" in which:
        golden_flownet2()
    if "universal" in which:
        golden_universal()
