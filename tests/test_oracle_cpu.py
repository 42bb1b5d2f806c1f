"""The oracle (CPU restatement) against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py) and against the reference's own C++ sampler build (oracle/_ref)."""
import os

import numpy as np
import pytest
import torch

from tests.util import load_golden, max_abs, rel_l2, t


@pytest.mark.parametrize("tag", ["a", "b"])
def test_corr_block_matches_reference(oracle_ops, tag):
    g = load_golden("corr_block_" + tag)
    f1, f2 = t(g["fmap1"]).requires_grad_(True), t(g["fmap2"]).requires_grad_(True)
    blk = oracle_ops.CorrBlock(f1, f2, num_levels=4, radius=4)
    outs = [blk(t(g[k])) for k in ("coords0", "coords1", "coords2")]
    for i, o in enumerate(outs):
        assert max_abs(o, t(g["out%d" % i])) == 0.0  # same ATen kernels, same order: bit-exact
    for lvl in (1, 2, 3):
        assert max_abs(blk.corr_pyramid[lvl], t(g["pyr%d" % lvl])) == 0.0
    go = t(g["grad_out"])
    ((outs[1] * go).sum() + (outs[2] * go.flip(1)).sum()).backward()
    assert rel_l2(f1.grad, t(g["dfmap1"])) < 1e-6
    assert rel_l2(f2.grad, t(g["dfmap2"])) < 1e-6


def test_lookup_first_principles_loops(oracle_ops):
    """Independent double-loop bilinear lookup (zero padding, x-major window) vs the grid_sample form."""
    g = load_golden("corr_block_a")
    f1, f2 = t(g["fmap1"]), t(g["fmap2"])  # 16x20: the coarsest level must stay >= 2 px
    pyr = oracle_ops.corr_pyramid(f1, f2, 4)
    coords = t(g["coords1"])
    a = oracle_ops.corr_lookup(pyr, coords, 4)
    b = oracle_ops.corr_lookup_loops(pyr, coords, 4)
    assert max_abs(a, b) < 2e-4 * float(a.abs().max())


@pytest.mark.parametrize("tag", ["pwc_a", "pwc_b", "pwc_c", "gen_a", "gen_b"])
def test_spatial_corr_c_port_matches_reference(oracle_ops, tag):
    g = load_golden("spatial_corr_" + tag)
    ks, ps, st, pad, dil, dp = (int(v) for v in g["params"])
    a, b = t(g["in1"]).requires_grad_(True), t(g["in2"]).requires_grad_(True)
    out = oracle_ops.spatial_correlation_sample(a, b, ks, ps, st, pad, dil, dp)
    assert out.shape == g["out"].shape
    assert max_abs(out, t(g["out"])) == 0.0          # same loop nest / accumulation order: bit-exact
    out.backward(t(g["grad_out"]))
    assert max_abs(a.grad, t(g["gin1"])) == 0.0
    assert max_abs(b.grad, t(g["gin2"])) == 0.0


def test_spatial_corr_c_port_vs_reference_build(oracle_ops):
    """Against the reference's own C++ build where it exists (oracle/_ref, built from /root/reference)."""
    from oracle import build_ref
    ref = build_ref.load_module()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    gen = torch.Generator().manual_seed(3)
    a, b = torch.randn(2, 13, 14, 19, generator=gen), torch.randn(2, 13, 14, 19, generator=gen)
    for (k, p, s, pad, dil, dp) in ((1, 9, 1, 0, 1, 1), (3, 5, 2, 2, 2, 2), (1, 21, 1, 0, 1, 2)):
        out_ref = ref.forward(a, b, k, k, p, p, pad, pad, dil, dil, dp, dp, s, s)
        out = oracle_ops.spatial_correlation_sample(a, b, k, p, s, pad, dil, dp)
        assert max_abs(out, out_ref) == 0.0
        go = torch.randn(out.shape, generator=gen)
        g1r, g2r = ref.backward(a, b, go, k, k, p, p, pad, pad, dil, dil, dp, dp, s, s)
        a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        oracle_ops.spatial_correlation_sample(a2, b2, k, p, s, pad, dil, dp).backward(go)
        assert max_abs(a2.grad, g1r) == 0.0 and max_abs(b2.grad, g2r) == 0.0
    # and the independent shift-multiply-sum formulation
    ssum = oracle_ops.spatial_correlation_shift_sum(a, b, 9)
    assert max_abs(ssum, oracle_ops.spatial_correlation_sample(a, b, 1, 9)) < 1e-4


def test_attack_math_matches_reference(oracle_ops):
    g = load_golden("attack_math")
    pred, target = t(g["pred"]), t(g["target"])
    img1, img2 = t(g["image1"]), t(g["image2"])
    for box, (ka, kb) in (("change_of_variables", ("w1", "w2")), ("clipping", ("c1", "c2"))):
        a, b = t(g[ka]).requires_grad_(True), t(g[kb]).requires_grad_(True)
        d1, d2 = oracle_ops.extract_deltas(a, b, img1, img2, box, eps_box=1e-7)
        assert max_abs(d1, t(g["delta1_" + box])) == 0.0 and max_abs(d2, t(g["delta2_" + box])) == 0.0
        gd = t(g["gdelta"])
        ((d1 * gd).sum() + (d2 * gd.flip(-1)).sum()).backward()
        assert max_abs(a.grad, t(g["gw1_" + box])) == 0.0
        for f_type in ("aee", "mse", "cosim"):
            for bound in (0.005, 10.0):
                key = "%s_%s_%g" % (box, f_type, bound)
                p = pred.clone().requires_grad_(True)
                dd1, dd2 = d1.detach().clone().requires_grad_(True), d2.detach().clone().requires_grad_(True)
                loss = oracle_ops.loss_delta_constraint(p, target, dd1, dd2, None, delta_bound=bound, mu=5e5,
                                                        f_type=f_type)
                loss.backward()
                assert abs(float(loss) - float(g["loss_" + key])) <= 1e-6 * abs(float(g["loss_" + key]))
                assert rel_l2(p.grad, t(g["gpred_" + key])) < 1e-6
                assert rel_l2(dd1.grad, t(g["gd1_" + key])) < 1e-6 or float(np.abs(g["gd1_" + key]).max()) == 0
    nd = t(g["nw_delta"]).requires_grad_(True)
    dj, dj2 = oracle_ops.extract_deltas_joint(nd, torch.max(img1, img2), torch.min(img1, img2))
    assert dj is dj2 and max_abs(dj, t(g["delta_joint"])) == 0.0
    assert abs(float(oracle_ops.avg_epe(pred, target)) - float(g["aee"])) < 1e-6
    assert abs(float(oracle_ops.avg_epe(pred[0], target[0])) - float(g["aee3"])) < 1e-6


# --------------------------------------------------------------------------- FlowNet2's three operators
# No reference output exists for them on this machine (CUDA-only extensions): the correlation is cross-pinned against
# the reference's C++ sampler (pinned above), Resample2d / ChannelNorm against element-by-element loops that restate
# the CUDA kernels independently of the vectorised oracle, and every backward against autograd.
def _sampler_as_flownet_corr(oracle_ops, a, b, md, s2):
    d = 2 * (md // s2) + 1
    out = oracle_ops.spatial_correlation_sample(a, b, 1, d, 1, 0, 1, s2)
    return out.reshape(a.shape[0], d * d, a.shape[2], a.shape[3]) / a.shape[1]


def _check_losses_names(L, g, dev, tol):
    """Shared by the CPU (oracle) and GPU (product) tests: every remaining public name of losses.py against the
    reference's values and gradients (tests/golden/losses_names.npz)."""
    pred = t(g["pred"], dev).requires_grad_(True)
    target = t(g["target"], dev)
    d1 = t(g["delta1"], dev).requires_grad_(True)
    d2 = t(g["delta2"], dev).requires_grad_(True)

    def check(fn, leaves, want, want_grads):
        for l in leaves:
            l.grad = None
        v = fn()
        assert abs(float(v) - float(want)) <= tol * max(abs(float(want)), 1e-12), (float(v), float(want))
        v.backward()
        for l, w in zip(leaves, want_grads):
            assert rel_l2(l.grad, t(w)) <= 10 * tol

    check(lambda: L.avg_mse(pred, target), [pred], g["avg_mse"], [g["g_avg_mse"]])
    assert abs(float(L.f_mse(pred, target)) - float(g["f_mse"])) <= tol * abs(float(g["f_mse"]))
    check(lambda: L.f_cosim(pred, target), [pred], g["f_cosim"], [g["g_f_cosim"]])
    check(lambda: L.two_norm_avg_delta_squared(d1, d2), [d1, d2], g["msq"], [g["g_msq_1"], g["g_msq_2"]])
    check(lambda: L.relu_penalty(d1, d2, dev, 0.005), [d1, d2], g["penalty_active"],
          [g["g_penalty_active_1"], g["g_penalty_active_2"]])
    for l in (d1, d2):
        l.grad = None
    v = L.relu_penalty(d1, d2, dev, 0.5)
    assert float(v) == 0.0 == float(g["penalty_inactive"])
    v.backward()
    assert float(d1.grad.abs().max()) == 0.0 and float(d2.grad.abs().max()) == 0.0


def test_losses_names_match_reference(oracle_ops):
    _check_losses_names(oracle_ops, load_golden("losses_names"), "cpu", 1e-6)


@pytest.mark.parametrize("shape,md,s2", [((1, 16, 20, 28), 20, 2), ((2, 5, 9, 13), 4, 2), ((1, 3, 7, 8), 3, 1)])
def test_flownet_corr_oracle_vs_pinned_sampler(oracle_ops, shape, md, s2):
    """FlowNetC's layer (pad = max_displacement, k = 1, stride1 = 1) is the sampler with patch 2*(md/s2)+1,
    dilation_patch s2, divided by C (correlation_cuda_kernel.cu:104,143) -- forward and both gradients."""
    g = torch.Generator().manual_seed(31)
    a = torch.randn(shape, generator=g)
    b = torch.randn(shape, generator=g)
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out = oracle_ops.flownet_correlation(a1, b1, md, 1, md, 1, s2)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    want = _sampler_as_flownet_corr(oracle_ops, a2, b2, md, s2)
    assert out.shape == want.shape
    assert max_abs(out, want) < 2e-6 * float(want.detach().abs().max())
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    want.backward(go)
    assert rel_l2(a1.grad, a2.grad) < 1e-6 and rel_l2(b1.grad, b2.grad) < 1e-6


def test_flownet_corr_oracle_geometry_and_kernel3():
    """Output geometry of correlation_cuda.cc:25-35 and the k = 3 window sums of the backward (:173-193)."""
    from oracle import ops as O
    g = torch.Generator().manual_seed(32)
    a = torch.randn(1, 4, 10, 12, generator=g, dtype=torch.float64)
    b = torch.randn(1, 4, 10, 12, generator=g, dtype=torch.float64)
    for pad, k, md, s2, want_hw in ((5, 3, 4, 2, (10, 12)), (2, 1, 4, 2, (6, 8)), (3, 3, 2, 1, (10, 12))):
        a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = O.flownet_corr_forward(a1, b1, pad, k, md, 1, s2)
        assert tuple(out.shape[-2:]) == want_hw and out.shape[1] == (2 * (md // s2) + 1) ** 2
        go = torch.randn(out.shape, generator=g, dtype=torch.float64)
        ga, gb = torch.autograd.grad(out, (a1, b1), go)
        g1, g2 = O.flownet_corr_backward(a, b, go, pad, k, md, 1, s2)
        assert rel_l2(g1, ga) < 1e-12 and rel_l2(g2, gb) < 1e-12


def _resample2d_loops(img, flow):
    """resample2d_kernel.cu:16-72 / :75-201, one output element at a time (numpy fp32 scalars)."""
    img, flow = img.numpy(), flow.numpy()
    B, C, H, W = img.shape
    out = np.zeros((B, C, H, W), np.float32)

    def clampi(v, hi):
        return max(min(int(v), hi), 0)
    for b in range(B):
        for y in range(H):
            for x in range(W):
                xf = np.float32(x) + flow[b, 0, y, x]
                yf = np.float32(y) + flow[b, 1, y, x]
                al = np.float64(xf - np.floor(xf))
                be = np.float64(yf - np.floor(yf))
                xL, xR = clampi(np.floor(xf), W - 1), clampi(np.floor(xf) + 1, W - 1)
                yT, yB = clampi(np.floor(yf), H - 1), clampi(np.floor(yf) + 1, H - 1)
                for c in range(C):
                    v = np.float32(0)
                    v += np.float32((1. - al) * (1. - be) * img[b, c, yT, xL])
                    v += np.float32(al * (1. - be) * img[b, c, yT, xR])
                    v += np.float32((1. - al) * be * img[b, c, yB, xL])
                    v += np.float32(al * be * img[b, c, yB, xR])
                    out[b, c, y, x] = v
    return torch.from_numpy(out)


def test_resample2d_oracle_vs_loops_and_autograd(oracle_ops):
    g = torch.Generator().manual_seed(33)
    img = torch.randn(2, 3, 9, 11, generator=g)
    flow = 4 * torch.randn(2, 2, 9, 11, generator=g)  # plenty of positions outside the image (clamped taps)
    flow[0, :, 0, 0] = torch.tensor([-0.25, -0.75])   # negative position next to the border (trunc vs floor weights)
    out = oracle_ops.resample2d(img, flow)
    assert max_abs(out, _resample2d_loops(img, flow)) == 0.0
    # nearest mode (resample2d_kernel.cu:65-70)
    near = oracle_ops.resample2d(img, flow, 1, False)
    xs = torch.arange(11).view(1, 1, 11) + flow[:, 0]
    ys = torch.arange(9).view(1, 9, 1) + flow[:, 1]
    xn = torch.floor(xs + 0.5).long().clamp(0, 10)
    yn = torch.floor(ys + 0.5).long().clamp(0, 8)
    for b in range(2):
        assert torch.equal(near[b], img[b][:, yn[b], xn[b]])
    # backward == autograd of the bilinear form (floor is piecewise constant)
    i1, f1 = img.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    go = torch.randn(out.shape, generator=g)
    oracle_ops.resample2d(i1, f1).backward(go)
    i2, f2 = img.clone().double().requires_grad_(True), flow.clone().double().requires_grad_(True)
    xf = torch.arange(11).view(1, 1, 11) + f2[:, 0]
    yf = torch.arange(9).view(1, 9, 1) + f2[:, 1]
    fx, fy = torch.floor(xf).detach(), torch.floor(yf).detach()
    al, be = (xf - fx).unsqueeze(1), (yf - fy).unsqueeze(1)
    xL, xR = fx.long().clamp(0, 10), (fx + 1).long().clamp(0, 10)
    yT, yB = fy.long().clamp(0, 8), (fy + 1).long().clamp(0, 8)
    G = oracle_ops._rs_gather
    ref = ((1 - al) * (1 - be) * G(i2, yT, xL) + al * (1 - be) * G(i2, yT, xR) + (1 - al) * be * G(i2, yB, xL)
           + al * be * G(i2, yB, xR))
    ref.backward(go.double())
    assert rel_l2(i1.grad, i2.grad) < 1e-6 and rel_l2(f1.grad, f2.grad) < 1e-6


def test_channelnorm_oracle(oracle_ops):
    g = torch.Generator().manual_seed(34)
    x = torch.randn(2, 3, 5, 7, generator=g)
    x[0, :, 0, 0] = 0.  # zero vector: the backward divides by (0 + 1e-9)
    x1 = x.clone().requires_grad_(True)
    out = oracle_ops.channelnorm(x1)
    want = np.sqrt((x.numpy()[:, 0] ** 2 + x.numpy()[:, 1] ** 2) + x.numpy()[:, 2] ** 2)[:, None]
    assert np.array_equal(out.detach().numpy(), want)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    assert torch.equal(x1.grad[0, :, 0, 0], torch.zeros(3))
    x2 = x.clone().double().requires_grad_(True)
    (x2.pow(2).sum(1, keepdim=True) + 1e-300).sqrt().backward(go.double())
    mask = torch.ones_like(x, dtype=torch.bool)
    mask[0, :, 0, 0] = False
    assert rel_l2(x1.grad[mask], x2.grad[mask]) < 1e-6
