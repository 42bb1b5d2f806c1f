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
