"""Parity of the HIP kernels (through the C-ABI) with the oracle and the reference's golden vectors.
All tests here need a real MI355X:  python -m pytest tests -m gpu

Tolerances (fp32 everywhere; stated per test):
  * elementwise kernels: <= 4 ulp-ish (2e-6 relative) -- device tanhf vs Sleef tanh;
  * lookup given the same pyramid: 2e-5 * max|corr| -- the kernel interpolates at cx/2^l + a - r directly,
    the reference round-trips the coordinate through grid_sample's [-1,1] normalisation (<= 2 ulp of x);
  * pyramid GEMM (K = D = 256, fp32 MFMA fma chain vs MKL sgemm): 2e-6 * sqrt(D) * max|corr|;
  * reductions (loss): 1e-6 relative;  gradients through GEMMs: 1e-5 relative L2.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from pcfa_amd import hip_ops, ops
from tests import closure_util
from tests.util import load_golden, max_abs, rel_l2, t

pytestmark = pytest.mark.gpu
DEV = "cuda"

# The two cases that are ~80-100 s of CPU-port work each and whose subject other tests and committed records also cover
# (pair 1 of the 3-step schedule parity: pair 0 here + the 20-step matrix profiles/r05/schedule_parity_matrix.json;
# FlowNet2 driver end to end against the 162 M-weight CPU port: its closure / operator / graph tests) run with
# PCFA_LONG_TESTS=1 only, so that the default `-m gpu` suite stays under ~8 minutes on a loaded box.
LONG_ONLY = pytest.mark.skipif(os.environ.get("PCFA_LONG_TESTS") != "1",
                               reason="long CPU-port comparison: set PCFA_LONG_TESTS=1 (see the comment at the top)")


def _grid(B, H, W):
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(B, 1, 1, 1)


# --------------------------------------------------------------------------- correlation pyramid + lookup
@pytest.mark.parametrize("tag", ["a", "b"])
def test_corr_block_vs_reference_golden(tag):
    g = load_golden("corr_block_" + tag)
    f1, f2 = t(g["fmap1"], DEV).requires_grad_(True), t(g["fmap2"], DEV).requires_grad_(True)
    blk = hip_ops.CorrBlock(f1, f2, num_levels=4, radius=4)
    pyr = blk.corr_pyramid
    cmax = float(np.abs(g["pyr1"]).max())
    D = f1.shape[1]
    for lvl in (1, 2, 3):
        assert pyr[lvl].shape == g["pyr%d" % lvl].shape
        assert max_abs(pyr[lvl], t(g["pyr%d" % lvl])) <= 2e-6 * np.sqrt(D) * cmax + 1e-6
    if "pyr0" in g:
        assert max_abs(pyr[0], t(g["pyr0"])) <= 2e-6 * np.sqrt(D) * float(np.abs(g["pyr0"]).max())
    outs = [blk(t(g[k], DEV)) for k in ("coords0", "coords1", "coords2")]
    omax = float(np.abs(g["out1"]).max())
    for i, o in enumerate(outs):
        assert o.shape == g["out%d" % i].shape
        assert max_abs(o, t(g["out%d" % i])) <= 3e-5 * omax, i
    go = t(g["grad_out"], DEV)
    ((outs[1] * go).sum() + (outs[2] * go.flip(1)).sum()).backward()
    assert rel_l2(f1.grad, t(g["dfmap1"])) < 2e-5
    assert rel_l2(f2.grad, t(g["dfmap2"])) < 2e-5


@pytest.mark.parametrize("shape", [(1, 256, 16, 20), (2, 256, 23, 37), (1, 64, 55, 128)])
def test_corr_block_vs_oracle(oracle_ops, shape):
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(H * W)
    f1c = torch.randn(B, D, H, W, generator=gen).requires_grad_(True)
    f2c = torch.randn(B, D, H, W, generator=gen).requires_grad_(True)
    coords = [_grid(B, H, W), _grid(B, H, W) + 2.5 * torch.randn(B, 2, H, W, generator=gen),
              _grid(B, H, W) + 40 * torch.randn(B, 2, H, W, generator=gen)]
    gos = [torch.randn(B, 324, H, W, generator=gen) for _ in coords]
    ref = oracle_ops.CorrBlock(f1c, f2c)
    ro = [ref(c) for c in coords]
    sum((o * g).sum() for o, g in zip(ro, gos)).backward()

    f1g = f1c.detach().to(DEV).requires_grad_(True)
    f2g = f2c.detach().to(DEV).requires_grad_(True)
    blk = hip_ops.CorrBlock(f1g, f2g)
    go_ = [blk(c.to(DEV)) for c in coords]
    omax = max(float(o.abs().max()) for o in ro)
    for a, b in zip(go_, ro):
        assert max_abs(a, b) <= 3e-5 * omax
    sum((o * g.to(DEV)).sum() for o, g in zip(go_, gos)).backward()
    assert rel_l2(f1g.grad, f1c.grad) < 2e-5
    assert rel_l2(f2g.grad, f2c.grad) < 2e-5


@pytest.mark.parametrize("levels,radius", [(4, 3), (3, 2), (2, 1), (1, 4), (5, 2)])
def test_corr_block_other_radii_and_levels(oracle_ops, levels, radius):
    """The kernels are instantiated for radius 1..4 and any level count <= 8 (RAFT-small uses r = 3).  Five levels
    (ADVICE r03): the window-segment records of the sparse backward hold four levels, deeper pyramids take the dense
    products -- gradients must still match."""
    B, D, H, W = (1, 48, 18, 27) if levels <= 4 else (1, 48, 36, 56)
    gen = torch.Generator().manual_seed(levels * 10 + radius)
    f1c = torch.randn(B, D, H, W, generator=gen).requires_grad_(True)
    f2c = torch.randn(B, D, H, W, generator=gen).requires_grad_(True)
    coords = _grid(B, H, W) + 2.0 * torch.randn(B, 2, H, W, generator=gen)
    want = oracle_ops.CorrBlock(f1c, f2c, num_levels=levels, radius=radius)(coords)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    f1g, f2g = f1c.detach().to(DEV).requires_grad_(True), f2c.detach().to(DEV).requires_grad_(True)
    got = hip_ops.CorrBlock(f1g, f2g, num_levels=levels, radius=radius)(coords.to(DEV))
    assert got.shape == want.shape == (B, levels * (2 * radius + 1) ** 2, H, W)
    assert max_abs(got, want) <= 3e-5 * float(want.abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(f1g.grad, f1c.grad) < 2e-5 and rel_l2(f2g.grad, f2c.grad) < 2e-5


@pytest.mark.parametrize("case", ["raft", "wild", "ragged"])
def test_pyramid_backward_windows_matches_dense(case):
    """CorrBlock's backward (corr.py:13-27,52-60 through autograd) skips the part of the volume's gradient that no
    lookup window touched (pcfa_corr_pyramid_bwd_windows: per 128-wide block, bounding rows of ALL lookups' windows).
    The skipped terms are exact zeros, so the result must equal the dense products up to the order in which the
    split-K partials are summed: 2e-6 relative L2.  'raft': 12 lookups drifting from the identity like a refinement;
    'wild': coordinates far outside the map, huge jumps between lookups, one lookup whose output gets no gradient;
    'ragged': 23 x 37 features (blocks straddle query rows, Q % 16 != 0 -> the dense fallback must still agree)."""
    B, D, H, W = (1, 64, 23, 37) if case == "ragged" else (2, 64, 55, 128)
    gen = torch.Generator().manual_seed(17)
    f1 = torch.randn(B, D, H, W, generator=gen).to(DEV)
    f2 = torch.randn(B, D, H, W, generator=gen).to(DEV)
    base = _grid(B, H, W)
    coords, n = [], 12 if case == "raft" else 5
    for i in range(n):
        if case == "wild":
            c = base + (40.0 * i) * torch.randn(B, 2, 1, 1, generator=gen) + 30 * torch.randn(B, 2, H, W, generator=gen)
        else:
            c = base + 0.7 * i * torch.randn(B, 2, 1, 1, generator=gen) + (0.3 + 0.2 * i) * torch.randn(B, 2, H, W, generator=gen)
        coords.append(c.to(DEV))
    gos = [torch.randn(B, 324, H, W, generator=gen).to(DEV) for _ in range(n)]

    def run(windows):
        a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        blk = hip_ops.CorrBlock(a, b, bwd_windows=windows)   # Config.pyramid_bwd_windows, as the networks pass it
        loss = 0.
        for i, (c, g) in enumerate(zip(coords, gos)):
            out = blk(c)
            if not (case == "wild" and i == 2):      # this lookup's backward never runs
                loss = loss + (out * g).sum()
        loss.backward()
        return a.grad, b.grad

    (a1, b1), (a0, b0) = run(True), run(False)
    assert float(a0.abs().max()) > 0 and float(b0.abs().max()) > 0
    assert rel_l2(a1, a0) < 2e-6 and rel_l2(b1, b0) < 2e-6, (rel_l2(a1, a0), rel_l2(b1, b0))
    (a2, b2) = run(True)
    assert torch.equal(a1, a2) and torch.equal(b1, b2)       # deterministic


def test_pyramid_backward_unpool_variant_matches_default():
    """PCFA_PYRAMID_UNPOOL=1 (opt-in, slower: see corr_pyramid.hip): the backward products over the level-0 columns with
    dpyr un-pooled in the operand loader, in a child process (the switch is read once per process) against the default
    products over all slab columns: same mathematics, different rounding order: 2e-6 relative L2."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from pcfa_amd import hip_ops
g = torch.Generator().manual_seed(3)
f1 = torch.randn(1, 64, 24, 32, generator=g).cuda().requires_grad_(True)
f2 = torch.randn(1, 64, 24, 32, generator=g).cuda().requires_grad_(True)
ys, xs = torch.meshgrid(torch.arange(24), torch.arange(32), indexing="ij")
c = (torch.stack([xs, ys], 0).float()[None] + 2 * torch.randn(1, 2, 24, 32, generator=g)).cuda()
out = hip_ops.CorrBlock(f1, f2)(c)
out.backward(torch.randn(out.shape, generator=g).cuda())
torch.save((f1.grad.cpu(), f2.grad.cpu()), sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    grads = []
    for flag in ("0", "1"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            env = dict(os.environ, PCFA_PYRAMID_UNPOOL=flag)
            subprocess.run([sys.executable, "-c", code, f.name], check=True, env=env, timeout=300)
            grads.append(torch.load(f.name))
    for a, b in zip(*grads):
        assert rel_l2(b, a) < 2e-6, rel_l2(b, a)


def test_lookup_given_same_pyramid_is_tight(oracle_ops):
    """Feed the ORACLE's pyramid into the HIP lookup: isolates the lookup kernel from the GEMM."""
    from pcfa_amd import _hip
    lib = _hip.load()
    B, D, H, W = 1, 32, 24, 40
    gen = torch.Generator().manual_seed(1)
    f1, f2 = torch.randn(B, D, H, W, generator=gen), torch.randn(B, D, H, W, generator=gen)
    pyr_levels = oracle_ops.corr_pyramid(f1, f2, 4)
    slab = lib.pcfa_corr_slab_floats(H, W, 4)
    pyr = torch.zeros(B * H * W, slab)
    for (idx, h_l, w_l), lv in zip(hip_ops.tiled_index_maps(H, W, 4), pyr_levels):
        pyr[:, idx] = lv.reshape(B * H * W, -1)
    pyr = pyr.to(DEV)
    for spread in (0.0, 1.7, 25.0):
        coords = _grid(B, H, W) + spread * torch.randn(B, 2, H, W, generator=gen)
        want = oracle_ops.corr_lookup(pyr_levels, coords, 4)
        got = torch.empty(B, 324, H, W, device=DEV)
        st = lib.pcfa_corr_lookup_fwd(ctypes.c_void_p(pyr.data_ptr()), ctypes.c_void_p(coords.to(DEV).data_ptr()),
                                      ctypes.c_void_p(got.data_ptr()), B, H, W, 4, 4,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == 0
        assert max_abs(got, want) <= 2e-5 * float(want.abs().max())


def test_lookup_properties_full_size():
    """BASELINE size (55x128 features, D=256): linearity in the feature map and <bwd(g), .> adjointness."""
    B, D, H, W = 1, 256, 55, 128
    gen = torch.Generator().manual_seed(0)
    f1 = torch.randn(B, D, H, W, generator=gen).to(DEV)
    f2 = torch.randn(B, D, H, W, generator=gen).to(DEV)
    f2b = torch.randn(B, D, H, W, generator=gen).to(DEV)
    coords = (_grid(B, H, W) + 4 * torch.randn(B, 2, H, W, generator=gen)).to(DEV)
    la = hip_ops.CorrBlock(f1, f2)(coords)
    lb = hip_ops.CorrBlock(f1, f2b)(coords)
    lab = hip_ops.CorrBlock(f1, f2 + 2 * f2b)(coords)
    assert max_abs(lab, la + 2 * lb) <= 2e-5 * float(lab.abs().max())
    # adjoint: <lookup(f1,f2), g> differentiated w.r.t. f1 equals the analytic contraction
    f1r = f1.clone().requires_grad_(True)
    g = torch.randn(la.shape, generator=gen).to(DEV)
    out = hip_ops.CorrBlock(f1r, f2)(coords)
    (out * g).sum().backward()
    eps_dir = torch.randn(f1.shape, generator=gen).to(DEV)
    lhs = float((f1r.grad * eps_dir).sum())
    rhs = float((hip_ops.CorrBlock(eps_dir, f2)(coords) * g).sum())  # linear in f1
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs), 1.0)
    # determinism: two backward passes give identical bits
    f1s = f1.clone().requires_grad_(True)
    (hip_ops.CorrBlock(f1s, f2)(coords) * g).sum().backward()
    assert torch.equal(f1s.grad, f1r.grad)


def test_corr_block_edge_cases():
    with pytest.raises(ValueError):
        hip_ops.CorrBlock(torch.zeros(1, 8, 4, 4, device=DEV), torch.zeros(1, 8, 4, 4, device=DEV))  # level 3 empty
    # NaN / huge coordinates must not fault: everything out of range reads as zero
    f = torch.randn(1, 16, 16, 16, device=DEV)
    c = _grid(1, 16, 16).to(DEV)
    c[0, 0, 0, 0] = 1e30
    c[0, 1, 1, 1] = -1e30
    out = hip_ops.CorrBlock(f, f)(c)
    assert torch.isfinite(out[:, :, 2:, 2:]).all()
    assert float(out[0, :, 0, 0].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(1, 64, 16, 20), (2, 32, 23, 37), (1, 256, 55, 128)])
def test_lookup_convc1_fused_vs_unfused_and_oracle(oracle_ops, shape):
    """pcfa_lookup_convc1_fwd / _bwd (lookup -> convc1 -> ReLU in one launch per direction, reference
    models/raft/corr.py:29-50 + update.py:79-93) against (a) the un-fused HIP path lookup + F.conv2d (same taps bit for
    bit, so only the fp32 summation order of the 324-term dot product differs: 1e-5 of the output range, gradients 2e-5
    relative L2) and (b) the oracle on CPU.  Queries not a multiple of 32, two images, coordinates far outside."""
    import torch.nn.functional as F
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(H + W)
    f1c, f2c = torch.randn(B, D, H, W, generator=gen), torch.randn(B, D, H, W, generator=gen)
    wc = (torch.randn(256, 324, 1, 1, generator=gen) / 18.0)
    bc = 0.1 * torch.randn(256, generator=gen)
    coords = [_grid(B, H, W) + 2.5 * torch.randn(B, 2, H, W, generator=gen),
              _grid(B, H, W) + 40 * torch.randn(B, 2, H, W, generator=gen)]
    gos = [torch.randn(B, 256, H, W, generator=gen) for _ in coords]

    def run(block_cls, dev, fused, grad_outs):
        f1 = f1c.clone().to(dev).requires_grad_(True)
        f2 = f2c.clone().to(dev).requires_grad_(True)
        w, b = wc.to(dev), bc.to(dev)
        blk = block_cls(f1, f2)
        outs = []
        for c in coords:
            o = blk.lookup_conv_relu(c.to(dev), w, b, True) if fused else F.relu(F.conv2d(blk(c.to(dev)), w, b))
            assert o is not None
            outs.append(o)
        sum((o * g.to(dev)).sum() for o, g in zip(outs, grad_outs)).backward()
        return [o.detach().cpu() for o in outs], f1.grad.cpu(), f2.grad.cpu()

    # A pre-activation that sits at zero may round to either side in the two summation orders; one such flip moves a
    # whole gradient row (1e-3 relative L2 at D = 256).  The tight fused-vs-unfused gradient check therefore runs with
    # the output gradient zeroed at the (few) flipped positions, which the mask check below bounds.
    of0, _, _ = run(hip_ops.CorrBlock, DEV, True, gos)
    ou0, _, _ = run(hip_ops.CorrBlock, DEV, False, gos)
    gos_tie = [g * ((a > 0) == (b > 0)).float() for g, a, b in zip(gos, of0, ou0)]
    of, g1f, g2f = run(hip_ops.CorrBlock, DEV, True, gos_tie)
    ou, g1u, g2u = run(hip_ops.CorrBlock, DEV, False, gos_tie)
    oc, g1c, g2c = run(oracle_ops.CorrBlock, "cpu", False, gos_tie)
    omax = max(float(o.abs().max()) for o in oc)
    for a, b, c in zip(of, ou, oc):
        assert a.shape == c.shape
        assert max_abs(a, b) <= 1e-5 * omax, max_abs(a, b) / omax
        assert max_abs(a, c) <= 5e-5 * omax, max_abs(a, c) / omax
        assert float(((a > 0) != (b > 0)).float().mean()) < 1e-4      # the ReLU masks agree (ties aside)
    assert rel_l2(g1f, g1u) < 2e-5 and rel_l2(g2f, g2u) < 2e-5, (rel_l2(g1f, g1u), rel_l2(g2f, g2u))
    # vs the CPU oracle the pyramid itself differs in the last bits (MKL sgemm vs the MFMA fma chain), which flips the
    # ReLU of a few pre-activations that sit at zero: a handful of whole gradient rows appear / vanish (1e-3 at D = 256)
    assert rel_l2(g1f, g1c) < 5e-3 and rel_l2(g2f, g2c) < 5e-3, (rel_l2(g1f, g1c), rel_l2(g2f, g2c))
    # bit-reproducible (no atomics in the scatter)
    of2, g1f2, g2f2 = run(hip_ops.CorrBlock, DEV, True, gos_tie)
    assert all(torch.equal(a, b) for a, b in zip(of, of2)) and torch.equal(g1f, g1f2) and torch.equal(g2f, g2f2)


@pytest.mark.parametrize("gemm", ["lib", "hip"])
def test_gma_attention_ops_vs_torch(gemm):
    """SURVEY 8f row f1 (models/gma/gma.py:34-77 Attention, :79-115 Aggregate): similarity product + row softmax and
    the attention-times-value products on the hand-written fp32 MFMA GEMM, against torch in float64.
    N = 1000 (not a multiple of 128: ragged tiles, rows held in registers) and 2 heads."""
    # gemm (Config.gma_gemm): plain products on rocBLAS (default) or on pcfa_gemm_f32
    gen = torch.Generator().manual_seed(3)
    h, n, d = 2, 1000, 128
    q = torch.randn(1, h, n, d, generator=gen).to(DEV).requires_grad_(True)
    k = torch.randn(1, h, n, d, generator=gen).to(DEV).requires_grad_(True)
    vs = [torch.randn(1, h, n, d, generator=gen).to(DEV).requires_grad_(True) for _ in range(3)]
    gos = [torch.randn(1, h, n, d, generator=gen).to(DEV) for _ in range(3)]
    scale = d ** -0.5
    attn = hip_ops.attention_softmax(q, k, scale, gemm=gemm)
    share = hip_ops.AttnGradShare(gemm)
    outs = [hip_ops.attn_times_value(attn, v, share) for v in vs]
    sum((o * g).sum() for o, g in zip(outs, gos)).backward()
    qd, kd = q.detach().double().requires_grad_(True), k.detach().double().requires_grad_(True)
    vd = [v.detach().double().requires_grad_(True) for v in vs]
    attn_d = torch.softmax(scale * qd @ kd.transpose(-1, -2), dim=-1)
    outs_d = [attn_d @ v for v in vd]
    sum((o * g.double()).sum() for o, g in zip(outs_d, gos)).backward()
    assert max_abs(attn, attn_d) <= 2e-6 * float(attn_d.max())
    assert float((attn.sum(-1) - 1).abs().max()) < 1e-5
    for o, od in zip(outs, outs_d):
        assert rel_l2(o, od) < 2e-6
    for v, w in zip(vs, vd):
        assert rel_l2(v.grad, w.grad) < 2e-6
    assert rel_l2(q.grad, qd.grad) < 1e-5 and rel_l2(k.grad, kd.grad) < 1e-5
    # generic softmax fallback (columns not a multiple of 4) and the plain GEMM entry with every layout
    x = torch.randn(7, 1001, generator=gen).to(DEV)
    y = torch.empty_like(x)
    hip_ops._call("pcfa_softmax_rows_fwd", hip_ops._ptr(x), hip_ops._ptr(y), 7, 1001)
    assert max_abs(y, torch.softmax(x.double(), -1)) < 1e-6
    a, b = torch.randn(70, 50, generator=gen).to(DEV), torch.randn(50, 33, generator=gen).to(DEV)
    want = a.double() @ b.double()
    assert rel_l2(hip_ops.gemm_f32(a, b, 0, 1), want) < 1e-6
    assert rel_l2(hip_ops.gemm_f32(a, b.t().contiguous(), 0, 0), want) < 1e-6
    assert rel_l2(hip_ops.gemm_f32(a.t().contiguous(), b, 1, 1, alpha=0.5, splits=3), 0.5 * want) < 1e-6


# --------------------------------------------------------------------------- PWC cost volume
@pytest.mark.parametrize("tag", ["pwc_a", "pwc_b", "pwc_c", "gen_a", "gen_b"])
def test_spatial_corr_vs_reference_golden(tag):
    g = load_golden("spatial_corr_" + tag)
    ks, ps, st, pad, dil, dp = (int(v) for v in g["params"])
    a, b = t(g["in1"], DEV).requires_grad_(True), t(g["in2"], DEV).requires_grad_(True)
    out = hip_ops.spatial_correlation_sample(a, b, ks, ps, st, pad, dil, dp)
    assert out.shape == g["out"].shape
    C = a.shape[1]
    assert max_abs(out, t(g["out"])) <= 1e-6 * C * float(np.abs(g["out"]).max()) + 1e-6
    out.backward(t(g["grad_out"], DEV))
    assert rel_l2(a.grad, t(g["gin1"])) < 1e-5
    assert rel_l2(b.grad, t(g["gin2"])) < 1e-5


@pytest.mark.parametrize("shape", [(1, 196, 6, 20), (1, 128, 12, 40), (2, 96, 24, 80), (1, 64, 48, 160),
                                   (1, 32, 96, 320), (1, 5, 7, 33)])
def test_spatial_corr_pwc_levels_vs_oracle(oracle_ops, shape):
    """The five PWC-Net level shapes at KITTI size + one ragged shape."""
    gen = torch.Generator().manual_seed(shape[1])
    a = torch.randn(*shape, generator=gen).requires_grad_(True)
    b = torch.randn(*shape, generator=gen).requires_grad_(True)
    want = oracle_ops.spatial_correlation_sample(a, b, 1, 9, 1)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    ag, bg = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    got = hip_ops.spatial_correlation_sample(ag, bg, kernel_size=1, patch_size=9, stride=1)
    assert max_abs(got, want) <= 2e-6 * shape[1] * float(want.abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(ag.grad, a.grad) < 1e-5 and rel_l2(bg.grad, b.grad) < 1e-5
    # determinism
    ag2, bg2 = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    hip_ops.spatial_correlation_sample(ag2, bg2, 1, 9, 1).backward(go.to(DEV))
    assert torch.equal(ag2.grad, ag.grad) and torch.equal(bg2.grad, bg.grad)


@pytest.mark.parametrize("shape", [(1, 196, 6, 20), (1, 128, 12, 40), (2, 96, 24, 80), (1, 64, 48, 160),
                                   (1, 32, 96, 320), (1, 32, 112, 256), (1, 5, 7, 36), (2, 33, 3, 8), (1, 3, 1, 4)])
def test_pwc_cost_volume_fused_vs_oracle(oracle_ops, shape):
    """leaky_relu(correlate(a, b)) as ONE launch per direction (pcfa_cost_volume9_fwd/bwd; PWCNet.py:45-58 + the
    LeakyReLU at :249...) against the oracle's composition: the five KITTI level shapes, the Sintel level-2 shape,
    ragged tiles, channel counts that are not a multiple of the 32-channel backward group.  Forward 2e-6 * max|out|
    (summation order over C), gradients 1e-5 relative L2, bit-reproducible."""
    gen = torch.Generator().manual_seed(shape[1] + shape[2])
    a = torch.randn(*shape, generator=gen).requires_grad_(True)
    b = torch.randn(*shape, generator=gen).requires_grad_(True)
    want = oracle_ops.pwc_cost_volume(a, b, 0.1)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    ag, bg = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    got = hip_ops.pwc_cost_volume(ag, bg, 0.1)
    assert got.shape == want.shape
    assert max_abs(got, want) <= 2e-6 * float(want.abs().max()) * max(1.0, np.sqrt(shape[1]))
    got.backward(go.to(DEV))
    assert rel_l2(ag.grad, a.grad) < 1e-5 and rel_l2(bg.grad, b.grad) < 1e-5
    ag2, bg2 = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    got2 = hip_ops.pwc_cost_volume(ag2, bg2, 0.1)
    got2.backward(go.to(DEV))
    assert torch.equal(got2, got) and torch.equal(ag2.grad, ag.grad) and torch.equal(bg2.grad, bg.grad)


# --------------------------------------------------------------------------- FlowNet2's three operators
# Oracle = restatement of the CUDA kernels (no reference output exists: CUDA-only extensions), correlation
# cross-pinned against the reference's C++ sampler in tests/test_oracle_cpu.py.  Tolerances: correlation
# 2e-6 * C * max|out| (summation order over C), gradients 1e-5 relative L2; Resample2d forward 1e-6 * max|img|
# (same operation order, fma contraction aside), backward 1e-5 relative L2 (atomic arrival order); ChannelNorm 2e-7.
@pytest.mark.parametrize("shape", [(1, 32, 24, 40), (2, 19, 13, 37), (1, 8, 8, 32), (1, 256, 56, 128)])
def test_flownet_corr_fast_path_vs_oracle(oracle_ops, shape):
    """FlowNetC's configuration (FlowNetC.py:31-35): ragged tiles, odd width (scalar stores), channel tail, and
    the full 448x1024 shape (256 x 56 x 128)."""
    gen = torch.Generator().manual_seed(shape[1] + shape[3])
    a = torch.randn(*shape, generator=gen).requires_grad_(True)
    b = torch.randn(*shape, generator=gen).requires_grad_(True)
    want = oracle_ops.flownet_correlation(a, b, 20, 1, 20, 1, 2)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    ag, bg = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    got = hip_ops.flownet_correlation(ag, bg, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2)
    assert got.shape == want.shape
    assert max_abs(got, want) <= 2e-6 * shape[1] * float(want.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(ag.grad, a.grad) < 1e-5 and rel_l2(bg.grad, b.grad) < 1e-5
    ag2, bg2 = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    got2 = hip_ops.flownet_correlation(ag2, bg2, 20, 1, 20, 1, 2)
    got2.backward(go.to(DEV))
    assert torch.equal(got2, got) and torch.equal(ag2.grad, ag.grad) and torch.equal(bg2.grad, bg.grad)


@pytest.mark.parametrize("params", [(4, 1, 4, 1, 2), (5, 3, 4, 1, 2), (3, 3, 2, 1, 1), (2, 1, 4, 1, 2),
                                    (6, 1, 4, 1, 2), (20, 1, 20, 1, 4)])
def test_flownet_corr_generic_path_vs_oracle(oracle_ops, params):
    gen = torch.Generator().manual_seed(sum(params))
    a = torch.randn(2, 6, 12, 14, generator=gen).requires_grad_(True)
    b = torch.randn(2, 6, 12, 14, generator=gen).requires_grad_(True)
    want = oracle_ops.flownet_correlation(a, b, *params)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    ag, bg = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    got = hip_ops.flownet_correlation(ag, bg, *params)
    assert got.shape == want.shape
    assert max_abs(got, want) <= 2e-6 * 6 * 9 * float(want.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(ag.grad, a.grad) < 1e-5 and rel_l2(bg.grad, b.grad) < 1e-5


def test_flownet_corr_argument_errors():
    x = torch.randn(1, 4, 8, 8, device=DEV)
    with pytest.raises(RuntimeError):
        hip_ops.flownet_correlation(x, x, 0, 2, 4, 1, 2)        # even kernel_size
    with pytest.raises(RuntimeError):
        hip_ops.flownet_correlation(x, x, 0, 1, 20, 1, 2)       # empty output
    y = hip_ops.flownet_correlation(x.requires_grad_(True), x, 4, 1, 4, 2, 2)  # stride1 = 2: forward only
    assert y.shape == (1, 25, 4, 4)
    with pytest.raises(RuntimeError):
        y.sum().backward()
    with pytest.raises(RuntimeError):
        hip_ops.flownet_correlation(x.cpu(), x.cpu(), 4, 1, 4, 1, 2)


@pytest.mark.parametrize("shape,scale", [((2, 3, 9, 11), 4.0), ((1, 3, 64, 96), 8.0), ((1, 3, 448, 1024), 20.0),
                                         ((1, 5, 17, 33), 0.4)])
def test_resample2d_vs_oracle(oracle_ops, shape, scale):
    gen = torch.Generator().manual_seed(shape[2])
    B, C, H, W = shape
    img = torch.randn(*shape, generator=gen).requires_grad_(True)
    flow = (scale * torch.randn(B, 2, H, W, generator=gen)).requires_grad_(True)
    want = oracle_ops.resample2d(img, flow)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    ig, fg = img.detach().to(DEV).requires_grad_(True), flow.detach().to(DEV).requires_grad_(True)
    got = hip_ops.resample2d(ig, fg)
    assert max_abs(got, want) <= 1e-6 * float(img.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(ig.grad, img.grad) < 1e-5 and rel_l2(fg.grad, flow.grad) < 1e-5
    near = hip_ops.resample2d(ig.detach(), fg.detach(), 1, False)
    assert torch.equal(near.cpu(), oracle_ops.resample2d(img.detach(), flow.detach(), 1, False))
    # a zero flow is the identity, an integer shift is a clamped shift
    assert torch.equal(hip_ops.resample2d(ig.detach(), torch.zeros_like(fg)), ig.detach())
    with pytest.raises(RuntimeError):
        hip_ops.resample2d(ig.detach(), fg.detach(), 3, True)  # kernel_size > 1 is out of bounds in the reference


@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 2, 448, 1024), (1, 3, 448, 1024)])
def test_channelnorm_vs_oracle(oracle_ops, shape):
    gen = torch.Generator().manual_seed(shape[3])
    x = torch.randn(*shape, generator=gen)
    x[0, :, 0, 0] = 0.
    xc = x.clone().requires_grad_(True)
    want = oracle_ops.channelnorm(xc)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    xg = x.to(DEV).requires_grad_(True)
    got = hip_ops.channelnorm(xg)
    assert got.shape == want.shape
    assert max_abs(got, want) <= 2e-7 * float(want.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(xg.grad, xc.grad) < 1e-6
    assert float(xg.grad[0, :, 0, 0].abs().max()) == 0.0


def test_conv3x3_cat_vs_oracle(oracle_ops):
    """The motion encoder's two concatenations (update.py:91-101): cat(relu(convc2(.)), relu(convf2(.))) and
    cat(relu(conv(.)), flow) written in place by the convolutions."""
    gen = torch.Generator().manual_seed(91)
    H, W = 55, 128
    a = torch.randn(1, 256, H, W, generator=gen).requires_grad_(True)
    b = torch.randn(1, 128, H, W, generator=gen).requires_grad_(True)
    flow = torch.randn(1, 2, H, W, generator=gen)
    wa, ba = torch.randn(192, 256, 3, 3, generator=gen) / 48, 0.1 * torch.randn(192, generator=gen)
    wb, bb = torch.randn(64, 128, 3, 3, generator=gen) / 34, 0.1 * torch.randn(64, generator=gen)
    wc, bc = torch.randn(126, 256, 3, 3, generator=gen) / 48, 0.1 * torch.randn(126, generator=gen)
    cf = oracle_ops.conv3x3_cat([(a, wa, ba), (b, wb, bb)])
    want = oracle_ops.conv3x3_cat([(cf, wc, bc)], (flow,))
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    ag, bg = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    d = lambda t: t.to(DEV)  # noqa: E731
    cfg = hip_ops.conv3x3_cat([(ag, d(wa), d(ba)), (bg, d(wb), d(bb))])
    got = hip_ops.conv3x3_cat([(cfg, d(wc), d(bc))], (d(flow),))
    assert got.shape == want.shape == (1, 128, H, W)
    assert max_abs(got, want) <= 1e-5 * float(want.detach().abs().max())
    assert torch.equal(got[:, 126:].cpu(), flow)
    got.backward(go.to(DEV))
    # two ReLU layers in a row: outputs within rounding of zero flip their mask bit, so the gradient's tolerance is
    # set by the number of flipped bits (F(4x4,3x3): 1.4e-4, F(2x2,3x3): 6e-5), not by the convolution's own error
    assert rel_l2(ag.grad, a.grad) < 5e-4 and rel_l2(bg.grad, b.grad) < 5e-4


@pytest.mark.parametrize("shape", [(2, 64, 64, 40, 48), (1, 12, 12, 9, 21),
                                   (1, 32, 32, 120, 256), (2, 32, 64, 120, 256)])   # F(4x4,3x3): channel-split / direct
def test_conv3x3_skip_gradient_sums_in_the_epilogue(shape):
    """conv3x3(.., skip=True): the gradient arriving on the residual alias is added by the data-gradient kernel
    (pcfa_conv3x3_fused_bwd) -- against the same graph with autograd's own accumulation: one fp32 add either way, so
    bit-identical; also with only one of the two outputs used."""
    B, K, N, H, W = shape
    gen = torch.Generator().manual_seed(K + W)
    x0 = torch.randn(B, K, H, W, generator=gen).to(DEV)
    w = (torch.randn(N, K, 3, 3, generator=gen) / (9 * K) ** .5).to(DEV)
    b = torch.randn(N, generator=gen).to(DEV)
    go, gs = torch.randn(B, N, H, W, generator=gen).to(DEV), torch.randn(B, K, H, W, generator=gen).to(DEV)

    def run(skip, use_y=True, use_s=True):
        x = x0.clone().requires_grad_(True)
        if skip:
            y, xs = hip_ops.conv3x3(x, w, b, True, skip=True)
        else:
            y, xs = hip_ops.conv3x3(x, w, b, True), x
        loss = (y * go).sum() * float(use_y) + (xs * gs).sum() * float(use_s)
        (gx,) = torch.autograd.grad(loss, x)
        return y.detach(), gx

    for use_y, use_s in ((True, True), (True, False)):
        (ya, ga), (yb, gb) = run(True, use_y, use_s), run(False, use_y, use_s)
        assert torch.equal(ya, yb) and torch.equal(ga, gb)


def test_context_encoder_block_deferred_masks_change_no_bit():
    """A folded-BatchNorm ResidualBlock (context encoder, extractor.py:23-58) with its three ReLU backward passes riding in
    neighbouring kernels (pcfa_conv3x3_fused_bwd mask + addend, pcfa_relu_bwd2) against one launch per ReLU and autograd's
    own residual add: masks multiply by exactly 0 or 1 and the residual sum is one fp32 add either way: bit-identical."""
    import dataclasses
    from pcfa_amd import config
    from pcfa_amd.nets import raft as raft_mod
    torch.manual_seed(5)
    blk = raft_mod.ResidualBlock(64, 64, norm_fn="batch", stride=1).to(DEV).eval()
    for m in (blk.norm1, blk.norm2):
        m.running_mean.normal_(0, 0.1)
        m.running_var.uniform_(0.5, 1.5)
    for p_ in blk.parameters():
        p_.requires_grad_(False)
    x0 = torch.randn(1, 64, 40, 48, device=DEV)
    go = torch.randn(1, 64, 40, 48, device=DEV)

    def run(defer):
        config.attach(blk, dataclasses.replace(config.DEFAULT, defer_relu=defer))   # as import_and_load(config=...) does
        x = x0.clone().requires_grad_(True)
        out = blk(x)   # the package's operator table is hip_ops
        (gx,) = torch.autograd.grad((out * go).sum(), x)
        return out.detach(), gx

    (oa, ga), (ob, gb) = run(True), run(False)
    assert torch.equal(oa, ob) and torch.equal(ga, gb) and float(ga.abs().max()) > 0


@pytest.mark.parametrize("shape", [(1, 55, 128), (2, 7, 70), (1, 3, 5)])
def test_convex_upsample_vs_oracle(oracle_ops, shape):
    """pcfa_convex_upsample_fwd / _bwd (models/raft/raft.py:72-83) against the reference's tensor expression: same
    softmax form (max, exp, sum, divide) and 9-term sums in k order: 1e-6 of the output range, gradients 2e-6 relative
    L2; columns not a multiple of 64, maps smaller than the 3x3 window's reach, two images."""
    N, H, W = shape
    gen = torch.Generator().manual_seed(H * 7 + W)
    flow = 3 * torch.randn(N, 2, H, W, generator=gen)
    mask = 2 * torch.randn(N, 576, H, W, generator=gen)
    go = torch.randn(N, 2, 8 * H, 8 * W, generator=gen)

    def run(mod, dev):
        f, m = flow.clone().to(dev).requires_grad_(True), mask.clone().to(dev).requires_grad_(True)
        out = mod.convex_upsample(f, m)
        out.backward(go.to(dev))
        return out.detach().cpu(), f.grad.cpu(), m.grad.cpu()

    (ow, fw, mw), (og, fg, mg) = run(oracle_ops, "cpu"), run(hip_ops, DEV)
    assert og.shape == ow.shape
    assert max_abs(og, ow) <= 1e-6 * float(ow.abs().max())
    assert rel_l2(fg, fw) < 2e-6 and rel_l2(mg, mw) < 2e-6, (rel_l2(fg, fw), rel_l2(mg, mw))
    og2, fg2, mg2 = run(hip_ops, DEV)
    assert torch.equal(og, og2) and torch.equal(fg, fg2) and torch.equal(mg, mg2)   # no atomics


def test_deferred_relu_masks_change_no_bit():
    """Motion encoder + GRU update with the ReLU backward of convc2 / convf2 / conv deferred into the kernels that
    produce those gradients anyway (conv3x3_cat flags, gru_step rest_relu_channels; pcfa_conv3x3_masked_fwd,
    pcfa_sepconv5_fwd_split_masked) against the same graph with one ReLU launch per layer: a mask multiplies by exactly
    0 or 1, so every gradient must be bit-identical."""
    gen = torch.Generator().manual_seed(17)
    H, W, C = 55, 128, 128
    rnd = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    a, b, flow = rnd(1, 256, H, W), rnd(1, 128, H, W), rnd(1, 2, H, W)
    wa, ba = rnd(192, 256, 3, 3) / 48, 0.1 * rnd(192)
    wb, bb = rnd(64, 128, 3, 3) / 34, 0.1 * rnd(64)
    wc, bc = rnd(126, 256, 3, 3) / 48, 0.1 * rnd(126)
    h = torch.tanh(rnd(1, C, H, W))
    halves = []
    for k in ((1, 5), (5, 1)):
        sc = (5 * (2 * C)) ** -.5
        halves.append((rnd(2 * C, 2 * C, *k) * sc, rnd(1, 2 * C, H, W), rnd(C, 2 * C, *k) * sc, rnd(1, C, H, W)))
    go = rnd(1, C, H, W)
    d = lambda t: t.to(DEV)  # noqa: E731

    def run(defer):
        ag, bg, hg = (t.detach().to(DEV).requires_grad_(True) for t in (a, b, h))
        cf = hip_ops.conv3x3_cat([(ag, d(wa), d(ba)), (bg, d(wb), d(bb))], grad_premasked=defer)
        mf = hip_ops.conv3x3_cat([(cf, d(wc), d(bc))], (d(flow),), grad_premasked=defer, mask_input_grads=defer)
        out = hip_ops.gru_step(hg, mf, tuple(tuple(d(t) for t in hf) for hf in halves), 126 if defer else 0)
        out.backward(d(go))
        return out.detach(), ag.grad, bg.grad, hg.grad

    for x, y in zip(run(False), run(True)):
        assert torch.equal(x, y)
    assert float(run(True)[1].abs().max()) > 0


# --------------------------------------------------------------------------- PWC-Net dense decoder block
@pytest.mark.parametrize("shape", [(1, 115, 12, 40), (1, 81, 6, 20), (1, 21, 7, 9)])
def test_dense_block_vs_oracle(oracle_ops, shape):
    """x = cat(leaky(conv_i(x)), x) five times (PWCNet.py:234-323) in one pre-allocated buffer vs the torch.cat loop.
    Winograd F(2x2,3x3) in fp32: 1e-5 relative on outputs, 1e-4 relative L2 on the input gradient."""
    gen = torch.Generator().manual_seed(shape[1])
    B, K, H, W = shape
    x = torch.randn(*shape, generator=gen).requires_grad_(True)
    layers, k = [], K
    for n in (128, 128, 96, 64, 32):
        layers.append((torch.randn(n, k, 3, 3, generator=gen) / (3 * k ** 0.5), 0.1 * torch.randn(n, generator=gen)))
        k += n
    want = oracle_ops.dense_block(x, layers, 0.1)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    xg = x.detach().to(DEV).requires_grad_(True)
    got = hip_ops.dense_block(xg, [(w.to(DEV), b.to(DEV)) for w, b in layers], 0.1)
    assert got.shape == want.shape
    assert max_abs(got, want) <= 1e-5 * float(want.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(xg.grad, x.grad) < 1e-4
    with pytest.raises(ValueError):
        hip_ops.dense_block(torch.cat([xg.detach(), xg.detach()]), [(w.to(DEV), b.to(DEV)) for w, b in layers], 0.1)
    # the input as the parts of the caller's concatenation (PWCNet.py:265: cat((corr, c1, up_flow, up_feat), 1)): written
    # into the block buffer directly, gradients handed back as views of the running gradient -- the same bits
    if K > 4:
        cuts = [0, K - 4, K - 2, K]
        parts = [x.detach()[:, a:b].clone().to(DEV).requires_grad_(True) for a, b in zip(cuts[:-1], cuts[1:])]
        got2 = hip_ops.dense_block(tuple(parts), [(w.to(DEV), b.to(DEV)) for w, b in layers], 0.1)
        assert torch.equal(got2, got)
        got2.backward(go.to(DEV))
        assert torch.equal(torch.cat([p.grad for p in parts], 1), xg.grad)
        grad_seen = go.to(DEV).clone()
        got3 = hip_ops.dense_block(xg.detach().requires_grad_(True), [(w.to(DEV), b.to(DEV)) for w, b in layers], 0.1)
        got3.backward(grad_seen)
        assert torch.equal(grad_seen, go.to(DEV))     # the incoming gradient is read, never written


def test_pwcnet_batch_of_pairs_matches_single_pairs():
    """PWC-Net on a batch of two pairs (the universal attack's local batch, attack_PCFA.py:344-350) against the same pairs
    one at a time: the batched pyramid (4 images per layer), the decoder's generic path (the dense-block buffer is a
    batch-1 layout) and the folded glue must give the per-pair flows and input gradients -- to rounding, the batch sizes
    pick other convolution algorithms."""
    from pcfa_amd.nets.pwcnet import PWCDCNet
    torch.manual_seed(11)
    net = PWCDCNet().to(DEV).eval()
    for p in net.parameters():
        p.requires_grad = False
    a = torch.rand(2, 3, 128, 192, device=DEV, requires_grad=True)
    b = torch.rand(2, 3, 128, 192, device=DEV, requires_grad=True)
    go = torch.randn(2, 2, 128, 192, device=DEV)
    flow = net(a, b)
    (flow * go).sum().backward()
    for i in range(2):
        ai = a.detach()[i:i + 1].clone().requires_grad_(True)
        bi = b.detach()[i:i + 1].clone().requires_grad_(True)
        fi = net(ai, bi)
        (fi * go[i:i + 1]).sum().backward()
        assert max_abs(flow.detach()[i:i + 1], fi.detach()) <= 1e-4 * float(fi.detach().abs().max())
        assert rel_l2(a.grad[i:i + 1], ai.grad) < 2e-3 and rel_l2(b.grad[i:i + 1], bi.grad) < 2e-3


def test_split_batch_views_and_one_concatenation():
    """ops.split_batch: the halves of a feature tensor computed for both images at once (nets/pwcnet.py) are views; the
    backward assembles their gradients with one concatenation, also when only one half is used."""
    x = torch.randn(4, 3, 5, 7, device=DEV, requires_grad=True)
    a, b = hip_ops.split_batch(x, 1)
    assert a.shape[0] == 1 and b.shape[0] == 3 and a.data_ptr() == x.data_ptr()
    ga, gb = torch.randn_like(a), torch.randn_like(b)
    ((a * ga).sum() + (b * gb).sum()).backward()
    assert torch.equal(x.grad, torch.cat((ga, gb), 0))
    x.grad = None
    a, b = hip_ops.split_batch(x, 2)
    (b * 2.0).sum().backward()
    assert torch.equal(x.grad[:2], torch.zeros_like(x.grad[:2])) and torch.equal(x.grad[2:], torch.full_like(x.grad[2:], 2.0))


# --------------------------------------------------------------------------- PWC-Net warp
@pytest.mark.parametrize("shape,scale", [((1, 128, 12, 40), 1.5), ((1, 96, 24, 80), 3.0), ((2, 64, 48, 160), 6.0),
                                         ((1, 32, 96, 320), 12.0), ((1, 5, 7, 9), 4.0)])
def test_pwc_warp_vs_oracle(oracle_ops, shape, scale):
    """PWCDCNet.warp at the four KITTI-size levels (+ a ragged shape) vs the reference's own statement sequence on
    the CPU (grid_sample x2, mask, multiply).  Flows reach outside the image (zero padding, mask = 0).  Tolerance:
    the coordinate arithmetic is the same fp32 sequence, products may be fused: 2e-6 * max|x|; gradients 2e-5."""
    gen = torch.Generator().manual_seed(shape[2])
    B, C, H, W = shape
    x = torch.randn(*shape, generator=gen).requires_grad_(True)
    flo = (scale * torch.randn(B, 2, H, W, generator=gen)).requires_grad_(True)
    want = oracle_ops.pwc_warp(x, flo)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    xg, fg = x.detach().to(DEV).requires_grad_(True), flo.detach().to(DEV).requires_grad_(True)
    got = hip_ops.pwc_warp(xg, fg)
    assert max_abs(got, want) <= 2e-6 * float(x.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(xg.grad, x.grad) < 2e-5 and rel_l2(fg.grad, flo.grad) < 2e-5
    # the default backward scatters fixed-point values (integer adds): bit-reproducible from run to run, and equal
    # to the fp32-atomics form (pcfa_pwc_warp_bwd) up to the rounding of the addends
    g1x, g1f = xg.grad.clone(), fg.grad.clone()
    for _ in range(2):
        xg.grad = fg.grad = None
        hip_ops.pwc_warp(xg, fg).backward(go.to(DEV))
        assert torch.equal(xg.grad, g1x) and torch.equal(fg.grad, g1f)
    xg.grad = fg.grad = None
    hip_ops.pwc_warp(xg, fg, deterministic=False).backward(go.to(DEV))   # Config.warp_bwd_deterministic = False
    assert rel_l2(xg.grad, g1x) < 1e-6 and rel_l2(fg.grad, g1f) < 1e-6
    # the fixed point is scaled per call, so gradients of any absolute size keep fp32's resolution -- the
    # AEE / npix-scaled gradients of the deep PWC levels (1e-9) and huge ones alike (an absolute 2^-40 unit flushed
    # addends below 4.5e-13 and overflowed above 8.4e6)
    for mag in (1e-9, 1e9):
        xg.grad = fg.grad = None
        hip_ops.pwc_warp(xg, fg).backward((mag * go).to(DEV))
        assert rel_l2(xg.grad / mag, x.grad) < 2e-5 and rel_l2(fg.grad / mag, flo.grad) < 2e-5, mag
        assert rel_l2(xg.grad / mag, g1x) < 1e-6
    # zero flow: the align_corners mismatch of the original code samples at x * W / (W - 1) - 0.5, not at x
    ident = hip_ops.pwc_warp(xg.detach(), torch.zeros_like(fg))
    assert max_abs(ident, oracle_ops.pwc_warp(x.detach(), torch.zeros_like(flo))) <= 2e-6 * float(x.detach().abs().max())
    # flow_scale: `self.warp(c2, up_flow * 0.625)` (PWCNet.py:262) in the kernel = the two element-wise launches, bit for bit
    for det in (True, False):
        for fs in (0.625, 5.0):
            xa, fa = x.detach().to(DEV).requires_grad_(True), flo.detach().to(DEV).requires_grad_(True)
            xb, fb = x.detach().to(DEV).requires_grad_(True), flo.detach().to(DEV).requires_grad_(True)
            ya = hip_ops.pwc_warp(xa, fa, deterministic=det, flow_scale=fs)
            yb = hip_ops.pwc_warp(xb, fb * fs, deterministic=det)
            assert torch.equal(ya, yb)
            ya.backward(go.to(DEV))
            yb.backward(go.to(DEV))
            if det:
                assert torch.equal(xa.grad, xb.grad) and torch.equal(fa.grad, fb.grad)
            else:   # hardware atomics: order-dependent roundings
                assert rel_l2(xa.grad, xb.grad) < 1e-6 and rel_l2(fa.grad, fb.grad) < 1e-6
    ws = oracle_ops.pwc_warp(x.detach(), flo.detach(), flow_scale=0.625)
    assert max_abs(hip_ops.pwc_warp(xg.detach(), fg.detach(), flow_scale=0.625), ws) <= 2e-6 * float(x.detach().abs().max())


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf")])
def test_pwc_warp_deterministic_backward_propagates_non_finite_gradients(bad):
    """the fixed-point scatter has no image of Inf / NaN (fmaxf drops NaN, the int64 conversion saturates), so
    a non-finite grad_out used to come back as FINITE garbage.  grid_sample's backward (PWCNet.py:193) lets the optimiser
    see such a fault; the deterministic path now flags the call and returns NaN in both gradients (a superset of the
    reference's poisoned taps), and a finite call right after it is unaffected."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 8, 40, 72, generator=g).to(DEV).requires_grad_(True)
    flo = (2.0 * torch.randn(1, 2, 40, 72, generator=g)).to(DEV).requires_grad_(True)
    go = torch.randn(1, 8, 40, 72, generator=g).to(DEV)
    good = torch.autograd.grad(hip_ops.pwc_warp(x, flo, deterministic=True), (x, flo), go)
    assert all(bool(torch.isfinite(t).all()) for t in good)
    poisoned = go.clone()
    poisoned[0, 3, 17, 29] = bad
    gx, gf = torch.autograd.grad(hip_ops.pwc_warp(x, flo, deterministic=True), (x, flo), poisoned)
    assert bool(torch.isnan(gx).all()) and bool(torch.isnan(gf).all())
    again = torch.autograd.grad(hip_ops.pwc_warp(x, flo, deterministic=True), (x, flo), go)
    assert torch.equal(again[0], good[0]) and torch.equal(again[1], good[1])


def test_pwc_warp_scatter_through_lds_window_moves_no_bit():
    """The deterministic warp backward scatters through a per-workgroup LDS window (r04: 123 -> 28 us on the 32 x 96 x 320
    level); PCFA_WARP_SCATTER=global is the r03 form, one global atomic per tap.  Both add the same fixed-point integers,
    so the gradients must be IDENTICAL -- smooth, textured and tearing flows (taps outside the window take the global
    path), ragged sizes, batch 2.  The switch is read once per process: the other form runs in a child."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from pcfa_amd import hip_ops
def chk(t): return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())
out = []
for shape, scale in (((1, 32, 96, 320), 6.0), ((2, 24, 40, 72), 3.0), ((1, 7, 33, 70), 40.0), ((1, 16, 50, 35), 1.0)):
    g = torch.Generator().manual_seed(shape[1])
    B, C, H, W = shape
    x = torch.randn(*shape, generator=g).cuda().requires_grad_(True)
    go = torch.randn(*shape, generator=g).cuda()
    for kind in range(3):
        base = torch.nn.functional.interpolate(scale * torch.randn(B, 2, max(H // 8, 1), max(W // 8, 1), generator=g),
                                               size=(H, W), mode="bilinear", align_corners=False)
        f = (base, base + 0.3 * torch.randn(B, 2, H, W, generator=g), scale * torch.randn(B, 2, H, W, generator=g))[kind]
        flo = f.contiguous().cuda().requires_grad_(True)
        gx, gf = torch.autograd.grad(hip_ops.pwc_warp(x, flo, flow_scale=1.25), (x, flo), go)
        out.append((chk(gx), chk(gf), float(gx.abs().max())))
print("CHK", out)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for env in ({}, {"PCFA_WARP_SCATTER": "global"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("CHK ")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        res.append(line[-1])
    assert res[0] == res[1]
    assert "0.0)" not in res[0]    # (the gradients are not all zero)


# --------------------------------------------------------------------------- flow-prediction convolutions
@pytest.mark.parametrize("shape,n", [((1, 256, 55, 128), 2), ((2, 37, 9, 13), 2), ((1, 1026, 14, 32), 2),
                                     ((1, 5, 3, 70), 1), ((2, 16, 17, 5), 3), ((1, 64, 24, 40), 4),
                                     ((1, 1024, 1, 2), 2), ((2, 9, 1, 1), 2), ((1, 8, 1, 3), 2), ((1, 6, 3, 1), 2)])
def test_conv3x3_fewout_vs_oracle(oracle_ops, shape, n):
    """FlowHead.conv2 at 440x1024 / 8, FlowNet2's predict_flow5 shape, ragged sizes, 1..4 output channels, with
    and without bias.  Tolerance: summation order over K*9 products (2e-6 * sqrt(9K) * max|out|)."""
    gen = torch.Generator().manual_seed(shape[1] + n)
    B, K, H, W = shape
    x = torch.randn(*shape, generator=gen).requires_grad_(True)
    w = torch.randn(n, K, 3, 3, generator=gen) / (3 * K ** 0.5)
    bias = torch.randn(n, generator=gen) if K % 2 else None
    want = oracle_ops.conv3x3_fewout(x, w, bias)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    xg = x.detach().to(DEV).requires_grad_(True)
    got = hip_ops.conv3x3_fewout(xg, w.to(DEV), None if bias is None else bias.to(DEV))
    assert got.shape == want.shape
    assert max_abs(got, want) <= 2e-6 * (9 * K) ** 0.5 * float(want.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(xg.grad, x.grad) < 1e-5
    xg2 = x.detach().to(DEV).requires_grad_(True)
    got2 = hip_ops.conv3x3_fewout(xg2, w.to(DEV), None if bias is None else bias.to(DEV))
    got2.backward(go.to(DEV))
    assert torch.equal(got2, got) and torch.equal(xg2.grad, xg.grad)
    with pytest.raises(ValueError):
        hip_ops.conv3x3_fewout(xg.detach(), torch.zeros(5, K, 3, 3, device=DEV))
    # skip=True: x's other consumer reads the alias, and its gradient is added inside the data-gradient kernel
    xg3 = x.detach().to(DEV).requires_grad_(True)
    got3, alias = hip_ops.conv3x3_fewout(xg3, w.to(DEV), None if bias is None else bias.to(DEV), skip=True)
    other = torch.randn(shape, generator=gen).to(DEV)
    (got3 * go.to(DEV)).sum().add((alias * other).sum()).backward()
    assert torch.equal(got3, got) and torch.equal(xg3.grad, xg.grad + other)
    xg4 = x.detach().to(DEV).requires_grad_(True)
    _, alias = hip_ops.conv3x3_fewout(xg4, w.to(DEV), None, skip=True)
    (alias * other).sum().backward()                 # only the alias is used: its gradient passes through
    assert torch.equal(xg4.grad, other)


@pytest.mark.parametrize("shape,n", [((1, 597, 48, 160), 2), ((1, 2, 48, 160), 2), ((1, 529, 6, 20), 2),
                                     ((1, 661, 12, 40), 2), ((2, 37, 9, 13), 2), ((1, 5, 3, 70), 1), ((2, 16, 17, 5), 3),
                                     ((1, 64, 24, 40), 4), ((1, 1024, 1, 2), 2), ((2, 9, 1, 1), 2), ((1, 6, 3, 1), 2)])
def test_deconv4s2_fewout_vs_oracle(oracle_ops, shape, n):
    """PWC-Net's deconv / upfeat layers (PWCNet.py:42-43: ConvTranspose2d(K, 2, 4, 2, 1)) at the KITTI level shapes,
    ragged sizes, 1..4 output channels, with and without bias, small planes with split channel sums.  Tolerance:
    summation order over 4K products per output; bit-reproducible."""
    gen = torch.Generator().manual_seed(shape[1] + 7 * n)
    B, K, H, W = shape
    x = torch.randn(*shape, generator=gen).requires_grad_(True)
    w = torch.randn(K, n, 4, 4, generator=gen) / (2 * K ** 0.5)
    bias = torch.randn(n, generator=gen) if K % 2 else None
    want = oracle_ops.deconv4s2_fewout(x, w, bias)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    xg = x.detach().to(DEV).requires_grad_(True)
    got = hip_ops.deconv4s2_fewout(xg, w.to(DEV), None if bias is None else bias.to(DEV))
    assert got.shape == want.shape == (B, n, 2 * H, 2 * W)
    assert max_abs(got, want) <= 2e-6 * (4 * K) ** 0.5 * float(want.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(xg.grad, x.grad) < 1e-5
    xg2 = x.detach().to(DEV).requires_grad_(True)
    got2 = hip_ops.deconv4s2_fewout(xg2, w.to(DEV), None if bias is None else bias.to(DEV))
    got2.backward(go.to(DEV))
    assert torch.equal(got2, got) and torch.equal(xg2.grad, xg.grad)
    with pytest.raises(ValueError):
        hip_ops.deconv4s2_fewout(xg.detach(), torch.zeros(K, 5, 4, 4, device=DEV))


@pytest.mark.parametrize("shape,factor", [((1, 2, 96, 320), 4), ((2, 2, 7, 5), 4), ((1, 3, 1, 9), 4), ((1, 1, 6, 1), 2),
                                          ((1, 2, 5, 300), 3), ((1, 2, 112, 256), 4)])
def test_upsample_bilinear_vs_oracle(oracle_ops, shape, factor):
    """`20 * self.upsample(flow2)` (PWCNet.py:73,321) and its backward as a gather: forward to 4 ulp of the output
    range, gradient to 1e-6 relative L2 against autograd of F.interpolate; bit-reproducible."""
    gen = torch.Generator().manual_seed(shape[2] * shape[3] + factor)
    x = torch.randn(*shape, generator=gen).requires_grad_(True)
    want = oracle_ops.upsample_bilinear(x, factor, 20.0)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    xg = x.detach().to(DEV).requires_grad_(True)
    got = hip_ops.upsample_bilinear(xg, factor, 20.0)
    assert got.shape == want.shape
    assert max_abs(got, want) <= 5e-7 * float(want.detach().abs().max())
    got.backward(go.to(DEV))
    assert rel_l2(xg.grad, x.grad) < 1e-6
    xg2 = x.detach().to(DEV).requires_grad_(True)
    hip_ops.upsample_bilinear(xg2, factor, 20.0).backward(go.to(DEV))
    assert torch.equal(xg2.grad, xg.grad)


@pytest.mark.parametrize("shape,n,k", [((1, 2, 55, 128), 128, 7), ((2, 2, 9, 13), 5, 7), ((1, 1, 7, 20), 16, 7),
                                       ((1, 2, 11, 6), 9, 5), ((1, 2, 4, 3), 3, 3), ((1, 2, 3, 130), 64, 7)])
def test_conv_fewin_vs_oracle(oracle_ops, shape, n, k):
    """convf1 of the motion encoder (2 -> 128, 7x7 on 55x128) and ragged / tiny shapes (image narrower than the
    kernel, rows wrapping inside a 64-pixel tile), with and without bias / ReLU.  2e-6 * sqrt(taps) * max|out|."""
    gen = torch.Generator().manual_seed(n + k)
    B, C, H, W = shape
    x = 3 * torch.randn(*shape, generator=gen)
    w = torch.randn(n, C, k, k, generator=gen) / k
    for bias, relu in ((torch.randn(n, generator=gen), True), (None, False)):
        want = oracle_ops.conv_fewin(x, w, bias, relu)
        got = hip_ops.conv_fewin(x.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV), relu)
        assert got.shape == want.shape
        assert max_abs(got, want) <= 2e-6 * (C * k * k) ** 0.5 * float(want.abs().max()) + 1e-6
    with pytest.raises(RuntimeError):
        hip_ops.conv_fewin(x.to(DEV).requires_grad_(True), w.to(DEV))
    with pytest.raises(ValueError):
        hip_ops.conv_fewin(torch.zeros(1, 3, 8, 8, device=DEV), torch.zeros(4, 3, 7, 7, device=DEV))


# --------------------------------------------------------------------------- encoder normalisation
@pytest.mark.parametrize("shape,relu", [((2, 64, 220, 512), True), ((2, 96, 110, 256), False), ((2, 128, 55, 128), True),
                                        ((1, 5, 7, 9), True), ((3, 2, 33, 21), False), ((1, 3, 1, 6), True)])
def test_instance_norm_relu_vs_oracle(oracle_ops, shape, relu):
    """relu?(InstanceNorm2d(x)) vs torch's own instance_norm on the CPU (what the reference runs): the three
    feature-encoder stages at 440x1024 with both images batched, odd planes (scalar path), tiny planes.
    Statistics are accumulated in fp64 here, in fp32 (Welford) there: 3e-6 absolute on O(1) outputs; gradients 2e-5."""
    gen = torch.Generator().manual_seed(shape[1] * shape[2])
    x = (torch.randn(*shape, generator=gen) * 1.7 + 0.4).requires_grad_(True)
    want = oracle_ops.instance_norm_relu(x, 1e-5, relu)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    xg = x.detach().to(DEV).requires_grad_(True)
    got = hip_ops.instance_norm_relu(xg, 1e-5, relu)
    assert max_abs(got, want) <= 3e-6 * max(1.0, float(want.detach().abs().max()))
    got.backward(go.to(DEV))
    assert rel_l2(xg.grad, x.grad) < 2e-5
    xg2 = x.detach().to(DEV).requires_grad_(True)
    got2 = hip_ops.instance_norm_relu(xg2, 1e-5, relu)
    got2.backward(go.to(DEV))
    assert torch.equal(got2, got) and torch.equal(xg2.grad, xg.grad)  # fixed summation order


def test_add_relu_vs_oracle(oracle_ops):
    gen = torch.Generator().manual_seed(77)
    for shape in ((2, 64, 55, 128), (1, 3, 5, 7)):
        a = torch.randn(*shape, generator=gen).requires_grad_(True)
        b = torch.randn(*shape, generator=gen).requires_grad_(True)
        want = oracle_ops.add_relu(a, b)
        go = torch.randn(want.shape, generator=gen)
        want.backward(go)
        ag, bg = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
        got = hip_ops.add_relu(ag, bg)
        assert torch.equal(got.cpu(), want.detach())
        got.backward(go.to(DEV))
        assert torch.equal(ag.grad.cpu(), a.grad) and torch.equal(bg.grad.cpu(), b.grad)


# --------------------------------------------------------------------------- attack math
def test_attack_math_vs_reference_golden():
    g = load_golden("attack_math")
    pred, target = t(g["pred"], DEV), t(g["target"], DEV)
    img1, img2 = t(g["image1"], DEV), t(g["image2"], DEV)
    for box, (ka, kb) in (("change_of_variables", ("w1", "w2")), ("clipping", ("c1", "c2"))):
        a, b = t(g[ka], DEV).requires_grad_(True), t(g[kb], DEV).requires_grad_(True)
        d1, d2 = hip_ops.extract_deltas(a, b, img1, img2, box, eps_box=1e-7)
        assert max_abs(d1, t(g["delta1_" + box])) <= 2e-7 and max_abs(d2, t(g["delta2_" + box])) <= 2e-7
        gd = t(g["gdelta"], DEV)
        ((d1 * gd).sum() + (d2 * gd.flip(-1)).sum()).backward()
        assert rel_l2(a.grad, t(g["gw1_" + box])) < 2e-6 and rel_l2(b.grad, t(g["gw2_" + box])) < 2e-6
        for f_type in ("aee", "mse", "cosim"):
            for bound in (0.005, 10.0):
                key = "%s_%s_%g" % (box, f_type, bound)
                p = pred.clone().requires_grad_(True)
                dd1 = t(g["delta1_" + box], DEV).requires_grad_(True)
                dd2 = t(g["delta2_" + box], DEV).requires_grad_(True)
                loss = hip_ops.loss_delta_constraint(p, target, dd1, dd2, None, delta_bound=bound, mu=5e5,
                                                     f_type=f_type)
                loss.backward()
                want = float(g["loss_" + key])
                assert abs(float(loss) - want) <= 2e-6 * abs(want), key
                assert rel_l2(p.grad, t(g["gpred_" + key])) < 2e-6, key
                if float(np.abs(g["gd1_" + key]).max()) == 0:
                    assert float(dd1.grad.abs().max()) == 0 and float(dd2.grad.abs().max()) == 0
                else:
                    assert rel_l2(dd1.grad, t(g["gd1_" + key])) < 2e-6, key
                    assert rel_l2(dd2.grad, t(g["gd2_" + key])) < 2e-6, key
    nd = t(g["nw_delta"], DEV).requires_grad_(True)
    dj, dj2 = hip_ops.extract_deltas_joint(nd, torch.max(img1, img2), torch.min(img1, img2))
    assert dj is dj2 and max_abs(dj, t(g["delta_joint"])) <= 1e-7
    pj = pred.clone().requires_grad_(True)
    lj = hip_ops.loss_delta_constraint(pj, target, dj, dj2, None, delta_bound=0.005, mu=5e5, f_type="aee")
    lj.backward()
    assert abs(float(lj) - float(g["loss_joint"])) <= 2e-6 * abs(float(g["loss_joint"]))
    assert rel_l2(nd.grad, t(g["gnd_joint"])) < 2e-6
    assert rel_l2(pj.grad, t(g["gpred_joint"])) < 2e-6
    assert abs(float(hip_ops.avg_epe(pred, target)) - float(g["aee"])) < 2e-6 * float(g["aee"])
    assert abs(float(hip_ops.avg_epe(pred[0], target[0])) - float(g["aee3"])) < 2e-6 * float(g["aee3"])
    assert abs(float(hip_ops.two_norm_avg(d1)) - float(g["l2_1"])) < 2e-6 * float(g["l2_1"])
    assert abs(float(hip_ops.two_norm_avg_delta(d1, d2)) - float(g["l2_12"])) < 2e-6 * float(g["l2_12"])


@pytest.mark.parametrize("shape", [
    # (B, Ca, Cb, Cout, H, W): RAFT z|r and q at the BASELINE feature size, GMA q, ragged everything, no second input
    (1, 128, 128, 256, 55, 128), (1, 128, 128, 128, 55, 128), (1, 128, 256, 128, 55, 128),
    (2, 5, 6, 7, 9, 70), (1, 12, 0, 66, 6, 3), (1, 3, 2, 4, 1, 1)])
@pytest.mark.parametrize("vertical", [False, True])
@pytest.mark.parametrize("algo", ["winograd", "direct"])
def test_sepconv5_vs_oracle(oracle_ops, sepconv5_algo, shape, vertical, algo):
    """SepConvGRU (1,5)/(5,1) gate convolutions (update.py:36-60) over [a | b].  direct: implicit MFMA GEMM, exact fp32
    products, different summation order than conv2d -> 2e-6 relative L2 forward and data gradients.  winograd: 1-D
    F(2,5) where the shape is eligible (the 55x128 shapes; the direct kernel elsewhere): transform constants up to 5,
    1.0e-6 rms against fp64 -> 5e-6 relative L2."""
    B, Ca, Cb, Cout, H, W = shape
    sepconv5_algo(algo)
    tol = 5e-6 if algo == "winograd" else 2e-6
    gen = torch.Generator().manual_seed(31 + Ca + W)
    a = torch.randn(B, Ca, H, W, generator=gen)
    b = torch.randn(B, Cb, H, W, generator=gen) if Cb else None
    w = torch.randn((Cout, Ca + Cb) + ((5, 1) if vertical else (1, 5)), generator=gen) / (5 * (Ca + Cb)) ** .5
    go = torch.randn(B, Cout, H, W, generator=gen)
    ca = a.clone().requires_grad_(True)
    cb = None if b is None else b.clone().requires_grad_(True)
    want = oracle_ops.sepconv5(ca, cb, w)
    want.backward(go)
    ga = a.clone().to(DEV).requires_grad_(True)
    gb = None if b is None else b.clone().to(DEV).requires_grad_(True)
    got = hip_ops.sepconv5(ga, gb, w.to(DEV))
    assert got.shape == want.shape
    assert rel_l2(got, want) < tol, rel_l2(got, want)
    got.backward(go.to(DEV))
    assert rel_l2(ga.grad, ca.grad) < tol
    if b is not None:
        assert rel_l2(gb.grad, cb.grad) < tol


@pytest.mark.parametrize("shape", [(1, 128, 128, 55, 128), (2, 8, 12, 9, 70), (1, 128, 256, 16, 24), (2, 128, 128, 17, 128),
                                   (1, 64, 64, 6, 256)])
@pytest.mark.parametrize("algo", ["winograd", "direct"])
def test_gru_step_vs_oracle(oracle_ops, sepconv5_algo, shape, algo):
    """The fused SepConvGRU update (one autograd node, gradients of h / motion features accumulated inside the
    kernels) against the oracle's composition of the same operators under autograd: values and every gradient
    (h, rest, the four hoisted context parts) to 5e-6 relative L2 -- only the order of a few additions differs --
    with the direct kernels, 1.5e-5 with the F(2,5) Winograd kernels (all four fused epilogues, both orientations,
    odd height = a row pair with one row, batch 2, the 32-channel x 4-group tile at 64 channels)."""
    B, C, Cr, H, W = shape
    sepconv5_algo(algo)
    gen = torch.Generator().manual_seed(7 + C + W)
    rnd = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    h, rest = torch.tanh(rnd(B, C, H, W)), rnd(B, Cr, H, W)
    halves = []
    for k in ((1, 5), (5, 1)):
        sc = (5 * (C + Cr)) ** -.5
        halves.append((rnd(2 * C, C + Cr, *k) * sc, rnd(B, 2 * C, H, W), rnd(C, C + Cr, *k) * sc, rnd(B, C, H, W)))
    go = rnd(B, C, H, W)

    def run(ops_mod, dev):
        leaf = lambda t, g: t.detach().clone().to(dev).requires_grad_(g)  # noqa: E731
        hh, rr = leaf(h, True), leaf(rest, True)
        hv = [tuple(leaf(t, i % 2 == 1) for i, t in enumerate(hf)) for hf in halves]
        out = ops_mod.gru_step(hh, rr, tuple(hv))
        out.backward(go.to(dev))
        return [out, hh.grad, rr.grad] + [hf[i].grad for hf in hv for i in (1, 3)]

    want, got = run(oracle_ops, "cpu"), run(hip_ops, DEV)
    for w_, g_ in zip(want, got):
        assert rel_l2(g_, w_) < (1.5e-5 if algo == "winograd" else 5e-6), rel_l2(g_, w_)
    again = run(hip_ops, DEV)
    assert all(torch.equal(x, y) for x, y in zip(got, again))   # fixed summation order


@pytest.mark.parametrize("shape", [
    # (B, Cin, Cout, H, W): update-block convolutions at the BASELINE feature size, then ragged everything
    (1, 256, 192, 55, 128), (1, 256, 126, 55, 128), (1, 128, 256, 55, 128), (1, 128, 64, 55, 128),
    (2, 5, 7, 9, 21), (1, 12, 70, 3, 5), (1, 3, 2, 1, 1), (2, 64, 64, 40, 48),
    # F(4x4,3x3) without the channel split (enough tiles), ragged tile rows / partial column blocks / odd channel counts
    (2, 64, 64, 220, 512), (1, 21, 37, 30, 68), (1, 96, 96, 110, 256), (1, 126, 256, 55, 128), (1, 565, 128, 24, 80),
    (1, 192, 256, 55, 128), (2, 96, 96, 110, 256),
    # PWC-Net's dense decoder blocks (PWCNet.py:110-158): many input channels on one image, maps from 96x320 to 6x20
    (1, 245, 128, 96, 320), (1, 501, 64, 48, 160), (1, 196, 196, 6, 20)])
@pytest.mark.parametrize("relu", [False, True, 0.1])
def test_conv3x3_winograd_vs_oracle(oracle_ops, shape, relu):
    """Winograd on fp32 MFMA against conv2d: forward (+bias, +ReLU / LeakyReLU) and data gradient, once with the
    transform the library picks for the shape (pcfa_conv3x3_algo) and -- in a child process, the switch is read once
    -- with F(4x4,3x3) forced wherever it is supported (tests/test_gpu_parity.py::test_conv3x3_f43_forced).
    Tolerance, relative L2 against torch's fp32 conv2d: F(2x2,3x3) 5e-6 (a few roundings per product), F(4x4,3x3)
    1.5e-5 (transform constants up to 8: ~2e-6 against fp64)."""
    import os
    B, Cin, Cout, H, W = shape
    lib = hip_ops._hip.load()
    f43 = lib.pcfa_conv3x3_algo(B, Cin, Cout, H, W) == 43
    if os.environ.get("PCFA_CONV3X3_ALGO") == "f43":
        assert f43 == (W % 4 == 0 and W >= 8)
    elif "PCFA_CONV3X3_ALGO" not in os.environ:
        # ((1, 565, 128, 24, 80) and (1, 196, 196, 6, 20) went from the channel-split F(4x4,3x3) path to F(2x2,3x3) with K
        # sliced over workgroups in r04; PCFA_CONV3X3_ALGO=f43 still runs them through F(4x4,3x3))
        # r05: F(4x4,3x3) serves PWC-Net's decoder shapes only -- RAFT / GMA shapes ((1, 192, 256, 55, 128),
        # (2, 64, 64, 220, 512)) left it: no speed at the step level, 1.1-2.3x farther from fp64 (csrc/conv3x3.hip use_f43)
        assert f43 == (shape in ((1, 245, 128, 96, 320), (1, 501, 64, 48, 160)))
    tol = 1.5e-5 if f43 else 5e-6
    gen = torch.Generator().manual_seed(3 + Cin + W)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (9 * Cin) ** .5
    b = torch.randn(Cout, generator=gen)
    go = torch.randn(B, Cout, H, W, generator=gen)
    cx = x.clone().requires_grad_(True)
    act = dict(relu=False, leaky_slope=relu) if isinstance(relu, float) else dict(relu=relu)   # 0.1: PWC-Net's LeakyReLU
    want = oracle_ops.conv3x3(cx, w, b, **act)
    want.backward(go)
    gx = x.clone().to(DEV).requires_grad_(True)
    got = hip_ops.conv3x3(gx, w.to(DEV), b.to(DEV), **act)
    assert got.shape == want.shape
    assert rel_l2(got, want) < tol, rel_l2(got, want)
    got.backward(go.to(DEV))
    # with an activation the gradient also sees the outputs within rounding of zero whose mask bit flips: every flipped
    # element moves the gradient by one output's share, sqrt(flips / outputs) in relative L2 (measured: 14 flips of
    # 14.4 M outputs at 220x512 -> 1.0e-3); the convolution's own backward error is the relu=False case of this test
    flips = int(((got.detach().cpu() > 0) != (want.detach() > 0)).sum())
    assert flips <= max(4, 4e-6 * want.numel()), flips
    # a flipped mask bit moves the gradient by |grad_out| at that output (N(0,1): up to ~4.5 in a few million draws)
    gtol = tol if relu is False else tol + 5.0 * (flips / want.numel()) ** .5
    assert rel_l2(gx.grad, cx.grad) < gtol, (rel_l2(gx.grad, cx.grad), flips)
    got2 = hip_ops.conv3x3(gx.detach(), w.to(DEV), b.to(DEV), **act)
    assert torch.equal(got2, got.detach())          # no atomics anywhere, split or not: bitwise reproducible


def test_conv3x3_f43_forced():
    """Every conv3x3 test once more with Winograd F(4x4,3x3) forced on all shapes it supports (the policy only picks it
    where it is faster): channel-split and direct paths, ragged tiles, odd channel counts, masks, residual gradients."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PCFA_CONV3X3_ALGO="f43")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "conv3x3 and not forced and not fewout"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout


def test_conv3x3_k_slices_forced():
    """Every conv3x3 test once more with the F(2x2,3x3) kernel's input channels sliced over three workgroups wherever the
    shape allows (H W % 4 == 0, >= 2 chunks; the policy only slices small single-image maps): partial outputs + the
    finish pass with every epilogue (bias, activations, masks, channel-prefix masks, addends), ragged K."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PCFA_CONV3X3_ALGO="f23", PCFA_CONV3X3_KSL="3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "conv3x3 and not forced and not fewout"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("shape", [
    # (B, Cin, N, k, H, W): the stem and the two stride-2 residual-block entries at the BASELINE size, ragged / tiny ones
    (2, 3, 64, 7, 440, 1024), (1, 64, 96, 3, 220, 512), (2, 96, 128, 3, 110, 256),
    (1, 3, 16, 3, 20, 36), (2, 3, 64, 7, 37, 52), (1, 10, 40, 3, 9, 264), (1, 16, 32, 3, 2, 4), (1, 3, 20, 7, 21, 72)])
def test_conv_s2_vs_oracle(oracle_ops, shape):
    """extractor.py:118 and :23-58 (stride 2): forward and data gradient against the CPU restatement."""
    B, Cin, N, k, H, W = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(N, Cin, k, k, generator=gen) / (Cin * k * k) ** .5
    b = torch.randn(N, generator=gen)
    for kw in (dict(relu=False), dict(relu=True), dict(leaky_slope=0.1)):
        bias = None if kw == dict(relu=False) else b
        xc = x.clone().requires_grad_(True)
        want = oracle_ops.conv_s2(xc, w, bias, **kw)
        go = torch.randn(want.shape, generator=gen)
        want.backward(go)
        xg = x.to(DEV).requires_grad_(True)
        wg = w.to(DEV)
        assert hip_ops.conv_s2_supported(xg, wg)
        got = hip_ops.conv_s2(xg, wg, None if bias is None else bias.to(DEV), **kw)
        got.backward(go.to(DEV))
        assert got.shape == want.shape
        scale = want.abs().max().item()
        assert (got.cpu() - want).abs().max().item() <= 2e-5 * max(scale, 1.0), kw
        flips = ((got.cpu() > 0) != (want > 0)).sum().item() if kw != dict(relu=False) else 0
        gerr = (xg.grad.cpu() - xc.grad).norm().item() / xc.grad.norm().item()
        assert gerr <= 2e-5 + 2 * (flips / want.numel()) ** .5, (kw, gerr, flips)
    assert not hip_ops.conv_s2_supported(torch.zeros(1, 3, 8, 10, device=DEV), torch.zeros(4, 3, 3, 3, device=DEV))  # W % 4
    assert not hip_ops.conv_s2_supported(torch.zeros(1, 4, 8, 12, device=DEV), torch.zeros(4, 4, 5, 5, device=DEV))
    with pytest.raises(RuntimeError):
        hip_ops.conv_s2(torch.zeros(1, 3, 8, 8), w[:, :3, :3, :3].contiguous())       # no CPU fallback


@pytest.mark.parametrize("shape", [(2, 64, 96, 220, 512), (1, 96, 128, 110, 256), (1, 10, 40, 9, 264), (2, 16, 16, 6, 8)])
def test_conv_s2_ds_vs_oracle(oracle_ops, shape):
    """extractor.py:23-58 (stride 2): conv1 (3x3) and downsample[0] (1x1) of the block input in one launch per direction,
    both outputs and the summed data gradient against the CPU restatement."""
    B, Cin, N, H, W = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(N, Cin, 3, 3, generator=gen) / (Cin * 9) ** .5
    wd = torch.randn(N, Cin, 1, 1, generator=gen) / Cin ** .5
    b, bd = torch.randn(N, generator=gen), torch.randn(N, generator=gen)
    for relu, bias in ((False, False), (True, True)):
        xc = x.clone().requires_grad_(True)
        want, want_d = oracle_ops.conv_s2_ds(xc, w, wd, b if bias else None, bd if bias else None, relu=relu)
        go, god = torch.randn(want.shape, generator=gen), torch.randn(want.shape, generator=gen)
        (want * go + want_d * god).sum().backward()
        xg = x.to(DEV).requires_grad_(True)
        assert hip_ops.conv_s2_ds_supported(xg, w.to(DEV), wd.to(DEV))
        got, got_d = hip_ops.conv_s2_ds(xg, w.to(DEV), wd.to(DEV), b.to(DEV) if bias else None,
                                        bd.to(DEV) if bias else None, relu=relu)
        (got * go.to(DEV) + got_d * god.to(DEV)).sum().backward()
        for g_, w_ in ((got, want), (got_d, want_d)):
            assert (g_.cpu() - w_).abs().max().item() <= 2e-5 * max(w_.abs().max().item(), 1.0)
        flips = ((got.cpu() > 0) != (want > 0)).sum().item() if relu else 0
        gerr = (xg.grad.cpu() - xc.grad).norm().item() / xc.grad.norm().item()
        assert gerr <= 2e-5 + 2 * (flips / want.numel()) ** .5, (relu, gerr, flips)
    assert not hip_ops.conv_s2_ds_supported(torch.zeros(1, 8, 8, 12, device=DEV), torch.zeros(8, 8, 3, 3, device=DEV),
                                            torch.zeros(8, 8, 1, 1, device=DEV))          # W % 8


def test_pm1_pair_is_bit_identical_to_the_torch_expression():
    """raft.py:88-89 `2 * (image / 255.0) - 1.0` for both images + the gradient sum of image 1's two uses, one launch each."""
    gen = torch.Generator().manual_seed(5)
    for shape in ((1, 3, 440, 1024), (2, 3, 37, 53)):
        a = (255 * torch.rand(shape, generator=gen)).to(DEV).requires_grad_(True)
        b = (255 * torch.rand(shape, generator=gen)).to(DEV).requires_grad_(True)
        gp = torch.randn((2 * shape[0],) + shape[1:], generator=gen).to(DEV)
        gc = torch.randn(shape, generator=gen).to(DEV)
        n1, n2 = 2 * (a / 255.0) - 1.0, 2 * (b / 255.0) - 1.0
        torch.autograd.backward([torch.cat([n1, n2], 0), n1], [gp, gc])
        want = (torch.cat([n1, n2], 0).detach(), n1.detach(), a.grad.clone(), b.grad.clone())
        a.grad = b.grad = None
        pair, cx = hip_ops.pm1_pair(a, b)
        torch.autograd.backward([pair, cx], [gp, gc])
        for got, w in zip((pair.detach(), cx.detach(), a.grad, b.grad), want):
            assert torch.equal(got, w)
        a.grad = b.grad = None
        pair, cx = hip_ops.pm1_pair(a, b)          # the context encoder's copy unused: its gradient is absent
        pair.backward(gp)
        n1, n2 = 2 * (a.detach().requires_grad_(True) / 255.0) - 1.0, None
        assert torch.equal(a.grad, (gp[:shape[0]] * 2) / 255.0)


def test_sepconv5_rejects_bad_operands():
    w = torch.zeros(4, 3, 1, 5, device=DEV)
    with pytest.raises(ValueError):
        hip_ops.sepconv5(torch.zeros(1, 2, 4, 4, device=DEV), None, w)          # channel mismatch
    with pytest.raises(ValueError):
        hip_ops.sepconv5(torch.zeros(1, 3, 4, 4, device=DEV), None, torch.zeros(4, 3, 3, 3, device=DEV))
    with pytest.raises(RuntimeError):
        hip_ops.sepconv5(torch.zeros(1, 3, 4, 4), None, w.cpu())               # no CPU fallback
    wg = w.clone().requires_grad_(True)
    out = hip_ops.sepconv5(torch.zeros(1, 3, 4, 4, device=DEV, requires_grad=True), None, wg)
    with pytest.raises(RuntimeError):
        out.sum().backward()                                                    # frozen-weight path only


def test_gru_gate_kernels_vs_oracle(oracle_ops):
    """SepConvGRU elementwise chain (update.py:45-60): device expf/tanhf vs torch CPU -> 2e-6 relative."""
    gen = torch.Generator().manual_seed(8)
    for shape in ((1, 128, 55, 128), (2, 7, 5, 3)):   # BASELINE size and a ragged one (n % 4 != 0)
        zc, rc, qc = (2 * torch.randn(shape, generator=gen) for _ in range(3))
        h = torch.tanh(torch.randn(shape, generator=gen))
        lv = [t_.clone().requires_grad_(True) for t_ in (zc, rc, qc, h)]
        C = shape[1]
        bz, br, bq = (torch.randn(C, generator=gen) for _ in range(3))
        z, rh = oracle_ops.gru_gates(lv[0], lv[1], lv[3], bz, br)
        want = oracle_ops.gru_update(z, lv[2] + rh, lv[3], bq)      # rh enters q's pre-activation like convq would
        go = torch.randn(shape, generator=gen)
        want.backward(go)
        gv = [t_.clone().to(DEV).requires_grad_(True) for t_ in (zc, rc, qc, h)]
        zg, rhg = hip_ops.gru_gates(gv[0], gv[1], gv[3], bz.to(DEV), br.to(DEV))
        got = hip_ops.gru_update(zg, gv[2] + rhg, gv[3], bq.to(DEV))
        assert max_abs(got, want) <= 2e-6
        got.backward(go.to(DEV))
        for a, b in zip(gv, lv):
            assert rel_l2(a.grad, b.grad) < 2e-6
        # stacked z|r pre-activations + the hoisted constant-input addends (SepConvGRU.step)
        zr = torch.randn(shape[0], 2 * C, *shape[2:], generator=gen)
        add_zr, add_q = torch.randn(zr.shape, generator=gen), torch.randn(shape, generator=gen)
        cl = [t_.clone().requires_grad_(True) for t_ in (zr, add_zr, qc, add_q, h)]
        zz, rhh = oracle_ops.gru_gates_packed(cl[0], cl[4], None, cl[1])
        want2 = oracle_ops.gru_update(zz, cl[2] + rhh, cl[4], None, cl[3])
        want2.backward(go)
        gl = [t_.clone().to(DEV).requires_grad_(True) for t_ in (zr, add_zr, qc, add_q, h)]
        zg2, rhg2 = hip_ops.gru_gates_packed(gl[0], gl[4], None, gl[1])
        got2 = hip_ops.gru_update(zg2, gl[2] + rhg2, gl[4], None, gl[3])
        assert max_abs(got2, want2) <= 2e-6
        got2.backward(go.to(DEV))
        for a, b in zip(gl, cl):
            assert rel_l2(a.grad, b.grad) < 2e-6
        # conv -> +bias -> ReLU tail
        x = torch.randn(shape, generator=gen).requires_grad_(True)
        wantr = oracle_ops.bias_relu(x, bz)
        wantr.backward(go)
        xg = x.detach().to(DEV).requires_grad_(True)
        gotr = hip_ops.bias_relu(xg, bz.to(DEV))
        assert max_abs(gotr, wantr) == 0.0
        gotr.backward(go.to(DEV))
        assert max_abs(xg.grad, x.grad) == 0.0


def test_box_transform_vs_oracle(oracle_ops):
    gen = torch.Generator().manual_seed(2)
    img = torch.rand(3, 3, 37, 53, generator=gen)
    for cov, scale, with_delta in ((True, 255., False), (False, 255., True), (False, 1., True), (True, 1., False)):
        x = (torch.atanh(2 * (1 - 1e-7) * img - (1 - 1e-7)) if cov else img + 0.3 * torch.randn(img.shape, generator=gen))
        x = x.requires_grad_(True)
        delta = (0.2 * torch.randn(3, 37, 53, generator=gen)).requires_grad_(True) if with_delta else None
        want = oracle_ops.box_transform(x, delta, cov, 1e-7, scale)
        go = torch.randn(want.shape, generator=gen)
        want.backward(go)
        xg = x.detach().to(DEV).requires_grad_(True)
        dg = delta.detach().to(DEV).requires_grad_(True) if with_delta else None
        got = hip_ops.box_transform(xg, dg, cov, 1e-7, scale)
        assert max_abs(got, want) <= 2e-6 * scale
        got.backward(go.to(DEV))
        assert rel_l2(xg.grad, x.grad) < 2e-6
        if with_delta:
            assert rel_l2(dg.grad, delta.grad) < 2e-6


def test_loss_on_unpadded_view_full_size():
    """BASELINE size: the loss reads the 436x1024 crop of a 440x1024 flow in place (strided)."""
    gen = torch.Generator().manual_seed(4)
    full = (5 * torch.randn(1, 2, 440, 1024, generator=gen)).to(DEV).requires_grad_(True)
    crop = full[..., 2:438, :]
    target = torch.zeros(1, 2, 436, 1024, device=DEV)
    d = (0.01 * torch.randn(1, 3, 440, 1024, generator=gen)).to(DEV).requires_grad_(True)
    loss = hip_ops.loss_delta_constraint(crop, target, d, d.detach().clone(), None, delta_bound=0.005, mu=5e5)
    want_sim = crop.detach().double().pow(2).sum(1).sqrt().mean()
    msq = d.detach().double().pow(2).sum() * 2 / (2 * d.numel())
    want = float(want_sim + 5e5 * max(0.0, float(msq) - 0.005 ** 2))
    assert abs(float(loss) - want) <= 2e-6 * want
    loss.backward()
    assert float(full.grad[..., :2, :].abs().max()) == 0 and float(full.grad[..., 438:, :].abs().max()) == 0
    # checksum-of-checksums: reproducible bits
    loss2 = hip_ops.loss_delta_constraint(crop, target, d, d.detach().clone(), None, delta_bound=0.005, mu=5e5)
    assert float(loss2) == float(loss)


# --------------------------------------------------------------------------- whole closure + attack
CASES = {
    "raft": ("RAFT", 128, 160, "change_of_variables", False, "zero", "aee", 1),
    "gma": ("GMA", 128, 160, "change_of_variables", False, "neg_flow", "aee", 2),
    "pwcnet": ("PWCNet", 120, 180, "clipping", True, "zero", "aee", 3),
    "spynet": ("SpyNet", 100, 150, "change_of_variables", False, "zero", "mse", 4),
    # wiring-only fixture: the reference's model code with its CUDA-only extensions bound to the oracle
    # (tests/golden/make_golden.py docstring)
    "flownet2": ("FlowNet2", 128, 192, "change_of_variables", False, "zero", "aee", 5),
}


@pytest.mark.parametrize("name", list(CASES))
def test_closure_on_gpu_vs_reference_golden(name):
    """One full closure on the MI355X vs the REFERENCE's CPU closure (golden): flow 1e-3 of its range
    (north_star's AEE tolerance), loss 1e-4 relative, gradient 1e-2 relative L2 (MIOpen vs MKL-DNN convs)."""
    g = load_golden("closure_" + name)
    net, h, w, box, joint, tgt, loss, seed = CASES[name]
    leaves = [t(g["leaf0"])] if joint else [t(g["leaf0"]), t(g["leaf1"])]
    torch.backends.cudnn.allow_tf32 = False
    r = closure_util.run_closure(net, h, w, box, joint, tgt, loss, seed, torch.device(DEV),
                                 images=(t(g["image1"].astype(np.float32)), t(g["image2"].astype(np.float32))),
                                 leaves=leaves)
    scale = float(np.abs(g["flow"]).max())
    assert max_abs(r["flow"], t(g["flow"])) <= 1e-3 * scale
    aee = float((r["flow"].cpu() - t(g["flow"])).pow(2).sum(1).sqrt().mean())
    assert aee <= 1e-3
    assert abs(r["loss"] - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    for i, gr in enumerate(r["grads"]):
        assert rel_l2(gr, t(g["grad%d" % i])) < 1e-2


BASELINE_CASES = {
    # BASELINE.json configs 2, 3, 4 at their full sizes (the reference cannot travel: the checker is the CPU port,
    # i.e. pcfa_amd's host code on the oracle operators, itself pinned to the reference at 128x160 above)
    "raft_436x1024": ("RAFT", 436, 1024, "change_of_variables", False, "zero", "aee", 11),
    "gma_436x1024": ("GMA", 436, 1024, "change_of_variables", False, "neg_flow", "aee", 12),
    "pwcnet_375x1242": ("PWCNet", 375, 1242, "clipping", True, "zero", "aee", 13),
}


@pytest.mark.parametrize("name", list(BASELINE_CASES))
def test_closure_at_baseline_size_vs_cpu_port(oracle_ops, name):
    """One closure at the BASELINE size on the MI355X vs the CPU port on the same seeded inputs: flow AEE <= 1e-3
    (north_star's tolerance), loss 1e-4 relative, gradient 1e-2 relative L2 (MIOpen / MFMA vs MKL-DNN convolutions).
    Reference: attack_PCFA.py:175-192."""
    net, h, w, box, joint, tgt, loss, seed = BASELINE_CASES[name]
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    gpu = closure_util.run_closure(net, h, w, box, joint, tgt, loss, seed, torch.device(DEV))
    with ops.override_for_testing(oracle_ops):
        cpu = closure_util.run_closure(net, h, w, box, joint, tgt, loss, seed, torch.device("cpu"))
    closure_util._MODELS.clear()     # 5 M .. 10 M parameters per model and device: do not keep them for later tests
    for k in ("flow_init", "flow"):
        d = gpu[k].cpu() - cpu[k]
        assert float(d.pow(2).sum(1).sqrt().mean()) <= 1e-3, (k, float(d.pow(2).sum(1).sqrt().mean()))
    assert abs(gpu["loss"] - cpu["loss"]) <= 1e-4 * abs(cpu["loss"]), (gpu["loss"], cpu["loss"])
    for a, b in zip(gpu["grads"], cpu["grads"]):
        assert rel_l2(a, b) < 1e-2, rel_l2(a, b)


def test_pcfa_attack_on_gpu_vs_reference_trajectory():
    from argparse import Namespace
    from pcfa_amd import attack_PCFA
    g = load_golden("trajectory_raft")
    ref8, ref3 = g["threads8"], g["threads3"]
    args = Namespace(net="RAFT", steps=5, joint_perturbation=False, boxconstraint="change_of_variables",
                     delta_bound=0.005, target="zero", custom_target_path="", loss="aee", save_frequency=1,
                     small_save=False, no_save=True, unregistered_artifacts=True, universal_perturbation=False,
                     mu=-1, weights="random:1234")
    model = closure_util.load_model("RAFT", True, torch.device(DEV))
    res = attack_PCFA.pcfa_attack(model, t(g["image1"].astype(np.float32)), t(g["image2"].astype(np.float32)),
                                  torch.zeros(1, 2, 128, 160), 0, None, 1e-7, torch.device(DEV), False,
                                  2500. / 0.005, args)
    res = np.array([np.nan if v is None else float(v) for v in res])
    assert abs(res[1] - ref8[1]) < 1e-3
    for idx in (4, 5, 8, 9, 10, 11):
        tol = max(1e-3, 3 * abs(ref8[idx] - ref3[idx])) * max(1.0, abs(ref8[idx]))
        assert abs(res[idx] - ref8[idx]) <= tol, (idx, res[idx], ref8[idx], ref3[idx])


def _cli_args(**kw):
    from argparse import Namespace
    base = dict(net="SpyNet", weights="random:1234", dataset="Synthetic", dataset_stage="evaluation", small_run=False,
                synthetic_size="64x96", synthetic_pairs=2, dstype="final", output_folder="experiment_data",
                small_save=False, save_frequency=1, no_save=True, unregistered_artifacts=True,
                joint_perturbation=False, steps=2, universal_perturbation=False, boxconstraint="change_of_variables",
                batch_size=2, delta_bound=0.005, mu=-1, epochs=1, target="zero", custom_target_path="", loss="aee")
    base.update(kw)
    return Namespace(**base)


@pytest.mark.parametrize("net,joint,box", [("RAFT", False, "change_of_variables"), ("PWCNet", True, "clipping")])
def test_attack_l2_pairs_in_flight_equals_sequential(net, joint, box, tmp_path):
    """`--pairs_in_flight 2` (the per-pair driver could not use the idle quarter of the chip): attack_l2
    attacks the dataset's pairs two at a time on one GPU.  Averages AND the saved per-pair artefacts must equal the sequential
    run bit for bit (three pairs: one full group and a remainder of one)."""
    import glob
    import os
    from pcfa_amd import attack_PCFA
    size = "128x160" if net == "RAFT" else "64x96"
    outs = []
    for nflight in (1, 2):
        folder = str(tmp_path / ("flight%d" % nflight))
        a = _cli_args(net=net, joint_perturbation=joint, boxconstraint=box, synthetic_size=size, synthetic_pairs=3, steps=2,
                      no_save=False, output_folder=folder, pairs_in_flight=nflight)
        res = attack_PCFA.attack_l2(a)
        files = sorted(glob.glob(os.path.join(folder, "**", "*.npy"), recursive=True))
        outs.append((res, {os.path.basename(f): np.load(f) for f in files}))
    (r1, f1), (r2, f2) = outs
    assert r1["pairs"] == r2["pairs"] == 3
    for k in r1:
        assert r1[k] == r2[k] or (np.isnan(r1[k]) and np.isnan(r2[k])), (k, r1[k], r2[k])
    assert f1 and sorted(f1) == sorted(f2)
    for name in f1:
        assert np.array_equal(f1[name], f2[name]), name


@pytest.mark.parametrize("net,joint,box", [("SpyNet", False, "change_of_variables"), ("PWCNet", True, "clipping"),
                                           ("RAFT", False, "clipping"),
                                           pytest.param("FlowNet2", False, "change_of_variables", marks=LONG_ONLY)])
def test_attack_l2_end_to_end_on_gpu_vs_cpu_port(oracle_ops, net, joint, box):
    """The whole driver (dataset -> model -> pcfa_attack per pair -> averages) on the GPU vs the same host code with
    the oracle operators on CPU: 2 pairs x 1 step (10 closures; longer runs amplify fp32 noise chaotically --
    SURVEY.md D10 -- and are covered by the trajectory test against the reference's own noise floor)."""
    from pcfa_amd import attack_PCFA
    size = {"RAFT": "128x160", "FlowNet2": "64x128"}.get(net, "64x96")
    a = _cli_args(net=net, joint_perturbation=joint, boxconstraint=box, synthetic_size=size, steps=1)
    got = attack_PCFA.attack_l2(a)
    import os
    os.environ["PCFA_USE_CPU"] = "1"
    try:
        import importlib
        from pcfa_amd.helper_functions import config_paths
        importlib.reload(config_paths)
        importlib.reload(attack_PCFA)
        with ops.override_for_testing(oracle_ops):
            want = attack_PCFA.attack_l2(a)
    finally:
        os.environ.pop("PCFA_USE_CPU")
        importlib.reload(config_paths)
        importlib.reload(attack_PCFA)
    assert got["pairs"] == want["pairs"] == 2
    keys = ("aee_avg_pred-tgt", "aee_avg_predadv-tgt", "aee_avg_pred-predadv", "l2_avg_delta12",
            "aee_avg_predadv-tgt_min", "l2_avg_delta12_min")
    if net in ("SpyNet", "FlowNet2"):
        # with these seeded weights one SpyNet pair sits on a knife edge: the CPU port ALONE lands on
        # l2 = 0.0025 (3 threads) or 0.28 (8 threads) after 10 closures, so only the deterministic part is compared
        # (FlowNet2: 162 M random weights and three warps with a piecewise gradient amplify fp32 noise the same way;
        # the closure itself is pinned by test_closure_on_gpu_vs_reference_golden[flownet2])
        keys = ("aee_avg_pred-tgt",)
        assert all(np.isfinite(got[k]) for k in got if isinstance(got[k], float) and "gt" not in k)
    for k in keys:
        assert abs(got[k] - want[k]) <= 5e-3 * max(1.0, abs(want[k])), (k, got[k], want[k])


def test_universal_attack_runs_on_gpu():
    from pcfa_amd import attack_PCFA
    res = attack_PCFA.attack_l2_universal(_cli_args(universal_perturbation=True, boxconstraint="clipping", steps=1))
    d = res["delta1"]
    assert d.is_cuda and d.shape == (3, 64, 128) and float(d.abs().max()) > 0
    assert len(res["history"]) == 1 and np.isfinite(res["history"][0]["aee_predadv-tgt"])


@pytest.mark.parametrize("graph", ["graph", "eager"])
def test_universal_attack_on_gpu_vs_reference_golden(graph, monkeypatch):
    """BASELINE config 5 (RAFT, --universal_perturbation) on the MI355X against the REFERENCE's own
    attack_l2_universal run (attack_PCFA.py:297-566; tests/golden/make_golden.py::golden_universal): per-step metrics
    and the final perturbation pair, tolerance = 3x the reference's own 8-vs-3-thread spread.  Run with
    the closure replayed from its hipGraph (captured once, reused for both batches) and launched eagerly."""
    from pcfa_amd import attack_PCFA
    monkeypatch.setenv("PCFA_HIP_GRAPH", "1" if graph == "graph" else "0")
    g = load_golden("universal_raft")
    args, loader = closure_util.universal_case(g)
    res = attack_PCFA.attack_l2_universal(args, data_loader=loader, has_gt=False)
    assert res["graphed"] == (graph == "graph")
    closure_util.check_universal_against_golden(res, g, rel_l2)


def test_universal_closure_on_gpu_vs_oracle(oracle_ops):
    """ONE universal closure (batch of 2 pairs, a shared non-zero delta pair, RAFT 128x160) on the GPU against the
    oracle on CPU: loss 1e-5, d loss / d delta 1e-4 relative L2... measured against the CPU port's own conv noise,
    see the assert messages.  Also checks the graph replay against the eager launch and the packed buffer of the
    single all-reduce."""
    from pcfa_amd import attack_PCFA, sharding
    g = load_golden("universal_raft")
    args, loader = closure_util.universal_case(g)
    gen = torch.Generator().manual_seed(3)
    d1 = 0.01 * torch.randn(3, 128, 160, generator=gen)
    d2 = 0.01 * torch.randn(3, 128, 160, generator=gen)

    def run(dev, use_graph):
        model = closure_util.load_model("RAFT", False, dev)
        ua = attack_PCFA.UniversalAttack(model, d1, d2, dev, 5e5, args, use_graph=use_graph)
        with torch.no_grad():
            ua.nw_delta1.copy_(d1)
            ua.nw_delta2.copy_(d2)
        ua.begin_batch(loader[0][0], loader[0][1])
        loss = float(ua.closure())
        return ua, loss, [p.grad.detach().clone().cpu() for p in ua.params]

    with ops.override_for_testing(oracle_ops):
        _, lc, gc = run(torch.device("cpu"), False)
    ua, le, ge = run(torch.device(DEV), False)
    _, lg, gg = run(torch.device(DEV), True)
    assert abs(le - lc) <= 1e-5 * abs(lc), (le, lc)
    assert abs(lg - le) <= 1e-6 * abs(le)
    for a, b, c in zip(ge, gg, gc):
        assert rel_l2(a, c) < 1e-3, rel_l2(a, c)      # MIOpen vs MKL-DNN convolutions (single-pair closures: 1e-2)
        assert rel_l2(b, a) < 1e-5
    # the buffer of the single collective: [grad delta1 | grad delta2 | loss]
    red = sharding.FlatReducer(ua.params)
    red.pack(torch.tensor(le, device=DEV))
    assert red.flat.numel() == 2 * 3 * 128 * 160 + 1 and float(red.flat[-1]) == pytest.approx(le)
    assert torch.equal(red.flat[:3 * 128 * 160].cpu(), ge[0].flatten())


def test_universal_closure_at_baseline_size_two_pairs_vs_cpu_port(oracle_ops):
    """BASELINE config 5 at size: ONE universal closure, RAFT 436x1024, TWO pairs on one rank (batch-2
    convolutions take other code paths: conv3x3 policy, conv_s2 batch handling, 110-tile sepconv5 grids), a shared non-zero
    delta pair, clipping -- eager and replayed from its hipGraph -- against the CPU port: loss 1e-5, gradient 1e-2
    relative L2 (attack_PCFA.py:427-505)."""
    import bench
    from pcfa_amd import attack_PCFA
    from pcfa_amd.helper_functions import datasets, ownutilities
    h, w = 436, 1024
    args = bench.attack_args("RAFT", "clipping", universal=True)
    pairs = [datasets.synthetic_pair(s, h, w) for s in (100, 101)]
    im1, im2 = torch.stack([p[0] for p in pairs]), torch.stack([p[1] for p in pairs])
    _, [p1, p2] = ownutilities.preprocess_img("RAFT", im1[:1], im2[:1])
    gen = torch.Generator().manual_seed(5)
    d1, d2 = 0.004 * torch.randn(p1[0].shape, generator=gen), 0.004 * torch.randn(p2[0].shape, generator=gen)
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))

    def run(dev, use_graph):
        model = bench.load_model("RAFT", dev, False)
        ua = attack_PCFA.UniversalAttack(model, p1[0], p2[0], dev, attack_PCFA.default_mu(args), args, use_graph=use_graph)
        with torch.no_grad():
            ua.nw_delta1.copy_(d1)
            ua.nw_delta2.copy_(d2)
        ua.begin_batch(im1, im2)
        loss = float(ua.closure())
        if use_graph:
            assert ua.st.graphed is not None
            loss = float(ua.closure())          # a second replay of the same graph
        return loss, [p.grad.detach().clone().cpu() for p in ua.params]

    with ops.override_for_testing(oracle_ops):
        lc, gc = run(torch.device("cpu"), False)
    le, ge = run(torch.device(DEV), False)
    lg, gg = run(torch.device(DEV), True)
    assert abs(le - lc) <= 1e-5 * abs(lc), (le, lc)
    assert abs(lg - le) <= 1e-6 * abs(le), (lg, le)
    for a, b, c in zip(ge, gg, gc):
        assert rel_l2(a, c) < 1e-2, rel_l2(a, c)
        assert rel_l2(b, a) < 1e-5, rel_l2(b, a)


@pytest.mark.parametrize("pair", [0, pytest.param(1, marks=LONG_ONLY)])
def test_schedule_parity_at_baseline_size_vs_cpu_port(pair):
    """The whole schedule at 436x1024 (BASELINE config 2): best-iterate AEE(adv, target), AEE(adv, init) and ||delta|| of
    a PCFA attack on the GPU against the CPU port, inside 3x the port's own spread between two thread counts
    (tools/schedule_parity.py).  3 steps (33 closure evaluations) here; the 20-step form over eight pairs is
    profiles/r04_schedule_parity_matrix.json (tools/parity_matrix.py).  Pair 0 is the pair bench.py times (VERDICT r03
    item 1c) -- the GPU path is bit-reproducible from process to process (test_fresh_processes_are_bit_identical), so
    which side of torch.optim.LBFGS's thresholds a pair lands on is a property of the build, not of the run.

    Background: torch.optim.LBFGS keeps a curvature pair iff y.s > 1e-10 (hard-coded), and on these random-weight problems
    the FIRST pair has y.s = -1.5e-10 .. +5.8e-10 with |y| = 0.7 % of |g| -- the size of the rounding noise of the network
    gradient itself (3e-3 between any two convolution back ends, the CPU port's included).  A pair that sits on that
    threshold runs the fixed-step optimiser's period-3 overshoot cycle one closure apart on the two sides (r03: pair 0 with
    the then-new conv_s2; r04 matrix: pair 6 on the GPU, pair 3 between the port's own two legs); pairs 0 and 1 are on the
    port's branch with this build (tools/parity_arbiter.py records the numbers per leg: profiles/r05/fp64_arbiter.json)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    steps = os.environ.get("PCFA_SCHEDULE_PARITY_STEPS", "3")
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "schedule_parity.py"), "--steps", steps,
                        "--threads", "16,8", "--seed", str(pair)], capture_output=True, text=True, timeout=3000,
                       env=_rank_env())
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["ok"] and r.returncode == 0, out["metrics"]


def test_fp64_arbiter_rule_on_raft_pair6(tmp_path):
    """RAFT pair 6 was the pair of r04's 20-step matrix whose GPU leg ended 0.69 AEE away from the CPU
    port while the port's two thread counts agreed to 5e-5.  tools/parity_arbiter.py evaluates GPU, port@16, port@8 and the port
    in FLOAT64 at the port's first four iterates (the legs' closure losses separate at the third) and asserts
        |g_gpu - g_64| <= max(1e-2 |g_64|, 3 |g_port - g_64|)   at every one of them.
    Recorded (profiles/r05/fp64_arbiter.json): at x0 the fp32 port is 3.8e-3 from float64, this build 4.05e-3, r04's
    F(4x4,3x3) policy 4.28e-3; the first curvature pair has |g0| / |y| = 170 and y.s = 3.0e-10 (GPU), 3.7e-10 (port),
    6.3e-10 (fp64): no leg on the other side of torch LBFGS's 1e-10 gate, every fp32 leg -- the port included -- 40-55 % away
    from the exact y.s, which sets the length of the second move.  (With F(4x4,3x3) restricted to PWC-Net's shapes the pair is
    back inside the matrix: 8/8.)  Also checked here: all legs on one side of the gate, and the GPU within 10x of the port's own
    distance from float64 -- the port's thread-count spread is not the yardstick, its distance from float64 is."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "parity_arbiter.py"), "run", "--net", "RAFT", "--seed", "6",
                        "--at-steps", "0", "--threads", "16", "--floor-threads", "8", "--out", str(tmp_path)],
                       capture_output=True, text=True, timeout=2400, env=_rank_env())
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.load(open(os.path.join(str(tmp_path), "raft_pair6_arbiter.json")))
    assert rec["rule_ok_everywhere"] and len(rec["points"]) == 4
    for p in rec["points"]:
        v = p["grad_rel_l2_vs_fp64"]
        assert v["gpu"] <= max(1e-2, 3 * v["port16"]), p
        assert v["gpu"] <= 10 * max(v["port16"], 1e-6), p
        assert abs(p["loss"]["gpu"] - p["loss"]["port_fp64"]) <= 2e-6 * abs(p["loss"]["port_fp64"]), p
    assert rec["legs_on_one_side_of_the_gate"]
    fp = rec["first_curvature_pair"]
    assert fp["port_fp64"]["amplification_g0_over_y"] > 50          # the ill-conditioning the record documents


@pytest.mark.parametrize("cfg", [("RAFT", "436x1024", []), ("PWCNet", "375x1242", ["--box", "clipping", "--joint"])],
                         ids=["raft", "pwcnet"])
def test_trajectory_closure_parity_vs_cpu_port(cfg):
    """Closure parity ALONG a real trajectory at the BASELINE sizes: the CPU port runs two attack steps (20 closure
    evaluations, the fixed-step optimiser's overshoot points included) and the GPU closure is evaluated at exactly
    those iterates: loss 1e-5 relative, gradient 1e-2 relative L2 at every point (tools/trajectory_closure_parity.py).
    Unlike end-of-attack metrics this cannot land on another branch of torch.optim.LBFGS's hard thresholds (DESIGN.md
    section 4); 3-step records under profiles/r0*_trajectory_closure_parity_*.json.  A point whose gradient misses 1e-2
    against the fp32 port is judged by the port in FP64 (a single LeakyReLU unit of PWC-Net's 6x20 level within rounding
    of zero moves the image gradient by 1.6e-2; r04 found the fp32 PORT on the wrong side of one at point 8):
    |gpu - fp64| <= max(1e-2, 3 |port_fp32 - fp64|)."""
    import json
    import os
    import subprocess
    import sys
    net, size, extra = cfg
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "trajectory_closure_parity.py"), "--net", net, "--size",
                        size, "--steps", "2"] + extra, capture_output=True, text=True, timeout=1500, env=_rank_env())
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["ok"] and r.returncode == 0, (out["max_loss_rel"], out["max_grad_rel_l2"])
    assert len(out["points"]) == 20


def _rank_env():
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                            "MASTER_PORT", "PCFA_SPAWNED_RANK")}
    return env


def test_universal_cosim_two_ranks_on_gpu_equals_single_process(tmp_path):
    """--loss cosim with the batch split over two ranks (VERDICT r02 missing #4): f_cosim is a ratio of sums over the
    GLOBAL batch (losses.py:76-88 in the universal loop attack_PCFA.py:469-490), so the three sums are all-reduced
    between forward and backward.  Two rank processes share this box's GPU (gloo) and run the HIP closure on one pair
    each; their averaged gradient and loss must equal ONE process on the 2-pair batch (MIOpen run-to-run noise)."""
    import os
    import subprocess
    import sys
    from pcfa_amd import launch
    yy, xx = np.meshgrid(np.linspace(-1, 1, 128), np.linspace(-1, 1, 160), indexing="ij")
    tgt = str(tmp_path / "target.npy")
    np.save(tgt, np.stack((1.0 + 0.5 * xx, -0.5 + 0.25 * yy), -1).astype(np.float32))
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_two_rank_gpu.py")
    env = _rank_env()
    assert launch.spawn_ranks([script, "cosim", str(tmp_path / "two"), tgt], 2, env=env, timeout_s=600) == 0
    subprocess.run([sys.executable, script, "cosim", str(tmp_path / "one"), tgt], env=env, check=True, timeout=600)
    r0, r1 = np.load(str(tmp_path / "two_r0.npy")), np.load(str(tmp_path / "two_r1.npy"))
    one = np.load(str(tmp_path / "one_r0.npy"))
    assert np.array_equal(r0, r1)                                   # replicas see identical reduced values
    assert abs(one[-1] - 1.0) > 1e-3 and np.abs(one[:-1]).max() > 0  # the similarity term is live
    assert abs(r0[-1] - one[-1]) <= 1e-5 * abs(one[-1]), (r0[-1], one[-1])
    # batch-1 and batch-2 convolutions take different MIOpen kernels: same bar as the single-pair closures (1e-2;
    # measured 1.8e-3), the loss above is the tight check
    rel = np.linalg.norm(r0[:-1] - one[:-1]) / np.linalg.norm(one[:-1])
    assert rel < 1e-2, rel


def test_rccl_single_rank_collectives_run_on_gpu(tmp_path):
    """RCCL needs one GPU per rank and this box has one, so the multi-rank tests run over gloo.  What CAN run here: every
    collective pcfa_amd.sharding issues, through backend "nccl" with a communicator of one rank -- library load, communicator
    set-up, all-reduce (fp32 SUM, fp64 SUM / MAX) and all-gather (fp64, int64) on device buffers, the universal closure's
    FlatReducer on a side stream.  With one rank every reduction is the identity."""
    import json
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_two_rank_gpu.py")
    out = str(tmp_path / "rccl1.json")
    r = subprocess.run([sys.executable, script, "rccl1", out], env=_rank_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.load(open(out))
    assert rec["backend"] == "nccl"
    assert rec["max"] == 3.25 and rec["all"] == [1.5] and rec["means"] == [2.0, -4.0]
    assert rec["rows"] == [[1.0, 2.0], [3.0, 4.0]]
    assert rec["loss"] == 7.5 and rec["grads_equal"] and rec["collectives"] == 1
    assert rec["batch_sums_world"] == 1 and rec["sums"] == [1.0, 2.0, 3.0]


def test_bench_gpus2_spawns_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no torchrun environment (the driver's command shape) must start two rank
    processes.  On this one-GPU box the ranks share the device and use gloo (PCFA_BENCH_SHARE_GPU / _BACKEND exist for
    exactly this rehearsal); on an 8-GPU node the same launcher gives every rank its own GPU and RCCL."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _rank_env()
    env.update(PCFA_BENCH_BACKEND="gloo", PCFA_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--size", "128x160"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["metric"] == "attack_steps_per_sec" and "rehearsal" not in out
    assert len(out["per_rank_ms_per_step"]) == 2
    assert out["config"]["closure_evals_per_step"] == 10.0
    u = out["universal"]
    assert u["global_batch"] == 2 and u["allreduces_per_closure"] == 1.0
    assert u["allreduce_bytes"] == (2 * 3 * 128 * 160 + 1) * 4


def test_bench_gpus2_headline_survives_an_extra_leg_that_does_not_return():
    """The universal leg runs last and under a deadline (bench.ExtraLegGuard).  With the deadline set to zero the guard
    fires while the leg is still running on both ranks: rank 0 must print the headline line it already has, with the
    reason in the leg's place, and the job must end with status 0."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _rank_env()
    env.update(PCFA_BENCH_BACKEND="gloo", PCFA_BENCH_SHARE_GPU="1", PCFA_BENCH_EXTRA_LEG_DEADLINE_S="0")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--size", "128x160"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and len(out["per_rank_ms_per_step"]) == 2
    assert "error" in out["universal"] and "headline" in out["universal"]["error"]


def test_poison_lds_reaches_the_next_kernel():
    """The control of the test below: after pcfa_poison_lds(pattern) a kernel that reads LDS it never wrote must see the
    pattern, on (nearly) every CU -- otherwise that test proves nothing."""
    from pcfa_amd import _hip
    lib = _hip.load()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.zeros(4096, dtype=torch.int32, device=DEV)
    for pattern in (0x7fc00000, 0x3f800000):
        _hip.check(lib.pcfa_poison_lds(pattern, s), "pcfa_poison_lds")
        _hip.check(lib.pcfa_peek_lds(ctypes.c_void_p(out.data_ptr()), out.numel(), s), "pcfa_peek_lds")
        torch.cuda.synchronize()
        hit = float((out == pattern).float().mean())
        assert hit > 0.95, (hex(pattern), hit)


@pytest.mark.parametrize("net,size,joint,box", [("RAFT", (128, 160), False, "change_of_variables"),
                                                ("RAFT", (436, 1024), False, "change_of_variables"),
                                                ("GMA", (128, 160), False, "change_of_variables"),
                                                ("PWCNet", (64, 96), True, "clipping"),
                                                ("PWCNet", (375, 1242), True, "clipping")])
def test_kernels_do_not_read_lds_they_never_wrote(net, size, joint, box):
    """LDS is not cleared between workgroups.  A kernel that reads LDS it never wrote gets its own earlier workgroups'
    leftovers when it runs alone -- identical from run to run, and plausible enough to stay inside a tolerance -- but
    another pair's kernels' leftovers with two pairs in flight: the kind of defect only a bit-identity test under
    concurrency would ever see, and then only sometimes.  Here pcfa_poison_lds fills every CU's LDS before EVERY entry-point
    call of one eager closure, once with quiet NaNs, once with zeros, once with 1.0: loss and gradients must come out
    finite and bit-identical.  Small sizes make every kernel run its partial-tile paths, BASELINE sizes the full ones."""
    import bench
    from pcfa_amd import _hip, ops
    dev = torch.device(DEV)
    lib = _hip.load()
    st = bench.AttackStepper(net, size[0], size[1], dev, 5, boxconstraint=box, joint=joint, use_graph=False)
    st.optimizer.zero_grad()
    st.closure_body()                      # weight packs, scratch growth
    torch.cuda.synchronize()
    results = []
    for pattern in (0x7fc00000, 0x00000000, 0x3f800000):
        def spy(name, args, invoke, pattern=pattern):
            _hip.check(lib.pcfa_poison_lds(pattern, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "pcfa_poison_lds")
            return invoke(name, *args)
        st.optimizer.zero_grad()
        ops.core.set_call_spy(spy)
        try:
            loss = st.closure_body()
            torch.cuda.synchronize()
        finally:
            ops.core.set_call_spy(None)
        grads = [p.grad.detach().clone() for p in st.optimizer._params]
        assert bool(torch.isfinite(loss.detach()).all()) and all(bool(torch.isfinite(g).all()) for g in grads), hex(pattern)
        results.append((loss.detach().clone(), grads))
    for loss, grads in results[1:]:
        assert torch.equal(loss, results[0][0])
        for a, b in zip(grads, results[0][1]):
            assert torch.equal(a, b), float((a - b).abs().max())


def test_lane_streams_are_nobodys_else():
    """torch.cuda.Stream() hands out 32 pooled HIP streams round-robin: the 33rd object is the first one again.  Lanes tell
    their work apart by the stream handle, so every stream of a lane comes from ops.core.own_stream: no two LIVE objects
    share a handle, a dead object's handle is reused, and the binding of a reused handle is the new owner's."""
    from pcfa_amd import ops
    dev = torch.device(DEV)
    pooled = [torch.cuda.Stream(dev).cuda_stream for _ in range(40)]
    assert len(set(pooled)) < 40                                   # the reason own_stream exists
    live = [ops.core.own_stream(dev, k % 3) for k in range(80)]
    handles = [s.cuda_stream for s in live]
    assert len(set(handles)) == 80 and not (set(handles) & set(pooled))
    for k, s in enumerate(live):
        with torch.cuda.stream(s):
            assert ops.core.current_lane() == k % 3
    x = torch.ones(1 << 16, device=dev)
    with torch.cuda.stream(live[5]):                               # an ordinary stream: work runs and synchronises on it
        y = (x * 3).sum()
    live[5].synchronize()
    assert float(y) == 3.0 * (1 << 16)
    gone = live[7].cuda_stream
    ops.core.release_stream(live[7])                                # what an owner does when it is done (explicitly:
    del live[7], s                                                  # torch's stream objects cannot carry a weak reference)
    again = ops.core.own_stream(dev, 0)
    assert again.cuda_stream == gone                                # reused, not leaked ...
    with torch.cuda.stream(again):
        assert ops.core.current_lane() == 0                         # ... and its old lane-1 binding is gone
    for s in live + [again]:
        ops.core.release_stream(s)
    run = ops.core.lane_run_streams(dev, 4)                         # the lanes' run streams: persistent, distinct, bound
    assert len({s.cuda_stream for s in run}) == 4 and not ({s.cuda_stream for s in run} & set(handles))
    assert [s.cuda_stream for s in ops.core.lane_run_streams(dev, 2)] == [s.cuda_stream for s in run[:2]]
    for k, s in enumerate(run):
        with torch.cuda.stream(s):
            assert ops.core.current_lane() == k


def test_lanes_never_share_conv_scratch_in_forward_or_backward():
    """Two pairs in flight may only share the frozen weights.  At a small map size the 3x3 convolutions slice K over
    workgroups and keep their partial sums in a scratch buffer per (device, main | side stream, lane).  With
    Config.overlap_encoders (opt-in) the context encoder runs on a side stream, and autograd runs its backward nodes on a
    thread of its own, where only the STREAM says which lane a launch belongs to: the side stream is therefore one per
    (device, lane) and bound to its lane (r05; one per device before, so that lane 1's context-encoder backward took lane
    0's scratch).  One eager closure per lane with every scratch pointer recorded: main + side scratch in both lanes, and
    the lanes' sets disjoint."""
    import dataclasses
    import bench
    from pcfa_amd import config as pcfa_config
    from pcfa_amd import ops
    dev = torch.device(DEV)
    model = bench.load_model("RAFT", dev, True, dataclasses.replace(pcfa_config.DEFAULT, overlap_encoders=True))
    used = {0: set(), 1: set()}
    streams = [ops.core.own_stream(dev, k) for k in (0, 1)]   # (released at the end of the test)
    for k in (0, 1):
        def spy(name, args, invoke, k=k):
            if name == "pcfa_conv3x3_run" and args[14].value:
                used[k].add(int(args[14].value))
            return invoke(name, *args)
        with ops.core.lane(k), torch.cuda.stream(streams[k]):
            st = bench.AttackStepper("RAFT", 128, 160, dev, 20 + k, use_graph=False, model=model)
            ops.core.set_call_spy(spy)
            try:
                st.optimizer.zero_grad()
                st.closure_body()
                torch.cuda.synchronize()
            finally:
                ops.core.set_call_spy(None)
    for stream in streams:
        ops.core.release_stream(stream)
    assert len(used[0]) == 2 and len(used[1]) == 2, used      # main-stream and side-stream scratch, both lanes
    assert not (used[0] & used[1]), used


def test_pairs_in_flight_bit_identical_to_solo():
    """two PairAttacks side by side on one GPU (attack_PCFA.PairsInFlight: one thread + stream + graph
    set + scratch lane per pair) must leave every pair with exactly the bits of a solo attack of that pair."""
    import bench
    from pcfa_amd import attack_PCFA
    dev = torch.device(DEV)
    model = closure_util.load_model("RAFT", True, dev)
    if hasattr(model, "_pcfa_pair_graphs"):
        model._pcfa_pair_graphs.clear()
    seeds, steps = (11, 12), 2
    flight = attack_PCFA.PairsInFlight(
        lambda k: bench.AttackStepper("RAFT", 128, 160, dev, seeds[k], use_graph=True, model=model), 2, dev)
    assert all(st.graphed is not None for st in flight.attacks)
    assert flight.attacks[0].graph_key != flight.attacks[1].graph_key and not flight.attacks[0].retired   # one set per lane
    last = flight.run(steps)
    for k, seed in enumerate(seeds):
        model._pcfa_pair_graphs.clear()
        solo = bench.AttackStepper("RAFT", 128, 160, dev, seed, use_graph=True, model=model, )
        for _ in range(steps):
            solo_last = solo.step()
        a = flight.attacks[k]
        assert tuple(last[k]) == tuple(solo_last), (k, last[k], solo_last)
        assert torch.equal(a.delta1, solo.delta1) and torch.equal(a.delta2, solo.delta2)
        assert torch.equal(a.flow_pred, solo.flow_pred) and a.closures == solo.closures == 10 * steps
        del solo
    model._pcfa_pair_graphs.clear()


def test_pairs_in_flight_with_overlapped_encoders_bit_identical_to_solo():
    """The same with Config.overlap_encoders (opt-in): every lane's context encoder runs on that lane's own side stream, a
    parallel branch of its captured graphs."""
    import dataclasses
    import bench
    from pcfa_amd import attack_PCFA
    from pcfa_amd import config as pcfa_config
    dev = torch.device(DEV)
    model = bench.load_model("RAFT", dev, True, dataclasses.replace(pcfa_config.DEFAULT, overlap_encoders=True))
    seeds = (21, 22)
    flight = attack_PCFA.PairsInFlight(
        lambda k: bench.AttackStepper("RAFT", 128, 160, dev, seeds[k], use_graph=True, model=model), 2, dev)
    assert all(st.graphed is not None for st in flight.attacks)
    last = flight.run(1)
    for k, seed in enumerate(seeds):
        model._pcfa_pair_graphs.clear()
        solo = bench.AttackStepper("RAFT", 128, 160, dev, seed, use_graph=True, model=model)
        solo_last = solo.step()
        a = flight.attacks[k]
        assert tuple(last[k]) == tuple(solo_last), (k, last[k], solo_last)
        assert torch.equal(a.delta1, solo.delta1) and torch.equal(a.delta2, solo.delta2)
        del solo
    model._pcfa_pair_graphs.clear()


@pytest.mark.parametrize("shape", [(2, 128, 256, 55, 128), (1, 256, 576, 55, 128), (1, 96, 40, 7, 9), (3, 7, 5, 4, 4)])
def test_conv1x1_vs_library(shape):
    """ops.conv1x1 (Config.conv1x1 = "hip": the encoders' output layer and the mask head's 1x1 layer on pcfa_gemm_f32,
    models/raft/extractor.py:146,186, update.py:118-121) against the library convolution in float64: forward and data
    gradient, 2e-6 relative to the largest value (fp32 fma chains over K <= 256)."""
    B, K, N, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, K, H, W, generator=g)
    w = torch.randn(N, K, 1, 1, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    go = torch.randn(B, N, H, W, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    y = hip_ops.conv1x1(xd, w.to(DEV), b.to(DEV))
    y.backward(go.to(DEV))
    x64 = x.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w.double(), b.double())
    y64.backward(go.double())
    assert max_abs(y, y64.float()) <= 2e-6 * float(y64.abs().max())
    assert max_abs(xd.grad, x64.grad.float()) <= 2e-6 * float(x64.grad.abs().max())
    with pytest.raises(RuntimeError, match="frozen"):
        wd = w.to(DEV).requires_grad_(True)
        hip_ops.conv1x1(x.to(DEV), wd, None).sum().backward()


def test_raft_closure_without_any_library_kernel():
    """Config.conv1x1 = "hip": a RAFT closure then launches no library GEMM / convolution kernel at all (torch.profiler sees
    no Tensile `Cijk_` and no MIOpen kernel), agrees with the default build's closure to fp32 noise, and two pairs in flight
    still equal their solo runs bit for bit."""
    import dataclasses
    import bench
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    from pcfa_amd import attack_PCFA
    from pcfa_amd import config as pcfa_config
    dev = torch.device(DEV)
    own = bench.load_model("RAFT", dev, True, dataclasses.replace(pcfa_config.DEFAULT, conv1x1="hip"))
    lib = bench.load_model("RAFT", dev, True)
    res = []
    for model in (own, lib):
        st = bench.AttackStepper("RAFT", 128, 160, dev, 3, use_graph=False, model=model)
        st.optimizer.zero_grad()
        st.closure_body()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            st.optimizer.zero_grad()
            loss = st.closure_body()
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if e.device_type == DeviceType.CUDA]
        res.append((float(loss), [p.grad.detach().clone() for p in st.optimizer._params],
                    [n for n in names if n.startswith("Cijk_") or "miopen" in n.lower() or "MIOpen" in n]))
    (l_own, g_own, k_own), (l_lib, g_lib, k_lib) = res
    assert k_lib and not k_own, (k_own[:3], len(k_lib))
    assert abs(l_own - l_lib) <= 1e-5 * abs(l_lib)
    for a, b in zip(g_own, g_lib):
        assert rel_l2(a, b) < 1e-3, rel_l2(a, b)
    flight = attack_PCFA.PairsInFlight(
        lambda k: bench.AttackStepper("RAFT", 128, 160, dev, 41 + k, use_graph=True, model=own), 2, dev)
    last = flight.run(1)
    for k in (0, 1):
        own._pcfa_pair_graphs.clear()
        solo = bench.AttackStepper("RAFT", 128, 160, dev, 41 + k, use_graph=True, model=own)
        assert tuple(solo.step()) == tuple(last[k]), k
        del solo
    own._pcfa_pair_graphs.clear()


def test_pairs_in_flight_refuses_closures_with_shared_library_workspaces():
    """GMA's attention products run on rocBLAS by default, and every graph of a process is captured with ONE rocBLAS handle,
    whose device workspace all its launches share: two lanes replaying side by side hung in their first step (r05).  Such a
    model is refused in flight; the build with the package's own products (gma_gemm='hip') runs, bit-identical to solo."""
    import dataclasses
    import bench
    from pcfa_amd import attack_PCFA
    from pcfa_amd import config as pcfa_config
    dev = torch.device(DEV)
    lib = bench.load_model("GMA", dev, True)
    assert pcfa_config.cfg(lib).gma_gemm == "lib"
    with pytest.raises(ValueError, match="gma_gemm='hip'"):
        attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("GMA", 128, 160, dev, 31 + k, use_graph=True, model=lib), 2, dev)
    del lib
    hip = bench.load_model("GMA", dev, True, dataclasses.replace(pcfa_config.DEFAULT, gma_gemm="hip"))
    flight = attack_PCFA.PairsInFlight(
        lambda k: bench.AttackStepper("GMA", 128, 160, dev, 31 + k, use_graph=True, model=hip), 2, dev)
    last = flight.run(1)
    for k in (0, 1):
        hip._pcfa_pair_graphs.clear()
        solo = bench.AttackStepper("GMA", 128, 160, dev, 31 + k, use_graph=True, model=hip)
        assert tuple(solo.step()) == tuple(last[k]), k
        assert torch.equal(flight.attacks[k].delta1, solo.delta1)
        del solo
    hip._pcfa_pair_graphs.clear()


def test_bench_default_command_prints_one_short_strict_json_line(tmp_path):
    """`python bench.py --gpus 1 --steps K --warmup W` (the driver's command, every leg on, at the
    BASELINE size) prints ONE line under 6000 bytes of strict JSON that carries the contract's keys, `roofline` and
    `cpu_baseline`; the full record goes to the detail file the line names."""
    import json
    import os
    import subprocess
    import sys
    import bench
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCFA_BENCH_DETAIL=str(tmp_path / "detail.json"))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--cpu-closures", "2"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < bench.LINE_BUDGET, len(lines[0])

    def bad(c):
        raise ValueError("non-strict JSON constant %r" % c)
    out = json.loads(lines[0], parse_constant=bad)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["steps"] == 2 and out["warmup"] == 1 and out["n_gpus"] == 1
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert 0 < out["roofline"]["frac"] < 1
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert "error" not in out.get("pwcnet", {}) and "error" not in out.get("gma", {})
    detail = json.load(open(env["PCFA_BENCH_DETAIL"]))
    assert "kernel_families" in detail and "kernels" in detail and out["detail"] == bench.DETAIL_FILE


@pytest.mark.parametrize("net,size", [("RAFT", (128, 160)), ("FlowNet2", (64, 128))])
def test_graphed_closure_matches_eager(net, size):
    """The hipGraph replay of a closure reproduces the eager launch: same kernels in the same order; the loss
    agrees to 1e-6 and the gradient to 1e-5 relative L2 (a few MIOpen backward kernels accumulate with atomics,
    so even two eager launches differ in the last bits).  FlowNet2: regression test for the Resample2d backward,
    whose hipMemsetAsync of grad_in1 was not replayed with the graph (gradients grew from replay to replay)."""
    import bench
    torch.backends.cudnn.benchmark = False
    st = bench.AttackStepper(net, size[0], size[1], torch.device(DEV), seed=3)
    st.optimizer.zero_grad()
    l_eager = float(st._closure_body())   # keep no autograd node alive: a live AccumulateGrad of nw1/nw2 would
                                          # pin the default stream into the capture (see graphed.py)
    g_eager = [st.nw1.grad.clone(), st.nw2.grad.clone()]
    st.enable_graph()
    with torch.no_grad():
        st.nw1.add_(0.01)          # move the variables in place, as L-BFGS does
    l_graph_moved = float(st.closure())
    with torch.no_grad():
        st.nw1.sub_(0.01)
    l_graph = float(st.closure())
    assert abs(l_graph - l_eager) <= 1e-6 * abs(l_eager) and abs(l_graph_moved - l_eager) > 1e-6 * abs(l_eager)
    tol = 1e-5 if net == "RAFT" else 1e-4  # FlowNet2: fp32 atomics in the warp's scatter and in MIOpen's deconvolutions
    assert rel_l2(st.nw1.grad, g_eager[0]) < tol and rel_l2(st.nw2.grad, g_eager[1]) < tol
    l_again = float(st.closure())
    assert abs(l_again - l_graph) <= 1e-6 * abs(l_graph)
    assert rel_l2(st.nw1.grad, g_eager[0]) < tol and rel_l2(st.nw2.grad, g_eager[1]) < tol


def test_pair_graph_reuse_equals_fresh_capture():
    """attack_l2 loops over equal-shape pairs (attack_PCFA.py:668-670): from the second pair on PairAttack adopts the
    first pair's static buffers, hipGraphs and optimiser (images / variables / target copied in, optimiser reset)
    instead of warming up and capturing again.  Every pair must come out as from a fresh capture.  Compared on the
    unattacked flow (bit-level: forward only) and on the loss of the first six closure evaluations (1e-5 relative; 3e-2 on overshooting steps):
    this 128x160 random-weight problem bifurcates around the tenth evaluation under last-bit noise (identical fresh
    runs end 2 % to 180 % apart: an r03 probe), so later iterates say nothing about the mechanism."""
    from pcfa_amd import attack_PCFA
    from pcfa_amd.helper_functions import datasets
    dev = torch.device(DEV)
    args = closure_util.cli_args(net="RAFT", steps=1)
    mu = attack_PCFA.default_mu(args)

    def run(reuse):
        model = closure_util.load_model("RAFT", True, dev)      # shared with other tests: start from an empty cache
        if hasattr(model, "_pcfa_pair_graphs"):
            model._pcfa_pair_graphs.clear()
        out = []
        for seed, (h, w) in ((0, (128, 160)), (1, (128, 160)), (2, (136, 168)), (3, (128, 160))):
            i1, i2, _ = datasets.synthetic_pair(seed, h, w)
            st = attack_PCFA.PairAttack(model, i1[None], i2[None], None, seed, attack_PCFA.EPS_BOX, dev, False, mu,
                                        args, use_graph=True, reuse_graphs=reuse)
            assert st.graphed is not None
            losses, inner = [], st.closure

            def recording():
                loss = inner()
                losses.append(float(loss.detach()))
                return loss
            st.closure = recording
            st.step()
            out.append((st.graphs_reused, losses, st.flow_pred_init.clone(), st.closures, st.aee_tgt))
        return out, len(getattr(model, "_pcfa_pair_graphs", {}))

    reused, n_kept = run(True)
    fresh, n_none = run(False)
    # the donor of an adopted graph set is retired (its variables / optimiser are the new pair's now), and
    # the per-model cache is bounded (LRU of Config.max_cached_shapes shapes; ADVICE r04: the cap is a Config field,
    # default 4 -- set to 2 on this model's top module for the eviction check, restored below)
    import dataclasses
    from pcfa_amd import config as pcfa_config
    model = closure_util.load_model("RAFT", True, dev)
    model._pcfa_pair_graphs.clear()
    cfg0 = pcfa_config.cfg(model)
    assert cfg0.max_cached_shapes == 4
    object.__setattr__(model, "_pcfa_config", dataclasses.replace(cfg0, max_cached_shapes=2))
    sts = []
    for seed, (h, w) in ((0, (128, 160)), (1, (128, 160)), (2, (136, 168)), (3, (144, 176))):
        i1, i2, _ = datasets.synthetic_pair(seed, h, w)
        sts.append(attack_PCFA.PairAttack(model, i1[None], i2[None], None, seed, attack_PCFA.EPS_BOX, dev, False, mu, args,
                                          use_graph=True, reuse_graphs=True))
    assert sts[0].retired and sts[0].graphed is None            # adopted by the second pair
    with pytest.raises(RuntimeError, match="retired"):
        sts[0].step()
    assert sts[1].retired and not sts[2].retired and not sts[3].retired   # three shapes, two kept: the oldest is evicted
    assert len(model._pcfa_pair_graphs) == 2
    sts[3].step()
    model._pcfa_pair_graphs.clear()
    object.__setattr__(model, "_pcfa_config", cfg0)
    assert [r[0] for r in reused] == [False, True, False, True] and n_kept == 2       # two shapes, two graph sets
    assert not any(r[0] for r in fresh) and n_none == 0
    for (_, la, fa, ca, ta), (_, lb, fb, cb, tb) in zip(reused, fresh):
        assert ca == cb == 10 and len(la) == len(lb) == 10
        assert max_abs(fa, fb) < 1e-4 and abs(ta - tb) < 1e-5
        for k in range(6):   # an un-damped L-BFGS step that overshoots (loss >> start) amplifies last-bit noise
            # from the third evaluation on the iterate depends on the first curvature pair, whose y is a difference of
            # nearly equal gradients (|y| < 1 % of |g|): run-to-run noise of the library kernels that
            # are left (atomics) shows up there at ~5e-5
            tol = (1e-5 if k < 2 else 3e-4) if lb[k] < 1.5 * lb[0] else 3e-2
            assert abs(la[k] - lb[k]) <= tol * abs(lb[k]), (k, la, lb)


def test_split_closure_shares_the_reprediction_forward():
    """graphed.SplitGraphedClosure (PCFA_SHARED_FORWARD=1): forward and backward captured as two graphs.  A closure call
    after forward() replays only the backward and must give the loss and gradient of forward + backward back to back --
    and both must match the eager closure (1e-6 / 1e-5 as in test_graphed_closure_matches_eager); moving the variables
    invalidates nothing by itself, the caller does (here: a plain closure call re-runs the forward)."""
    import bench
    torch.backends.cudnn.benchmark = False
    st = bench.AttackStepper("RAFT", 128, 160, torch.device(DEV), seed=3)
    st.optimizer.zero_grad()
    l_eager = float(st._closure_body())
    g_eager = [st.nw1.grad.clone(), st.nw2.grad.clone()]
    st.enable_graph(share_forward=True)
    assert st.graphed is not None and hasattr(st.graphed, "forward")
    for _ in range(2):                                   # forward + backward
        l = float(st.closure())
        assert abs(l - l_eager) <= 1e-6 * abs(l_eager)
        assert rel_l2(st.nw1.grad, g_eager[0]) < 1e-5 and rel_l2(st.nw2.grad, g_eager[1]) < 1e-5
    (d1, d2), flow = st.repredict()                      # the re-prediction ...
    assert flow.shape[-2:] == (128, 160) and not flow.requires_grad
    shared = st.graphed.forwards_shared
    l = float(st.closure())                              # ... is the forward of this evaluation: backward only
    assert st.graphed.forwards_shared == shared + 1
    assert abs(l - l_eager) <= 1e-6 * abs(l_eager)
    assert rel_l2(st.nw1.grad, g_eager[0]) < 1e-5 and rel_l2(st.nw2.grad, g_eager[1]) < 1e-5
    with torch.no_grad():
        st.nw1.add_(0.01)
    assert abs(float(st.closure()) - l_eager) > 1e-6 * abs(l_eager)   # moved variables: the forward runs again


# --------------------------------------------------------------------------- #
# L-BFGS: pcfa_amd.lbfgs.LBFGS (HIP vector math) against torch.optim.LBFGS, the optimiser of the reference loop
# (attack_PCFA.py:97,114).  Tolerance: the two run the same operation sequence and differ only in the summation
# order inside dot products, so iterates agree to ~1e-5 relative on a smooth, well-conditioned objective.
# --------------------------------------------------------------------------- #
def _lbfgs_problem(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    x0 = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    c = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    w = [(0.5 + torch.rand(s, generator=g)).to(DEV) for s in shapes]

    def make(params):
        def closure():
            for p in params:
                p.grad = None
            loss = 0.
            for p, ci, wi in zip(params, c, w):
                r = p - ci
                v = r.reshape(-1)
                mixed = v + 0.5 * torch.roll(v, 1) - 0.3 * torch.roll(v, 7)      # a fixed banded linear map
                loss = loss + 0.5 * (wi.reshape(-1) * mixed * mixed).sum() + 0.01 * (v ** 4).sum()
            loss = loss / sum(p.numel() for p in params)
            loss.backward()
            return loss
        return closure
    return x0, make


@pytest.mark.parametrize("direction", ["gram", "two_loop"])
@pytest.mark.parametrize("history", [100, 4])
def test_lbfgs_matches_torch_optimizer(history, direction):
    """Tolerance 2e-4 relative on the iterates after 40 iterations for BOTH forms: "two_loop" runs torch's operation
    sequence (only the summation order inside a dot product differs), "gram" (the default) evaluates the same
    recursion on stored inner products -- a different rounding order, fp64 in the coefficient space."""
    from pcfa_amd.lbfgs import LBFGS
    shapes = [(3, 17, 33), (1001,)]          # 2684 elements: exercises the n % 4 tail and two parameter tensors
    x0, make = _lbfgs_problem(5, shapes)
    pa = [x.clone().requires_grad_(True) for x in x0]
    pb = [x.clone().requires_grad_(True) for x in x0]
    oa = torch.optim.LBFGS(pa, max_iter=10, history_size=history)
    ob = LBFGS(pb, max_iter=10, history_size=history, direction=direction)
    ca, cb = make(pa), make(pb)
    for step in range(4):                    # 40 iterations: the 4-pair ring wraps many times
        la, lb = float(oa.step(ca)), float(ob.step(cb))
        assert abs(la - lb) <= 2e-4 * abs(la) + 1e-7, (step, la, lb)
        for a, b in zip(pa, pb):
            assert rel_l2(b.detach(), a.detach()) < 2e-4, (step, rel_l2(b.detach(), a.detach()))
    assert oa.state[pa[0]]["n_iter"] == ob.state[pb[0]]["n_iter"]
    assert oa.state[pa[0]]["func_evals"] == ob.state[pb[0]]["func_evals"]
    assert ob.history_count() == min(history, len(oa.state[pa[0]]["old_dirs"]))
    # host synchronisations per iteration: closure scalars + direction scalars (+ the curvature test in two_loop)
    assert ob.host_syncs <= 4 * (1 + 10 * (2 if direction == "gram" else 3))
    assert la < 0.05 * float(make([x.clone().requires_grad_(True) for x in x0])())   # and it actually minimises


def test_lbfgs_two_loop_against_dense_reference():
    """pcfa_lbfgs_direction alone against the textbook two-loop recursion in float64 on the same (S, Y) history."""
    from pcfa_amd.hip_ops import _call, _ptr
    g = torch.Generator().manual_seed(11)
    n, m, rows, first = 4099, 6, 9, 5         # ring that wraps: rows 5,6,7,8,0,1
    ld = (n + 3) // 4 * 4
    S = torch.randn(rows, ld, generator=g).to(DEV)
    Y = (S + 0.3 * torch.randn(rows, ld, generator=g).to(DEV)).contiguous()   # y.s > 0
    grad = torch.randn(ld, generator=g).to(DEV)
    order = [(first + k) % rows for k in range(m)]
    ro = torch.zeros(rows, device=DEV)
    for r in order:
        ro[r] = 1.0 / (Y[r, :n] @ S[r, :n])
    newest = order[-1]
    H = ((Y[newest, :n] @ S[newest, :n]) / (Y[newest, :n] @ Y[newest, :n])).reshape(1)
    al, d = torch.empty(rows, device=DEV), torch.empty(ld, device=DEV)
    ws = torch.empty(int(hip_ops._hip.load().pcfa_lbfgs_workspace_floats()), device=DEV)
    _call("pcfa_lbfgs_direction", _ptr(grad), _ptr(S), _ptr(Y), _ptr(ro), _ptr(H), _ptr(al), _ptr(d), _ptr(ws),
          first, m, rows, ld, n)
    q = -grad[:n].double()
    Sd, Yd, rod = S[:, :n].double(), Y[:, :n].double(), ro.double()
    a = {}
    for r in reversed(order):
        a[r] = (Sd[r] @ q) * rod[r]
        q = q - a[r] * Yd[r]
    rr = q * H.double()
    for r in order:
        be = (Yd[r] @ rr) * rod[r]
        rr = rr + (a[r] - be) * Sd[r]
    assert rel_l2(d[:n].double(), rr) < 1e-5


def test_lbfgs_gram_form_against_dense_two_loop():
    """pcfa_lbfgs_gram_update / _direction (the default optimiser path) against the textbook two-loop recursion in
    float64: 9 candidate pairs fed through a 6-pair ring (it wraps), one of them with y.s < 0 (the optimiser's
    curvature test must reject it on the device), checked after every update."""
    from pcfa_amd.hip_ops import _call, _ptr
    lib = hip_ops._hip.load()
    g = torch.Generator().manual_seed(13)
    n, cap, t_step = 4099, 6, 0.75
    ld = (n + 3) // 4 * 4
    rows = cap + 1
    state = torch.zeros(int(lib.pcfa_lbfgs_gram_state_bytes(cap)), dtype=torch.uint8, device=DEV)
    ws = torch.empty(int(lib.pcfa_lbfgs_gram_workspace_bytes(cap, ld)) // 4, device=DEV)
    S, Y = torch.zeros(rows, ld, device=DEV), torch.zeros(rows, ld, device=DEV)
    grad, g_prev, d = torch.zeros(ld, device=DEV), torch.zeros(ld, device=DEV), torch.zeros(ld, device=DEV)
    out2 = torch.empty(2, device=DEV)
    grad[:n] = torch.randn(n, generator=g).to(DEV)
    g_prev.copy_(grad)
    _call("pcfa_lbfgs_gram_reset", _ptr(state), cap)
    kept, H = [], 1.0
    for k in range(9):
        s_k = torch.randn(n, generator=g).to(DEV)
        y_k = (s_k + 0.3 * torch.randn(n, generator=g).to(DEV)) * (-1.0 if k == 4 else 1.0)
        d[:n] = s_k / t_step
        grad[:n] = g_prev[:n] + y_k                      # the kernel forms y = g - g_prev, s = t d itself
        s_true, y_true = (d[:n] * t_step).double(), (grad[:n] - g_prev[:n]).double()
        _call("pcfa_lbfgs_gram_update", _ptr(grad), _ptr(g_prev), _ptr(d), t_step, _ptr(S), _ptr(Y), _ptr(state),
              _ptr(ws), cap, ld)
        _call("pcfa_lbfgs_gram_direction", _ptr(grad), _ptr(S), _ptr(Y), _ptr(state), _ptr(d), _ptr(out2), _ptr(ws),
              cap, ld)
        hdr = state[:16].view(torch.int32).tolist()
        ys = float(y_true @ s_true)
        assert hdr[2] == (1 if ys > 1e-10 else 0), (k, hdr, ys)
        if ys > 1e-10:
            kept.append((s_true, y_true))
            kept = kept[-cap:]
            H = ys / float(y_true @ y_true)
        assert hdr[1] == len(kept) and hdr[3] == rows
        assert torch.equal(g_prev, grad)
        q = -grad[:n].double()
        al = []
        for s_i, y_i in reversed(kept):
            a = (s_i @ q) / (y_i @ s_i)
            al.append(a)
            q = q - a * y_i
        r = q * H
        for (s_i, y_i), a in zip(kept, reversed(al)):
            be = (y_i @ r) / (y_i @ s_i)
            r = r + (a - be) * s_i
        assert rel_l2(d[:n].double(), r) < 2e-5, (k, rel_l2(d[:n].double(), r))
        gtd, dmax = out2.tolist()
        assert abs(gtd - float(grad[:n].double() @ r)) <= 2e-4 * abs(float(grad[:n].double() @ r))
        assert abs(dmax - float(r.abs().max())) <= 1e-4 * float(r.abs().max())
        assert float(d[n:].abs().max()) == 0.0 if ld > n else True


def test_lbfgs_rejects_cpu_parameters():
    from pcfa_amd.lbfgs import LBFGS
    with pytest.raises(RuntimeError):
        LBFGS([torch.zeros(4, requires_grad=True)], max_iter=10)
    with pytest.raises(NotImplementedError):
        LBFGS([torch.zeros(4, device=DEV, requires_grad=True)], line_search_fn="strong_wolfe")


# --------------------------------------------------------------------------- #
# Boundary as files: the drop-in modules of pcfa_amd/dropin carry the names the reference imports
# (correlation_sampler.cpp:114-124 pybind names; correlation_cuda.cc / resample2d_cuda.cc / channelnorm_cuda.cc).
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("tag", ["pwc_a", "pwc_b", "pwc_c", "gen_a", "gen_b"])
def test_dropin_sampler_backend_positional_signature(tag):
    """`import spatial_correlation_sampler_backend` resolves to the drop-in; forward / backward are called with the
    reference's 12 positional integers (correlation_sampler.cpp:58-112) against the reference sampler's goldens."""
    from pcfa_amd import dropin
    dropin.install()
    import spatial_correlation_sampler_backend as backend
    assert backend.__file__.startswith(dropin.install())
    g = load_golden("spatial_corr_" + tag)
    ks, ps, st, pad, dil, dp = (int(v) for v in g["params"])
    a, b = t(g["in1"], DEV), t(g["in2"], DEV)
    out = backend.forward(a, b, ks, ks, ps, ps, pad, pad, dil, dil, dp, dp, st, st)
    assert tuple(out.shape) == g["out"].shape
    assert max_abs(out, t(g["out"])) <= 1e-6 * a.shape[1] * float(np.abs(g["out"]).max()) + 1e-6
    g1, g2 = backend.backward(a, b, t(g["grad_out"], DEV), ks, ks, ps, ps, pad, pad, dil, dil, dp, dp, st, st)
    assert rel_l2(g1, t(g["gin1"])) < 1e-5 and rel_l2(g2, t(g["gin2"])) < 1e-5
    with pytest.raises(RuntimeError):
        backend.forward(a.cpu(), b.cpu(), ks, ks, ps, ps, pad, pad, dil, dil, dp, dp, st, st)


def test_dropin_flownet_extensions_positional_signature(oracle_ops):
    """correlation_cuda / resample2d_cuda / channelnorm_cuda under their reference names: caller-allocated outputs,
    return value 1 (correlation_cuda.cc:10-87,89-167; resample2d_cuda.cc:6-24; channelnorm_cuda.cc:6-25)."""
    from pcfa_amd import dropin
    dropin.install()
    import channelnorm_cuda
    import correlation_cuda
    import resample2d_cuda
    gen = torch.Generator().manual_seed(5)
    a = torch.randn(1, 16, 12, 20, generator=gen)
    b = torch.randn(1, 16, 12, 20, generator=gen)
    want = oracle_ops.flownet_correlation(a.clone().requires_grad_(True), b.clone().requires_grad_(True), 20, 1, 20, 1, 2)
    out, r1, r2 = torch.empty(0, device=DEV), torch.empty(0, device=DEV), torch.empty(0, device=DEV)
    assert correlation_cuda.forward(a.to(DEV), b.to(DEV), r1, r2, out, 20, 1, 20, 1, 2, 1) == 1
    assert out.shape == want.shape and max_abs(out, want) <= 2e-6 * 16 * float(want.abs().max())
    go = torch.randn(want.shape, generator=gen)
    ga, gb = oracle_ops.flownet_corr_backward(a, b, go, 20, 1, 20, 1, 2)
    g1, g2 = torch.empty(0, device=DEV), torch.empty(0, device=DEV)
    assert correlation_cuda.backward(a.to(DEV), b.to(DEV), r1, r2, go.to(DEV), g1, g2, 20, 1, 20, 1, 2, 1) == 1
    assert rel_l2(g1, ga) < 1e-5 and rel_l2(g2, gb) < 1e-5
    img = torch.randn(1, 3, 16, 24, generator=gen)
    flo = 2 * torch.randn(1, 2, 16, 24, generator=gen)
    outr = torch.empty(1, 3, 16, 24, device=DEV)
    assert resample2d_cuda.forward(img.to(DEV), flo.to(DEV), outr, 1, True) == 1
    assert max_abs(outr, oracle_ops.resample2d_forward(img, flo, 1, True)) <= 1e-5 * float(img.abs().max())
    gr = torch.randn(1, 3, 16, 24, generator=gen)
    wa, wb = oracle_ops.resample2d_backward(img, flo, gr)
    gi, gf = torch.empty_like(outr), torch.empty(1, 2, 16, 24, device=DEV)
    assert resample2d_cuda.backward(img.to(DEV), flo.to(DEV), gr.to(DEV), gi, gf, 1, True) == 1
    assert rel_l2(gi, wa) < 1e-5 and rel_l2(gf, wb) < 1e-4
    x = torch.randn(2, 3, 8, 8, generator=gen)
    outn = torch.empty(2, 1, 8, 8, device=DEV)
    assert channelnorm_cuda.forward(x.to(DEV), outn, 2) == 1
    wn = oracle_ops.channelnorm_forward(x)
    assert max_abs(outn, wn) <= 2e-7 * float(wn.abs().max()) + 1e-7
    gn = torch.randn(2, 1, 8, 8, generator=gen)
    gx = torch.empty(2, 3, 8, 8, device=DEV)
    assert channelnorm_cuda.backward(x.to(DEV), outn, gn.to(DEV), gx, 2) == 1
    assert rel_l2(gx, oracle_ops.channelnorm_backward(x, wn, gn)) < 1e-6


def test_losses_names_on_gpu_vs_reference_golden():
    """avg_mse / f_mse / f_cosim / two_norm_avg_delta_squared / relu_penalty (losses.py:32-88,110-126,177-197) on the
    fused HIP loss kernels against the reference's values and gradients."""
    from pcfa_amd.helper_functions import losses
    from tests.test_oracle_cpu import _check_losses_names
    _check_losses_names(losses, load_golden("losses_names"), DEV, 2e-6)


# --------------------------------------------------------------------------- #
# SURVEY 8f row f3 on the GPU: attack_FGSM.py:21-56,59-308 and evaluate_PCFA.py:21-79,86-299
# --------------------------------------------------------------------------- #
def _f3_cli(**kw):
    return _cli_args(**dict(dict(boxconstraint="clipping", steps=2, epochs=2, epsilon=0.00025,
                                 perturbation_sourcefolder=None, origin_net=None, synthetic_size="64x64"), **kw))


def test_fgsm_driver_on_gpu_vs_cpu_port(oracle_ops):
    """I-FGSM (attack_FGSM.py:21-56 step, :59-308 driver): 2 pairs x 2 sign-gradient iterations on the MI355X against
    the CPU port.  A sign step is discontinuous, so the comparison is on the metrics (1e-3) and on the perturbation
    norm, which is exact by construction (every element moves by +-epsilon per iteration)."""
    from pcfa_amd import attack_FGSM
    a = _f3_cli(loss="mse")
    got = attack_FGSM.attack(a)
    import importlib
    import os
    os.environ["PCFA_USE_CPU"] = "1"
    try:
        from pcfa_amd.helper_functions import config_paths
        importlib.reload(config_paths)
        importlib.reload(attack_FGSM)
        with ops.override_for_testing(oracle_ops):
            want = attack_FGSM.attack(a)
    finally:
        os.environ.pop("PCFA_USE_CPU")
        importlib.reload(config_paths)
        importlib.reload(attack_FGSM)
    assert got["pairs"] == want["pairs"] == 2
    assert 0 < got["l2_delta-avg"] <= 2 * 0.00025 + 1e-9
    for k in ("aee_pred-tgt", "aee_predadv-tgt", "l2_delta-avg"):
        assert abs(got[k] - want[k]) <= 1e-3 * max(1.0, abs(want[k])), (k, got[k], want[k])


def test_universal_artifacts_round_trip_through_evaluate_on_gpu(tmp_path):
    """attack_l2_universal writes `NNNNN_delta1_e{E}.npy` (attack_PCFA.py:524,531); evaluate_PCFA reads the folder back
    (evaluate_PCFA.py:21-58), re-pads across network families (:60-79) and evaluates on the GPU (:86-299)."""
    import os
    from pcfa_amd import attack_PCFA, evaluate_PCFA
    size = "130x170"   # SpyNet pads it to 192x192 (div 64), RAFT to 136x176 (div 8): the re-padding path is exercised
    args = _f3_cli(universal_perturbation=True, no_save=False, output_folder=str(tmp_path), steps=1, epochs=2,
                   synthetic_size=size)
    res = attack_PCFA.attack_l2_universal(args)
    run_dir = None
    for root, dirs, files in os.walk(str(tmp_path)):
        if os.path.basename(root) == "patches" and any(f.endswith("_delta1_e1.npy") for f in files):
            run_dir = os.path.dirname(root)
    assert run_dir is not None
    epochs, d1, d2 = evaluate_PCFA.extract_epoch_patchlist(run_dir)
    assert epochs == 2 and len(d1) == 2 and len(d2) == 2
    assert np.array_equal(np.load(d1[-1]), res["delta1"].cpu().numpy())
    ev = evaluate_PCFA.eval_l2_universal(_f3_cli(universal_perturbation=True, perturbation_sourcefolder=run_dir,
                                                 origin_net="SpyNet", synthetic_size=size))
    assert len(ev) == 2 and ev[0]["images"] == 2 and np.isfinite(ev[1]["epoch_aee_pred-predadv"])
    # black-box transfer: the SpyNet perturbation (padded to 64) evaluated on RAFT (padded to 8) -- evaluate_PCFA.py:60-79
    ev2 = evaluate_PCFA.eval_l2_universal(_f3_cli(net="RAFT", universal_perturbation=True, synthetic_size=size,
                                                  perturbation_sourcefolder=run_dir, origin_net="SpyNet"))
    assert len(ev2) == 2 and np.isfinite(ev2[1]["epoch_aee_pred-predadv"])


# --------------------------------------------------------------------------- process-to-process reproducibility
@pytest.mark.parametrize("net,size,extra,seeds,steps", [
    ("PWCNet", "375x1242", ["--box", "clipping", "--joint"], "0,1,2,3", 20),
    ("RAFT", "436x1024", ["--box", "change_of_variables"], "0", 4)])
def test_fresh_processes_are_bit_identical(net, size, extra, seeds, steps):
    """two FRESH processes (children of this test, started one after the other; the parent of
    the children never touches the GPU itself) run the captured-graph attack on the same synthetic pairs and must
    agree bit for bit -- every operator output and gradient of one recorded eager closure, the loss of every closure
    evaluation, every per-step metric and the best-iterate results (tools/process_repro.py).  PWC-Net at BASELINE
    config 4's shape on pairs 0-3 (r03: 38.92 vs 40.41 between two processes while the library still ran the
    transposed convolutions and the bilinear up-sampling backward), RAFT at config 2's shape."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "process_repro.py"), "--net", net, "--size", size,
                        "--steps", str(steps), "--procs", "2", "--seeds", seeds] + extra,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode in (0, 1), p.stderr[-3000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["identical"], rec["first_difference"]
    assert all(pr["graphed"] for run in rec["per_process"] for pr in run)


def test_conv_workspace_never_frees_a_captured_buffer():
    """the split-K scratch of pcfa_conv3x3_run is baked into captured hipGraphs that outlive the pair
    (attack_PCFA._PairGraphs); when a larger shape makes it grow, the old buffer must stay allocated."""
    dev = torch.device(DEV)
    a = hip_ops._conv_workspace(dev, 1 << 20)
    pa = a.data_ptr()
    n = a.numel() * 4
    b = hip_ops._conv_workspace(dev, 4 * n)
    assert b.data_ptr() != pa and b.numel() * 4 >= 4 * n
    assert any(t.data_ptr() == pa for t in hip_ops._CONV_WS_RETIRED)
    assert hip_ops._conv_workspace(dev, n) is b                # a smaller request keeps the current buffer


def test_pwcnet_deferred_leaky_masks_change_no_bit():
    """PWC-Net closure (PWCNet.py:227-330) with the LeakyReLU backward of single-consumer layers applied in the
    consumer's data-gradient epilogue (pyramid chains conv_a -> conv_aa -> conv_b, context network dc_conv1..6:
    nets/pwcnet._chain; dense decoder blocks: Config.dense_block_fused_masks -- 40 elementwise launches less per
    closure) against one pcfa_leaky_relu_bwd launch per layer: the same products in the same order, so loss, flow and
    gradient must be bit-identical."""
    import dataclasses
    from pcfa_amd import config
    dev = torch.device(DEV)

    def run(conf):
        closure_util._MODELS.clear()
        r = closure_util.run_closure("PWCNet", 192, 256, "clipping", True, "zero", "aee", seed=5, device=dev, config=conf)
        closure_util._MODELS.clear()
        return r["loss"], r["flow"].clone(), r["grads"][0].clone()

    fused = run(config.DEFAULT)
    plain = run(dataclasses.replace(config.DEFAULT, defer_leaky=False, dense_block_fused_masks=False))
    assert fused[0] == plain[0] and torch.equal(fused[1], plain[1]) and torch.equal(fused[2], plain[2])
    assert float(fused[2].abs().max()) > 0
    # Config.pwc_fold_glue (r04): `up_flow * s` inside the warp, decoder inputs handed to the dense block as parts, ONE
    # re-gridding copy between dilated layers move no bit either; RGB -> BGR folded into conv1a's weights sums that
    # layer's three input channels in another order (the stride-2 kernel's K axis), so against the reference's explicit
    # stack the closure agrees to rounding only
    unfolded = run(dataclasses.replace(config.DEFAULT, pwc_fold_glue=False))
    assert abs(fused[0] - unfolded[0]) <= 1e-6 * abs(unfolded[0])
    assert max_abs(fused[1], unfolded[1]) <= 2e-5 * float(unfolded[1].abs().max())
    assert rel_l2(fused[2], unfolded[2]) < 2e-3
