"""ORACLE -- test infrastructure, not product code.

CPU restatement (PyTorch fp32 / plain C) of the operators on the PCFA hot path,
exposing the same interface as :mod:`pcfa_amd.hip_ops` so that tests can (a)
compare the HIP kernels with it on identical inputs and (b) drive the host-side
attack logic on machines without a GPU.  Pinned against golden vectors produced
by the real reference (tests/golden/make_golden.py -> tests/golden/*.npz) and,
for the cost volume, against the reference's own C++ build (oracle/_ref).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package (plus the parity checkers the tests invoke as child
processes: tools/schedule_parity.py, tools/parity_matrix.py,
tools/trajectory_closure_parity.py -- always as the thing compared AGAINST);
nothing under pcfa_amd/ does.

Every function cites the reference lines it follows (paths relative to the
reference root).
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as F

# the optimiser of the reference attack loop (attack_PCFA.py:97,114,382,388) is torch's own
LBFGS = torch.optim.LBFGS

_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# RAFT / GMA correlation volume + lookup
# --------------------------------------------------------------------------- #
def _f32_unless_f64(t):
    """dtype of the reference's `.float()` casts: float32 -- except when the parity ARBITER (tools/parity_arbiter.py,
    tools/trajectory_closure_parity.py) evaluates this port in float64 to judge which fp32 leg is closer to exact."""
    return torch.float64 if t.dtype == torch.float64 else torch.float32


def corr_volume(fmap1, fmap2):
    """models/raft/corr.py:52-60 -- all-pairs dot products scaled by 1/sqrt(dim)."""
    b, d, h, w = fmap1.shape
    a = fmap1.reshape(b, d, h * w).transpose(1, 2)
    vol = torch.matmul(a, fmap2.reshape(b, d, h * w))
    vol = vol.reshape(b, h, w, 1, h, w)
    return vol / torch.sqrt(torch.tensor(d).to(_f32_unless_f64(vol)))


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """models/raft/corr.py:13-27 -- level 0 + (num_levels-1) 2x2 average poolings."""
    vol = corr_volume(fmap1, fmap2)
    b, h1, w1, one, h2, w2 = vol.shape
    level = vol.reshape(b * h1 * w1, one, h2, w2)
    pyramid = [level]
    for _ in range(num_levels - 1):
        level = F.avg_pool2d(level, 2, stride=2)
        pyramid.append(level)
    return pyramid


def _sample_pixels(img, xy):
    """models/raft/utils/utils.py:57-71 (bilinear_sampler): pixel coords -> grid_sample."""
    hh, ww = img.shape[-2:]
    x, y = xy.split([1, 1], dim=-1)
    x = 2 * x / (ww - 1) - 1
    y = 2 * y / (hh - 1) - 1
    return F.grid_sample(img, torch.cat([x, y], dim=-1), align_corners=True)


def corr_lookup(pyramid, coords, radius=4):
    """models/raft/corr.py:29-50 -- window lookup on every level.

    Channel l*(2r+1)^2 + a*(2r+1) + b samples level l at
    (x = cx/2^l + (a-r), y = cy/2^l + (b-r)): the reference stacks
    meshgrid(dy, dx) onto (x, y), which makes the window x-major.
    """
    r = radius
    n1 = 2 * r + 1
    pts = coords.permute(0, 2, 3, 1)
    b, h1, w1, _ = pts.shape
    offs = torch.linspace(-r, r, n1, dtype=_f32_unless_f64(pts))
    first, second = torch.meshgrid(offs, offs, indexing="ij")
    window = torch.stack([first, second], dim=-1).to(pts.device).view(1, n1, n1, 2)
    outs = []
    for lvl, vol in enumerate(pyramid):
        centre = pts.reshape(b * h1 * w1, 1, 1, 2) / 2 ** lvl
        taps = _sample_pixels(vol, centre + window)
        outs.append(taps.view(b, h1, w1, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().to(_f32_unless_f64(coords))


class CorrBlock:
    """models/raft/corr.py:12-50."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, bwd_windows=True):   # bwd_windows: a scheduling hint of the product
        self.num_levels = num_levels
        self.radius = radius
        self.corr_pyramid = corr_pyramid(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        return corr_lookup(self.corr_pyramid, coords, self.radius)


def corr_lookup_loops(pyramid, coords, radius=4):
    """First-principles double loop (numpy-speed: small cases only) of the same lookup,
    written from the bilinear / zero-padding definition rather than grid_sample."""
    import numpy as np
    r, n1 = radius, 2 * radius + 1
    c = coords.detach().numpy()
    b, _, h1, w1 = c.shape
    out = np.zeros((b, len(pyramid) * n1 * n1, h1, w1), dtype=np.float64)
    for lvl, vol_t in enumerate(pyramid):
        vol = vol_t.detach().numpy().astype(np.float64)
        hh, ww = vol.shape[-2:]
        for bi in range(b):
            for y in range(h1):
                for x in range(w1):
                    q = (bi * h1 + y) * w1 + x
                    cx, cy = c[bi, 0, y, x] / 2 ** lvl, c[bi, 1, y, x] / 2 ** lvl
                    for a in range(n1):
                        for bb in range(n1):
                            sx, sy = cx + (a - r), cy + (bb - r)
                            x0, y0 = math.floor(sx), math.floor(sy)
                            fx, fy = sx - x0, sy - y0
                            acc = 0.0
                            for (yy, xx, wgt) in ((y0, x0, (1 - fx) * (1 - fy)), (y0, x0 + 1, fx * (1 - fy)),
                                                  (y0 + 1, x0, (1 - fx) * fy), (y0 + 1, x0 + 1, fx * fy)):
                                if 0 <= yy < hh and 0 <= xx < ww:
                                    acc += wgt * vol[q, 0, yy, xx]
                            out[bi, lvl * n1 * n1 + a * n1 + bb, y, x] = acc
    return torch.from_numpy(out).float()


# --------------------------------------------------------------------------- #
# PWC-Net cost volume (plain-C restatement in spatial_corr.c)
# --------------------------------------------------------------------------- #
_clib = None


def _c():
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _clib = ctypes.CDLL(path)
    return _clib


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _fp(t):
    return ctypes.c_void_p(t.data_ptr())


class _SpatialCorr(torch.autograd.Function):
    """spatial_correlation_sampler.py:45-91 on top of oracle/spatial_corr.c."""

    @staticmethod
    def forward(ctx, input1, input2, kernel_size, patch_size, stride, padding, dilation, dilation_patch):
        kH, kW = _pair(kernel_size)
        pH, pW = _pair(patch_size)
        padH, padW = _pair(padding)
        dilH, dilW = _pair(dilation)
        dpH, dpW = _pair(dilation_patch)
        dH, dW = _pair(stride)
        a, b = input1.contiguous().float(), input2.contiguous().float()
        B, C, iH, iW = a.shape
        oH, oW = ctypes.c_int(), ctypes.c_int()
        _c().oracle_scorr_out_size(iH, iW, kH, kW, padH, padW, dilH, dilW, dH, dW, ctypes.byref(oH),
                                   ctypes.byref(oW))
        out = torch.empty((B, pH, pW, oH.value, oW.value), dtype=torch.float32)
        ctx.params = (B, C, iH, iW, kH, kW, pH, pW, padH, padW, dilH, dilW, dpH, dpW, dH, dW)
        _c().oracle_scorr_forward(_fp(a), _fp(b), _fp(out), *ctx.params)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        a, b = ctx.saved_tensors
        g = grad_output.contiguous().float()
        g1, g2 = torch.empty_like(a), torch.empty_like(b)
        _c().oracle_scorr_backward(_fp(a), _fp(b), _fp(g), _fp(g1), _fp(g2), *ctx.params)
        return g1, g2, None, None, None, None, None, None


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                               dilation_patch=1):
    """spatial_correlation_sampler.py:9-42."""
    return _SpatialCorr.apply(input1, input2, kernel_size, patch_size, stride, padding, dilation, dilation_patch)


def fanout(x, n):
    """n uses of x (the product sums their gradients with one kernel; autograd does it here)."""
    return tuple(x for _ in range(n))


def pwc_cost_volume(input1, input2, slope=0.1):
    """leakyRELU(correlate(input1, input2)): models/PWCNet/PWCNet.py:45-58 followed by :249,264,278,292,308."""
    if input1.dtype == torch.float64:   # the fp64 arbiter of tools/trajectory_closure_parity.py (the C kernel is fp32)
        out = spatial_correlation_shift_sum(input1, input2, 9)
    else:
        out = spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=9, stride=1)
    b, ph, pw, h, w = out.size()
    return F.leaky_relu(out.view(b, ph * pw, h, w) / input1.size(1), slope)


def spatial_correlation_shift_sum(input1, input2, patch_size=9):
    """Independent formulation of the k=1 / stride-1 cost volume (shift, multiply, sum over C)."""
    B, C, H, W = input1.shape
    r = (patch_size - 1) // 2
    padded = F.pad(input2, (r, r, r, r))
    out = torch.empty((B, patch_size, patch_size, H, W), dtype=input1.dtype)
    for i in range(patch_size):
        for j in range(patch_size):
            out[:, i, j] = (input1 * padded[:, :, i:i + H, j:j + W]).sum(1)
    return out


# --------------------------------------------------------------------------- #
# FlowNet2's native operators.  The reference has them as CUDA extensions only (no CPU build, nothing to run
# here): the functions below restate the CUDA kernels operation by operation.  PARITY UNPINNED for Resample2d and
# ChannelNorm (no reference output exists on this machine); the correlation is cross-pinned against the
# reference's own C++ sampler (patch 21, dilation_patch 2, k = 1), tests/test_oracle_cpu.py.
# --------------------------------------------------------------------------- #
def _shift2d(t, dy, dx):
    """out[..., y, x] = t[..., y + dy, x + dx], zero outside."""
    H, W = t.shape[-2:]
    out = torch.zeros_like(t)
    ys, ye = max(0, -dy), min(H, H - dy)
    xs, xe = max(0, -dx), min(W, W - dx)
    if ys < ye and xs < xe:
        out[..., ys:ye, xs:xe] = t[..., ys + dy:ye + dy, xs + dx:xe + dx]
    return out


def _fcorr_geometry(H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_cuda.cc:25-35."""
    kr = (kernel_size - 1) // 2
    border = kr + max_displacement
    oH = math.ceil(float(H + 2 * pad_size - 2 * border) / float(stride1))
    oW = math.ceil(float(W + 2 * pad_size - 2 * border) / float(stride1))
    drad = max_displacement // stride2
    return kr, oH, oW, drad, 2 * drad + 1


def flownet_corr_forward(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_cuda_kernel.cu:74-147: products over the zero-padded inputs, divided by k*k*C."""
    B, C, H, W = input1.shape
    kr, oH, oW, drad, D = _fcorr_geometry(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    m = pad_size + kr + drad * stride2  # generous zero margin; index = padded index + (m - pad_size)
    P1 = F.pad(input1, (m, m, m, m))
    P2 = F.pad(input2, (m, m, m, m))
    off = m - pad_size
    out = torch.zeros((B, D * D, oH, oW), dtype=input1.dtype)
    ys = max_displacement + off
    for tj in range(-drad, drad + 1):
        for ti in range(-drad, drad + 1):
            acc = torch.zeros((B, oH, oW), dtype=input1.dtype)
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    a = P1[:, :, ys + j:ys + j + (oH - 1) * stride1 + 1:stride1,
                           ys + i:ys + i + (oW - 1) * stride1 + 1:stride1]
                    y2, x2 = ys + tj * stride2 + j, ys + ti * stride2 + i
                    b = P2[:, :, y2:y2 + (oH - 1) * stride1 + 1:stride1, x2:x2 + (oW - 1) * stride1 + 1:stride1]
                    acc = acc + (a * b).sum(1)
            out[:, (tj + drad) * D + (ti + drad)] = acc / float(kernel_size * kernel_size * C)
    return out


def flownet_corr_backward(input1, input2, grad_output, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_cuda_kernel.cu:150-241 (input1), :243-333 (input2); stride1 = 1."""
    assert stride1 == 1
    B, C, H, W = input1.shape
    kr, oH, oW, drad, D = _fcorr_geometry(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    nelems = float(kernel_size * kernel_size * C)
    off = pad_size - max_displacement
    # S[tc][y][x] = sum of grad_output[tc] over the window [y+off-kr, y+off+kr] x [x+off-kr, x+off+kr],
    # clipped to the output (the xmin/xmax/ymin/ymax logic of :173-193)
    Lp = max(0, kr - off)
    Rh = max(0, H - 1 + off + kr - (oH - 1))
    Rw = max(0, W - 1 + off + kr - (oW - 1))
    gp = F.pad(grad_output, (Lp, Rw, Lp, Rh))
    box = F.avg_pool2d(gp, kernel_size, stride=1) * float(kernel_size * kernel_size) if kernel_size > 1 else gp
    s0 = off - kr + Lp
    S = box[:, :, s0:s0 + H, s0:s0 + W]
    g1 = torch.zeros_like(input1)
    g2 = torch.zeros_like(input2)
    for tc in range(D * D):
        i2 = (tc % D - drad) * stride2
        j2 = (tc // D - drad) * stride2
        w = S[:, tc:tc + 1]
        g1 = g1 + w * _shift2d(input2, j2, i2)
        g2 = g2 + _shift2d(w * input1, -j2, -i2)
    return g1 / nelems, g2 / nelems


class _FlownetCorr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1, input2, *geom):
        ctx.save_for_backward(input1, input2)
        ctx.geom = geom
        return flownet_corr_forward(input1, input2, *geom)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        g1, g2 = flownet_corr_backward(*ctx.saved_tensors, grad_output, *ctx.geom)
        return g1, g2, None, None, None, None, None


def flownet_correlation(input1, input2, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2,
                        corr_multiply=1):
    """correlation_package/correlation.py:53-67 (corr_multiply is ignored by the reference kernels)."""
    return _FlownetCorr.apply(input1.contiguous(), input2.contiguous(), pad_size, kernel_size, max_displacement,
                              stride1, stride2)


def _rs_taps(xf, yf, h, w):
    """resample2d_kernel.cu:50-53: the four neighbour indices, each clamped on its own."""
    fx, fy = torch.floor(xf), torch.floor(yf)
    xL = fx.long().clamp(0, w - 1)
    xR = (fx + 1).long().clamp(0, w - 1)
    yT = fy.long().clamp(0, h - 1)
    yB = (fy + 1).long().clamp(0, h - 1)
    return xL, xR, yT, yB, xf - fx, yf - fy


def _rs_positions(flow):
    B, _, H, W = flow.shape
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    return xs + flow[:, 0], ys + flow[:, 1]


def _rs_gather(img, yy, xx):
    """img [B,C,iH,iW], yy/xx [B,H,W] long -> [B,C,H,W]."""
    B, C, iH, iW = img.shape
    lin = (yy * iW + xx).view(B, 1, -1).expand(B, C, -1)
    return img.reshape(B, C, -1).gather(2, lin).view(B, C, *yy.shape[1:])


def resample2d_forward(input1, flow, kernel_size=1, bilinear=True):
    """resample2d_kernel.cu:16-72 (kernel_size 1)."""
    assert kernel_size == 1
    B, C, iH, iW = input1.shape
    H, W = flow.shape[-2:]
    xf, yf = _rs_positions(flow)
    if not bilinear:
        xN = torch.floor(xf + 0.5).long().clamp(0, W - 1)
        yN = torch.floor(yf + 0.5).long().clamp(0, H - 1)
        return _rs_gather(input1, yN, xN)
    xL, xR, yT, yB, alpha, beta = _rs_taps(xf, yf, H, W)
    a, b = alpha.double().unsqueeze(1), beta.double().unsqueeze(1)
    img = input1.double()
    val = torch.zeros((B, C, H, W), dtype=torch.float32)
    val = val + ((1. - a) * (1. - b) * _rs_gather(img, yT, xL)).float()
    val = val + (a * (1. - b) * _rs_gather(img, yT, xR)).float()
    val = val + ((1. - a) * b * _rs_gather(img, yB, xL)).float()
    val = val + (a * b * _rs_gather(img, yB, xR)).float()
    return val


def resample2d_backward(input1, flow, grad_output):
    """resample2d_kernel.cu:75-123 (input1, scatter-add) and :125-201 (flow)."""
    B, C, iH, iW = input1.shape
    H, W = flow.shape[-2:]
    xf, yf = _rs_positions(flow)
    g = grad_output
    # input1: weights from truncation (xf - int(xf), :103-104), indices clamped against the input size
    xL, xR, yT, yB, _, _ = _rs_taps(xf, yf, iH, iW)
    a = (xf - torch.trunc(xf)).unsqueeze(1)
    b = (yf - torch.trunc(yf)).unsqueeze(1)
    g1 = torch.zeros((B, C, iH * iW), dtype=torch.float32)
    for yy, xx, wgt in ((yT, xL, (1 - a) * (1 - b)), (yT, xR, a * (1 - b)), (yB, xL, (1 - a) * b),
                        (yB, xR, a * b)):
        lin = (yy * iW + xx).view(B, 1, -1).expand(B, C, -1)
        g1.scatter_add_(2, lin, (wgt * g).reshape(B, C, -1))
    g1 = g1.view(B, C, iH, iW)
    # flow: indices clamped against the flow size, gamma = 1 - fraction (:159-197)
    xL, xR, yT, yB, alpha, beta = _rs_taps(xf, yf, H, W)
    iTL, iTR = _rs_gather(input1, yT, xL), _rs_gather(input1, yT, xR)
    iBL, iBR = _rs_gather(input1, yB, xL), _rs_gather(input1, yB, xR)
    gx_, gy_ = (1 - beta).unsqueeze(1), (1 - alpha).unsqueeze(1)  # gamma of channel 0 (dx) / channel 1 (dy)
    gdx = torch.zeros((B, H, W), dtype=torch.float32)
    gdy = torch.zeros((B, H, W), dtype=torch.float32)
    for c in range(C):
        gv = g[:, c]
        gdx = gdx + gx_[:, 0] * gv * iTR[:, c]
        gdx = gdx - gx_[:, 0] * gv * iTL[:, c]
        gdx = gdx + (1 - gx_[:, 0]) * gv * iBR[:, c]
        gdx = gdx - (1 - gx_[:, 0]) * gv * iBL[:, c]
        gdy = gdy + gy_[:, 0] * gv * iBL[:, c]
        gdy = gdy - gy_[:, 0] * gv * iTL[:, c]
        gdy = gdy + (1 - gy_[:, 0]) * gv * iBR[:, c]
        gdy = gdy - (1 - gy_[:, 0]) * gv * iTR[:, c]
    return g1, torch.stack((gdx, gdy), 1)


class _Resample2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1, flow, kernel_size, bilinear):
        ctx.save_for_backward(input1, flow)
        return resample2d_forward(input1, flow, kernel_size, bilinear)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        g1, g2 = resample2d_backward(*ctx.saved_tensors, grad_output)
        return g1, g2, None, None


def resample2d(input1, input2, kernel_size=1, bilinear=True):
    """resample2d_package/resample2d.py:45-56."""
    return _Resample2d.apply(input1.contiguous(), input2.contiguous(), kernel_size, bilinear)


def channelnorm_forward(input1):
    """channelnorm_kernel.cu:18-60: sqrt of the sum of squares over channels, accumulated in channel order."""
    r = torch.zeros_like(input1[:, 0])
    for c in range(input1.shape[1]):
        r = r + input1[:, c] * input1[:, c]
    return torch.sqrt(r).unsqueeze(1)


def channelnorm_backward(input1, out, grad_output):
    """channelnorm_kernel.cu:63-96: float product / (double(norm) + 1e-9)."""
    return ((grad_output * input1).double() / (out.double() + 1e-9)).float()


class _ChannelNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1):
        out = channelnorm_forward(input1)
        ctx.save_for_backward(input1, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        return channelnorm_backward(*ctx.saved_tensors, grad_output)


def channelnorm(input1, norm_deg=2):
    """channelnorm_package/channelnorm.py:38-45 (the kernels ignore norm_deg)."""
    return _ChannelNorm.apply(input1.contiguous())


# --------------------------------------------------------------------------- #
# SepConvGRU gate arithmetic
# --------------------------------------------------------------------------- #
def _cb(bias):
    return 0 if bias is None else bias.view(1, -1, 1, 1)


def _opt(t):
    return 0 if t is None else t


def gru_gates(zc, rc, h, bias_z=None, bias_r=None, add_z=None, add_r=None):
    """models/raft/update.py:47-49 / :54-56 -- z = sigmoid(convz(hx)), r = sigmoid(convr(hx)); returns (z, r*h).
    The convolution is linear in hx = [h | inp | motion]: zc / rc carry the part that changes per iteration,
    add_z / add_r the part of the constant context features, bias_z / bias_r the bias."""
    return torch.sigmoid(zc + _opt(add_z) + _cb(bias_z)), torch.sigmoid(rc + _opt(add_r) + _cb(bias_r)) * h


def gru_gates_packed(zr, h, bias_zr=None, add_zr=None):
    """gru_gates on the stacked pre-activations [zc | rc] of one convolution with weights [Wz; Wr]."""
    c = zr.shape[1] // 2
    bz = None if bias_zr is None else bias_zr[:c]
    br = None if bias_zr is None else bias_zr[c:]
    az = None if add_zr is None else add_zr[:, :c]
    ar = None if add_zr is None else add_zr[:, c:]
    return gru_gates(zr[:, :c], zr[:, c:], h, bz, br, az, ar)


def gru_update(z, qc, h, bias_q=None, add_q=None):
    """models/raft/update.py:49-50 / :57-58 -- q = tanh(convq(.)); h = (1-z) * h + z * q."""
    q = torch.tanh(qc + _opt(add_q) + _cb(bias_q))
    return (1 - z) * h + z * q


def sepconv5(a, b, weight):
    """The SepConvGRU gate convolution on hx = cat([a, b]) (models/raft/update.py:45-47,52-54) without bias."""
    x = a if b is None else torch.cat([a, b], dim=1)
    return F.conv2d(x, weight, None, padding=(weight.shape[2] // 2, weight.shape[3] // 2))


def conv3x3(x, weight, bias=None, relu=False, leaky_slope=None, skip=False, grad_premasked=False, mask_input_grad=False,
            input_slope=0.):
    """3x3 / stride 1 / pad 1 convolution (+ bias, + ReLU or LeakyReLU): models/raft/update.py:6-16,79-101,
    models/PWCNet/PWCNet.py:29-35.  skip: also return x (the product sums the residual path's gradient in a kernel).
    grad_premasked / mask_input_grad / input_slope: scheduling hints of the product (which kernel applies an
    activation's backward); they change no value."""
    y = F.conv2d(x, weight, bias, stride=1, padding=1)
    if leaky_slope is not None:
        y = F.leaky_relu(y, leaky_slope)
    elif relu:
        y = F.relu(y)
    return (y, x) if skip else y


def gru_step(h, rest, halves, rest_relu_channels=0):
    """SepConvGRU update (models/raft/update.py:45-60) from the hoisted context parts, composed from the oracle's own
    operators: halves = ((w_zr, p_zr, w_q, p_q), (w_zr, p_zr, w_q, p_q)).  rest_relu_channels: a scheduling hint of the
    product (where the ReLU backward of `rest` runs); autograd differentiates every ReLU in place here."""
    for w_zr, p_zr, w_q, p_q in halves:
        z, rh = gru_gates_packed(sepconv5(h, rest, w_zr), h, None, p_zr)
        h = gru_update(z, sepconv5(rh, rest, w_q), h, None, p_q)
    return h


FEWIN_SHAPES = {(2, 7), (1, 7), (2, 5), (2, 3)}


def conv_fewin(x, weight, bias=None, relu=False):
    """models/raft/update.py:79-101: F.relu(self.convf1(flow)) (any k x k "same" convolution)."""
    y = F.conv2d(x, weight, bias, stride=1, padding=weight.shape[-1] // 2)
    return F.relu(y) if relu else y


def conv_s2_supported(x, weight):
    N, Cin, kh, kw = weight.shape
    return kh == kw and ((Cin, kh) == (3, 7) or kh == 3) and x.shape[3] % 4 == 0


def conv_s2(x, weight, bias=None, relu=False, leaky_slope=None, grad_premasked=False, own_bwd=True):
    """models/raft/extractor.py:118 (stem) and :23-58 (first convolution of a stride-2 residual block):
    conv2d(x, w, b, stride=2, padding=k//2) followed by the activation the caller fuses."""
    y = F.conv2d(x, weight, bias, stride=2, padding=weight.shape[-1] // 2)
    if leaky_slope is not None:
        return F.leaky_relu(y, leaky_slope)
    return F.relu(y) if relu else y


def conv_s2_ds_supported(x, weight, weight_d):
    N, Cin, kh, kw = weight.shape
    return (kh, kw) == (3, 3) and tuple(weight_d.shape) == (N, Cin, 1, 1) and x.shape[3] % 8 == 0


def conv_s2_ds(x, weight, weight_d, bias=None, bias_d=None, relu=False):
    """models/raft/extractor.py:23-58 (stride 2): conv1 (3x3) and downsample[0] (1x1) of the block input."""
    y = F.conv2d(x, weight, bias, stride=2, padding=1)
    return (F.relu(y) if relu else y), F.conv2d(x, weight_d, bias_d, stride=2)


def conv3x3_cat(convs, tails=(), grad_premasked=False, mask_input_grads=False):
    """models/raft/update.py:91-101: torch.cat([relu(conv(x)) ...] + tails, dim=1).  The two flags are scheduling hints
    of the product (which kernel applies a ReLU mask); they change no value."""
    return torch.cat([F.relu(F.conv2d(x, w, b, stride=1, padding=1)) for x, w, b in convs] + list(tails), dim=1)


def dense_block(x, layers, slope=0.1, fused_masks=True):   # fused_masks: a scheduling hint of the product
    """models/PWCNet/PWCNet.py:234-323: x = cat((conv_i(x), x), 1) for the five decoder convolutions of a level."""
    if isinstance(x, (tuple, list)):   # the product takes the parts of PWCNet.py:265's concatenation separately
        x = torch.cat(tuple(x), 1)
    for w, b in layers:
        x = torch.cat((F.leaky_relu(F.conv2d(x, w, b, stride=1, padding=1), slope), x), 1)
    return x


def pwc_warp(x, flo, mask_threshold=0.0001, deterministic=True, flow_scale=1.0):   # deterministic: a scheduling hint
    """models/PWCNet/PWCNet.py:166-206, statement by statement; flow_scale: the caller's `up_flow * 0.625` (:262)."""
    if flow_scale != 1.0:
        flo = flo * flow_scale
    B, C, H, W = x.size()
    xx = torch.arange(0, W).view(1, -1).repeat(H, 1)
    yy = torch.arange(0, H).view(-1, 1).repeat(1, W)
    xx = xx.view(1, 1, H, W).repeat(B, 1, 1, 1)
    yy = yy.view(1, 1, H, W).repeat(B, 1, 1, 1)
    grid = torch.cat((xx, yy), 1).to(x.dtype)   # (.float() in the reference; fp64 only for the arbiter run)
    vgrid = grid + flo
    vx = 2.0 * vgrid[:, 0, :, :] / max(W - 1, 1) - 1.0
    vy = 2.0 * vgrid[:, 1, :, :] / max(H - 1, 1) - 1.0
    vgrid = torch.stack((vx, vy), dim=3)
    output = F.grid_sample(x, vgrid, align_corners=False)
    mask = F.grid_sample(torch.ones(x.size(), dtype=x.dtype), vgrid, align_corners=False)
    mask = (mask >= mask_threshold).to(x.dtype)
    return output * mask


def conv3x3_fewout(x, weight, bias=None, skip=False):
    """The flow-prediction convolutions (models/raft/update.py:6-14, PWCNet.py:37-38, FlowNet/submodules.py:33-34).
    skip: also return x (the product sums the other consumer's gradient of x in its data-gradient kernel)."""
    y = F.conv2d(x, weight, bias, stride=1, padding=1)
    return (y, x) if skip else y


def split_batch(x, b):
    """The product computes PWC-Net's two feature pyramids (PWCNet.py:233-244) in one batch; this hands the halves back."""
    return x[:b], x[b:]


def deconv4s2_fewout(x, weight, bias=None):
    """deconv() of PWC-Net (models/PWCNet/PWCNet.py:42-43): nn.ConvTranspose2d(in, out, 4, 2, 1)."""
    return F.conv_transpose2d(x, weight, bias, stride=2, padding=1)


def upsample_bilinear(x, factor, mul=1.0):
    """mul * nn.Upsample(scale_factor=factor, mode='bilinear')(x): `20 * self.upsample(flow2)` (PWCNet.py:73,321)."""
    return mul * F.interpolate(x, scale_factor=factor, mode='bilinear', align_corners=False)


def instance_norm_relu(x, eps=1e-5, relu=False):
    """models/raft/extractor.py:23-58: nn.InstanceNorm2d (no affine, batch statistics) then the optional ReLU."""
    y = F.instance_norm(x, eps=eps)
    return F.relu(y) if relu else y


def convex_upsample(flow, mask):
    """models/raft/raft.py:72-83 upsample_flow: [N,2,H,W] -> [N,2,8H,8W] by a softmax-weighted 3x3 combination."""
    N, _, H, W = flow.shape
    mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, 8 * H, 8 * W)


def flow_step(coords1, delta, coords0):
    """models/raft/raft.py:122-137: coords1 = coords1 + delta_flow; the flow is coords1 - coords0."""
    c = coords1 + delta
    return c, c - coords0


def add_relu(a, b, b_is_relu=False):
    """models/raft/extractor.py:50-58: relu(x + y).  b_is_relu: scheduling hint of the product, no effect here."""
    return F.relu(a + b)


def conv1x1(x, weight, bias=None):
    """A 1x1 / stride-1 convolution (models/raft/extractor.py:146,186; update.py:118-121): the library's."""
    return F.conv2d(x, weight, bias)


def bias_relu(x, bias=None):
    """F.relu(conv(x)) with the convolution's bias split off (models/raft/update.py:12-16,91-101)."""
    return torch.relu(x + _cb(bias))


# --------------------------------------------------------------------------- #
# attack math
# --------------------------------------------------------------------------- #
def pm1_pair(image1, image2):
    """models/raft/raft.py:88-89: image = 2 * (image / 255.0) - 1.0 for both images; (fnet batch, cnet input)."""
    n1 = 2 * (image1 / 255.0) - 1.0
    n2 = 2 * (image2 / 255.0) - 1.0
    return torch.cat([n1, n2], dim=0), n1


def box_transform(image, delta=None, change_of_variables=False, eps_box=0., scale=1.):
    """helper_functions/own_models.py:62-85 for one image tensor."""
    x = image
    if delta is not None:
        x = x + delta.repeat([image.size()[0], 1, 1, 1])
    if change_of_variables:
        x = (1. / 2.) * 1. / (1. - eps_box) * (torch.tanh(x) + (1 - eps_box))
    x = torch.clamp(x, 0., 1.)
    if scale != 1.:
        x = scale * x
    return x


def extract_deltas(nw_input1, nw_input2, image1, image2, boxconstraint, eps_box=0.):
    """attack_PCFA.py:20-29."""
    if boxconstraint in ['change_of_variables']:
        k = (1. / 2.) * 1. / (1. - eps_box)
        return (k * (torch.tanh(nw_input1) + (1. - eps_box)) - image1,
                k * (torch.tanh(nw_input2) + (1. - eps_box)) - image2)
    return torch.clamp(nw_input1, 0., 1.) - image1, torch.clamp(nw_input2, 0., 1.) - image2


def extract_deltas_joint(nw_delta, images_max, images_min):
    """attack_PCFA.py:32-37."""
    upper = torch.clamp(nw_delta + images_max, 0., 1.) - images_max
    delta = torch.clamp(upper + images_min, 0., 1.) - images_min
    return delta, delta


def avg_epe(flow1, flow2):
    """helper_functions/losses.py:3-30."""
    sq = (flow1 - flow2) ** 2
    if sq.dim() == 3:
        return torch.mean(torch.sum(sq, dim=0).sqrt())
    if sq.dim() == 4:
        return torch.mean(torch.sum(sq, dim=1).sqrt())
    raise ValueError("The flow tensors do not have a valid number of dimensions "
                     "(either [b,2,M,N] or [2,M,N]). Here: %s" % str(flow1.size()))


def avg_mse(flow1, flow2):
    """helper_functions/losses.py:32-44."""
    return torch.mean((flow1 - flow2) ** 2)


def f_epe(pred, target):
    """helper_functions/losses.py:47-58."""
    return avg_epe(pred, target)


def f_mse(pred, target):
    """helper_functions/losses.py:61-73."""
    return avg_mse(pred, target)


def f_cosim(pred, target):
    """helper_functions/losses.py:76-88 (operator precedence kept: the quotient is MULTIPLIED by |t|)."""
    return 1 - torch.sum(pred * target) / torch.sqrt(torch.sum(pred * pred)) * torch.sqrt(torch.sum(target * target))


def two_norm_avg(x):
    """helper_functions/losses.py:129-142."""
    return torch.sqrt(torch.sum(torch.pow(torch.flatten(x), 2))) / (torch.numel(x) ** 0.5)


def two_norm_avg_delta(delta1, delta2):
    """helper_functions/losses.py:91-107."""
    n = (torch.numel(delta1) + torch.numel(delta2)) ** 0.5
    return torch.sqrt(torch.sum(torch.pow(torch.flatten(delta1), 2)) +
                      torch.sum(torch.pow(torch.flatten(delta2), 2))) / n


def two_norm_avg_delta_squared(delta1, delta2):
    """helper_functions/losses.py:110-126."""
    n = torch.numel(delta1) + torch.numel(delta2)
    return (torch.sum(torch.pow(torch.flatten(delta1), 2)) + torch.sum(torch.pow(torch.flatten(delta2), 2))) / n


def relu_penalty(delta1, delta2, device=None, delta_bound=0.001):
    """losses.py:177-197."""
    zero = torch.tensor(0.)
    return torch.max(zero, two_norm_avg_delta_squared(delta1, delta2) - torch.tensor(delta_bound ** 2))


def get_loss(f_type, pred, target):
    """helper_functions/losses.py:145-174."""
    if f_type == "aee":
        return avg_epe(pred, target)
    if f_type == "cosim":
        return f_cosim(pred, target)
    if f_type == "mse":
        return avg_mse(pred, target)
    raise NotImplementedError(
        "The requested loss type %s does not exist. Please choose one of 'aee', 'mse' or 'cosim'" % f_type)


def loss_delta_constraint(pred, target, delta1, delta2, device=None, delta_bound=0.001, mu=100., f_type="aee",
                          batch_sums=None):
    """helper_functions/losses.py:200-230.

    batch_sums (multi-rank universal attack with `--loss cosim` only; None everywhere else): `f_cosim` is a ratio of
    sums over the WHOLE batch (losses.py:88), so a rank that holds a slice of the batch needs the global sums before
    its backward.  batch_sums(t) all-reduces the three local sums [p.t, p.p, t.t] in place (SUM) and returns the
    number of ranks; the value of the loss is then the single-process value on the global batch, and the gradient
    of the similarity term carries the factor `world` that the averaging all-reduce of the gradients removes."""
    if batch_sums is not None and f_type == "cosim":
        local = torch.stack((torch.sum(pred * target), torch.sum(pred * pred), torch.sum(target * target)))
        glob = local.detach().clone()
        world = batch_sums(glob)
        s_ = glob + world * (local - local.detach())       # value: global sums; d/d(local sums): world
        sim = 1 - s_[0] / torch.sqrt(s_[1]) * torch.sqrt(s_[2])
    else:
        sim = get_loss(f_type, pred, target)
    excess = two_norm_avg_delta_squared(delta1, delta2) - torch.tensor(delta_bound ** 2).to(pred.device)
    penalty = torch.max(torch.tensor(0.).to(pred.device), excess)
    return sim + mu * penalty
