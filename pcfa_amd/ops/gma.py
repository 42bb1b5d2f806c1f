"""GMA attention (models/gma/gma.py:34-77,79-115): row softmax in one pass, attention products,
the shared attention gradient; pcfa_gemm_f32."""
import ctypes
import os
import weakref

import torch

from .. import _hip
from . import core
from .core import _call, _dev, _note_work, _pair, _ptr, _ptr_off, _stream


# --------------------------------------------------------------------------- #
# GMA attention (models/gma/gma.py:34-77,79-115)
# --------------------------------------------------------------------------- #
def gemm_f32(a, b, a_kmajor, b_kmajor, alpha=1.0, splits=1, out=None):
    """C[..., m, n] = alpha * sum_k A(m, k) B(k, n) on the fp32 matrix cores (pcfa_gemm_f32).  `a` is [.., M, K]
    (a_kmajor = 0) or [.., K, M] (1); `b` is [.., N, K] (b_kmajor = 0) or [.., K, N] (1); leading dims = batch."""
    _dev(a, b)
    a, b = a.contiguous(), b.contiguous()
    M, K = (a.shape[-1], a.shape[-2]) if a_kmajor else (a.shape[-2], a.shape[-1])
    N = b.shape[-1] if b_kmajor else b.shape[-2]
    if (b.shape[-2] if b_kmajor else b.shape[-1]) != K or a.shape[:-2] != b.shape[:-2]:
        raise ValueError("gemm_f32: operand shapes %s / %s do not match" % (tuple(a.shape), tuple(b.shape)))
    batch = 1
    for d in a.shape[:-2]:
        batch *= d
    if out is None:
        out = torch.empty(a.shape[:-2] + (M, N), device=a.device, dtype=torch.float32)
    lib = _hip.load()
    ws, nbytes = None, 0
    if splits > 1:
        nbytes = int(lib.pcfa_gemm_f32_workspace_bytes(M, N, batch, splits))
        ws = torch.empty(nbytes // 4, device=a.device, dtype=torch.float32)
    _call("pcfa_gemm_f32", _ptr(a), _ptr(b), _ptr(out), M, N, K, a.shape[-1], b.shape[-1], N, int(a_kmajor),
          int(b_kmajor), batch, M * K, N * K, M * N, float(alpha), int(splits), _ptr(ws), ctypes.c_size_t(nbytes))
    return out


def _attn_mm(a, b, a_kmajor, b_kmajor, alpha=1.0, splits=1, gemm="lib"):
    """A plain GEMM of the attention block.  gemm = "lib" (Config.gma_gemm's default): the library (rocBLAS through
    torch.matmul) -- these are plain dense products and it runs them at 107-126 TFLOP/s; "hip" routes them through
    pcfa_gemm_f32 (80-105 TFLOP/s, tools/bench_gemm.py), which the parity test exercises either way."""
    if gemm == "hip":
        return gemm_f32(a, b, a_kmajor, b_kmajor, alpha=alpha, splits=splits)
    at = a.transpose(-1, -2) if a_kmajor else a
    bt = b if b_kmajor else b.transpose(-1, -2)
    out = torch.matmul(at, bt)
    return out if alpha == 1.0 else out.mul_(alpha)


class _AttentionSoftmax(torch.autograd.Function):
    """attn = softmax(scale * q k^T) (gma.py:52-74, content-only branch): the similarity product (plain GEMM), then the
    row softmax as ONE read and ONE write of the [N, N] matrix, in place (pcfa_softmax_rows_fwd: a 28 KB row lives in
    the registers of one workgroup; the library makes three passes), and the same in the backward: d sim = attn * (g -
    rowsum(g * attn)) in one pass, dq = scale * dsim k, dk = scale * dsim^T q."""

    @staticmethod
    def forward(ctx, q, k, scale, gemm="lib"):
        _dev(q, k)
        ctx.gemm = gemm
        q, k = q.contiguous(), k.contiguous()
        sim = _attn_mm(q, k, 0, 0, alpha=scale, gemm=gemm)            # [.., N, N]
        n = sim.shape[-1]
        _call("pcfa_softmax_rows_fwd", _ptr(sim), _ptr(sim), sim.numel() // n, n)
        ctx.scale = float(scale)
        ctx.save_for_backward(q, k, sim)
        return sim

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        q, k, attn = ctx.saved_tensors
        g = g.contiguous()
        n = attn.shape[-1]
        # never in place on `g`: autograd forbids mutating a gradient it hands in (a hook, retain_grad() on the attention
        # matrix or a second consumer would see the overwritten values).  Same traffic either way (one read of attn and
        # g, one write); the price is a 198 MB temporary at 55x128.
        ds = torch.empty_like(g)
        _call("pcfa_softmax_rows_bwd", _ptr(attn), _ptr(g), _ptr(ds), attn.numel() // n, n)
        dq = _attn_mm(ds, k, 0, 1, alpha=ctx.scale, splits=8, gemm=ctx.gemm) if ctx.needs_input_grad[0] else None  # dsim k
        dk = _attn_mm(ds, q, 1, 1, alpha=ctx.scale, splits=8, gemm=ctx.gemm) if ctx.needs_input_grad[1] else None  # dsim^T q
        return dq, dk, None, None


def attention_softmax(q, k, scale, gemm="lib"):
    """softmax(scale * q k^T, dim=-1) for q, k [.., N, d]; gemm: "lib" | "hip" (Config.gma_gemm)."""
    return _AttentionSoftmax.apply(q, k, scale, gemm)


class AttnGradShare:
    """One attention matrix multiplied by a different value tensor in every refinement iteration (gma.py:79-115 called
    from update.py:128-130): its gradient is sum_i g_i v_i^T.  The nodes park (g_i, v_i); whichever runs last forms
    ONE product [g_1 | .. | g_n] [v_1 | .. | v_n]^T (K = n * 128) instead of n read-modify-write products over the
    198 MB matrix."""

    def __init__(self, gemm="lib"):
        self.pending = 0
        self.gs, self.vs = [], []
        self.gemm = gemm   # "lib" | "hip" (Config.gma_gemm): which GEMM the nodes sharing this object run


class _AttnTimesValue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn, v, shared):
        _dev(attn, v)
        v = v.contiguous()
        ctx.save_for_backward(attn, v)
        ctx.shared = shared
        shared.pending += 1
        return _attn_mm(attn, v, 0, 1, splits=8, gemm=shared.gemm)     # [.., N, d]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        attn, v = ctx.saved_tensors
        sh = ctx.shared
        g = g.contiguous()
        dv = _attn_mm(attn, g, 1, 1, splits=8, gemm=sh.gemm) if ctx.needs_input_grad[1] else None     # attn^T g
        d_attn = None
        if ctx.needs_input_grad[0]:
            if sh.pending <= 0:
                raise RuntimeError("GMA attention gradient: backward re-entered after the shared buffers were released; "
                                   "run a fresh forward (retain_graph is not supported on this path)")
            sh.gs.append(g)
            sh.vs.append(v)
            sh.pending -= 1
            if sh.pending == 0:
                gcat, vcat = torch.cat(sh.gs, dim=-1), torch.cat(sh.vs, dim=-1)
                sh.gs, sh.vs = [], []
                d_attn = _attn_mm(gcat, vcat, 0, 0, gemm=sh.gemm)      # [.., N, N], K = n * d
        return d_attn, dv, None


def attn_times_value(attn, v, shared):
    return _AttnTimesValue.apply(attn, v, shared)
