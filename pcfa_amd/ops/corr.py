"""RAFT / GMA all-pairs correlation pyramid and window lookup (models/raft/corr.py:12-60 == models/gma/corr.py:15-63),
also fused with convc1 + ReLU (update.py:79-93)."""
import ctypes
import os
import weakref

import torch

from .. import _hip
from . import core
from .core import _call, _dev, _note_work, _pair, _ptr, _ptr_off, _stream


# --------------------------------------------------------------------------- #
# RAFT / GMA correlation pyramid
# --------------------------------------------------------------------------- #
class _CorrState:
    """Device buffers shared by the build node and its lookup nodes."""
    __slots__ = ("B", "D", "H", "W", "L", "r", "slab", "pyr", "f2ext", "dpyr", "token_grad", "coords_bwd", "bwd_windows")


class _CorrBuild(torch.autograd.Function):
    """fmap1, fmap2 -> 1-element token; the pyramid itself lives in `state`.

    The token only carries the autograd dependency: every lookup consumes it, so
    this node's backward runs after ALL lookup backwards have accumulated into
    state.dpyr, and performs the two GEMMs of the volume's backward once.
    """

    @staticmethod
    def forward(ctx, fmap1, fmap2, state):
        lib = _hip.load()
        B, D, H, W = fmap1.shape
        f1 = fmap1.contiguous()
        f2 = fmap2.contiguous()
        slab = state.slab
        state.f2ext = torch.empty((B, D, slab), device=f1.device, dtype=torch.float32)
        state.pyr = torch.empty((B * H * W, slab), device=f1.device, dtype=torch.float32)
        _call("pcfa_corr_f2ext_fwd", _ptr(f2), _ptr(state.f2ext), B, D, H, W, state.L)
        _call("pcfa_corr_pyramid_fwd", _ptr(f1), _ptr(state.f2ext), _ptr(state.pyr), B, D, H, W,
                                             state.L)
        ctx.state = state
        ctx.save_for_backward(f1)
        return torch.zeros(1, device=f1.device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, grad_token):
        st = ctx.state
        (f1,) = ctx.saved_tensors
        if st.dpyr is None:  # no lookup contributed a gradient
            z = torch.zeros_like(f1)
            st.token_grad = None
            return z, z.clone(), None
        lib = _hip.load()
        B, D, H, W = st.B, st.D, st.H, st.W
        df1 = torch.empty_like(f1)
        df2 = torch.empty_like(f1)
        # the coordinates of every lookup that accumulated into dpyr: the products skip what no window touched
        # (the per-block segment record of corr_window_segments_kernel holds four levels: more levels -> dense products)
        cs = st.coords_bwd if (st.coords_bwd and len(st.coords_bwd) <= 32 and st.bwd_windows and st.L <= 4) else []
        if cs:
            nbytes = lib.pcfa_corr_pyramid_bwd_windows_workspace_bytes(B, D, H, W, st.L)
            ws = torch.empty((nbytes + 3) // 4, device=f1.device, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * len(cs))(*[c.data_ptr() for c in cs])
            _call("pcfa_corr_pyramid_bwd_windows", _ptr(st.dpyr), _ptr(f1), _ptr(st.f2ext), _ptr(df1), _ptr(df2), _ptr(ws),
                  ctypes.c_size_t(nbytes), ptrs, len(cs), st.r, B, D, H, W, st.L)
            if core.work_recorder() is not None:
                # executed matrix work of the two sparse products: the K segments the kernels walked, read back from the
                # workspace (csrc/corr_pyramid.hip: per 128-wide column block {count, (begin, end) x 4, pad} ints behind
                # the split-K area; segA = blocks of dfmap1's Q columns, segB = blocks of df2ext's slab columns)
                base = (int(lib.pcfa_corr_pyramid_bwd_workspace_bytes(B, D, H, W, st.L)) + 15) & ~15
                nbA, nbB = -(-(H * W) // 128), -(-st.slab // 128)
                seg = ws.view(torch.int32)[base // 4: base // 4 + 10 * B * (nbA + nbB)].cpu().view(-1, 10).long()
                live = torch.arange(4)[None, :] < seg[:, :1]
                k = ((seg[:, 2:9:2] - seg[:, 1:8:2]) * live).sum(1)
                dense = 2.0 * B * D * (H * W) ** 2
                _note_work("corr_pyramid_gemm_dfmap1", dense, 2.0 * D * 128 * float(k[:B * nbA].sum()))
                _note_work("corr_pyramid_gemm_df2ext", dense, 2.0 * D * 128 * float(k[B * nbA:].sum()))
        else:   # the dense products under their own entry point (and their own launch indices in DispatchTimer's plan)
            nbytes = lib.pcfa_corr_pyramid_bwd_workspace_bytes(B, D, H, W, st.L)
            ws = torch.empty((nbytes + 3) // 4, device=f1.device, dtype=torch.float32)
            _call("pcfa_corr_pyramid_bwd", _ptr(st.dpyr), _ptr(f1), _ptr(st.f2ext), _ptr(df1), _ptr(df2), _ptr(ws),
                  ctypes.c_size_t(nbytes), B, D, H, W, st.L)
        st.dpyr = None
        st.token_grad = None
        st.coords_bwd = None
        return df1, df2, None


def _token_grad(st, device):
    """The 1-element token only orders the build node behind every lookup node: ONE lookup per backward pass hands it a
    (zero) gradient, the others return None -- twelve zeros(1) fills and eleven 1-element accumulations per closure
    otherwise (each a kernel launch)."""
    if st.token_grad is None:
        st.token_grad = torch.zeros(1, device=device, dtype=torch.float32)
        return st.token_grad
    return None


class _CorrLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, coords, state):
        if coords.requires_grad and torch.is_grad_enabled():
            # models/raft/corr.py's bilinear_sampler differentiates w.r.t. coords; RAFT / GMA detach them
            # (raft.py:122-123) and this operator does not implement that gradient: refuse instead of returning zeros
            raise RuntimeError("CorrBlock lookup: coords.requires_grad is not supported (detach the coordinates, as "
                               "models/raft/raft.py:122-123 does)")
        lib = _hip.load()
        st = state
        c = coords.contiguous()
        n1 = 2 * st.r + 1
        out = torch.empty((st.B, st.L * n1 * n1, st.H, st.W), device=c.device, dtype=torch.float32)
        _call("pcfa_corr_lookup_fwd", _ptr(st.pyr), _ptr(c), _ptr(out), st.B, st.H, st.W, st.L, st.r)
        ctx.state = st
        ctx.save_for_backward(c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        st = ctx.state
        (c,) = ctx.saved_tensors
        lib = _hip.load()
        if st.dpyr is None:
            st.dpyr = torch.zeros_like(st.pyr)
            st.coords_bwd = []
        st.coords_bwd.append(c)
        g = grad_out.contiguous()
        _call("pcfa_corr_lookup_bwd", _ptr(st.dpyr), _ptr(c), _ptr(g), st.B, st.H, st.W, st.L, st.r)
        return _token_grad(st, g.device), None, None


_convc1_packs = {}


def _convc1_packed(weight):
    """pcfa_lookup_convc1_pack_weights of a frozen [256, 324, 1, 1] weight (both operand orders), cached per version."""
    key = id(weight)
    hit = _convc1_packs.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        lib = _hip.load()
        cout = weight.shape[0]
        w = weight.detach().reshape(cout, -1).contiguous()
        packed = torch.empty(int(lib.pcfa_lookup_convc1_packed_floats(cout)), device=w.device, dtype=torch.float32)
        _call("pcfa_lookup_convc1_pack_weights", _ptr(w), _ptr(packed), cout, w.shape[1])
        hit = (weakref.ref(weight, lambda _r, k=key: _convc1_packs.pop(k, None)), weight._version, packed)
        _convc1_packs[key] = hit
    return hit[2]


class _CorrLookupConv(torch.autograd.Function):
    """relu(convc1(lookup(coords))) in one launch per direction (pcfa_lookup_convc1_fwd / _bwd): the lookup node of
    _CorrLookup with the motion encoder's 1x1 convolution (frozen weight) folded in.  Backward accumulates into the
    shared state.dpyr exactly like _CorrLookup."""

    @staticmethod
    def forward(ctx, token, coords, state, weight, bias, relu):
        if coords.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("CorrBlock lookup: coords.requires_grad is not supported (detach the coordinates, as "
                               "models/raft/raft.py:122-123 does)")
        st = state
        c = coords.contiguous()
        packed = _convc1_packed(weight)
        out = torch.empty((st.B, weight.shape[0], st.H, st.W), device=c.device, dtype=torch.float32)
        _call("pcfa_lookup_convc1_fwd", _ptr(st.pyr), _ptr(c), _ptr(packed), _ptr(bias), _ptr(out), st.B, st.H, st.W,
              st.L, st.r, weight.shape[0], int(relu))
        ctx.state, ctx.packed, ctx.relu, ctx.cout = st, packed, int(relu), weight.shape[0]
        ctx.save_for_backward(c, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            raise RuntimeError("lookup_conv is the frozen-weight path: no weight / bias gradient")
        st = ctx.state
        c, out = ctx.saved_tensors
        if st.dpyr is None:
            st.dpyr = torch.zeros_like(st.pyr)
            st.coords_bwd = []
        st.coords_bwd.append(c)
        g = grad_out.contiguous()
        _call("pcfa_lookup_convc1_bwd", _ptr(st.dpyr), _ptr(c), _ptr(ctx.packed), _ptr(out), _ptr(g), st.B, st.H, st.W,
              st.L, st.r, ctx.cout, ctx.relu)
        return _token_grad(st, g.device), None, None, None, None, None


class CorrBlock:
    """Drop-in for models/raft/corr.py:12-50 -- same constructor and __call__."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, bwd_windows=True):
        """bwd_windows (Config.pyramid_bwd_windows): the backward products skip what no lookup window touched; False =
        the dense products (A/B, parity tests)."""
        _dev(fmap1, fmap2)
        if fmap1.shape != fmap2.shape or fmap1.dim() != 4:
            raise ValueError("CorrBlock expects two [B,D,H,W] feature maps of equal shape")
        lib = _hip.load()
        self.num_levels = num_levels
        self.radius = radius
        st = _CorrState()
        st.B, st.D, st.H, st.W = fmap1.shape
        st.L, st.r = num_levels, radius
        st.slab = lib.pcfa_corr_slab_floats(st.H, st.W, num_levels)
        if st.slab <= 0 or (st.H >> (num_levels - 1)) < 1 or (st.W >> (num_levels - 1)) < 1:
            raise ValueError("feature map %dx%d too small for %d pyramid levels" % (st.H, st.W, num_levels))
        st.dpyr = None
        st.token_grad = None
        st.coords_bwd = None
        st.bwd_windows = bool(bwd_windows)
        self._state = st
        self._token = _CorrBuild.apply(fmap1, fmap2, st)

    def __call__(self, coords):
        _dev(coords)
        return _CorrLookup.apply(self._token, coords, self._state)

    def lookup_conv_relu(self, coords, weight, bias, relu=True):
        """relu(conv1x1(self(coords), weight, bias)) without materialising the lookup (update.py:79-93 convc1);
        None when the shape is not the fused kernel's (4 levels, radius 4, 256 x 324 weight, bias present)."""
        if (self.num_levels != 4 or self.radius != 4 or bias is None or weight.dim() != 4
                or tuple(weight.shape) != (256, 324, 1, 1) or weight.requires_grad or bias.requires_grad):
            return None
        _dev(coords, weight, bias)
        return _CorrLookupConv.apply(self._token, coords, self._state, weight, bias, relu)

    @property
    def corr_pyramid(self):
        """Per-level tensors [B*Q,1,H_l,W_l] gathered out of the tiled slab matrix (for inspection/tests)."""
        st = self._state
        return [st.pyr[:, idx.to(st.pyr.device)].reshape(-1, 1, h, w)
                for (idx, h, w) in tiled_index_maps(st.H, st.W, st.L)]


def tiled_index_maps(H, W, num_levels):
    """[(index tensor [H_l*W_l] into a query slab, H_l, W_l)] -- the 4x4-tile layout of include/pcfa_hip.h."""
    out, off, h, w = [], 0, H, W
    for _ in range(num_levels):
        tw = (w + 3) // 4
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        idx = off + ((ys // 4) * tw + xs // 4) * 16 + (ys % 4) * 4 + xs % 4
        out.append((idx.reshape(-1), h, w))
        off += ((h + 3) // 4) * tw * 16
        h, w = h // 2, w // 2
    return out
