"""GMA (Jiang et al. 2021) = RAFT + global motion aggregation, for the PCFA hot path.

Behavioural reference (cv-stuttgart/PCFA):
    models/gma/network.py:26-129     RAFTGMA wiring (6 iterations in PCFA, ownutilities.py:327)
    models/gma/gma.py:7-115          RelPosEmb, Attention, Aggregate
    models/gma/update.py:112-139     GMAUpdateBlock
    models/gma/corr.py:15-63         -> pcfa_amd.ops.get().CorrBlock (same HIP kernels as RAFT)

Parameter names follow the public gma-sintel.pth checkpoint.  The reference
config enables fp16 autocast on CUDA (models/_config/gma_config.json:5); the CPU
path the parity target is defined on runs fp32, so this implementation is fp32
on MI355X as well.
"""
import torch
import torch.nn as nn

from .. import ops
from ..config import cfg
from .raft import (BasicEncoder, BasicMotionEncoder, FlowHead, LookupRef, SepConvGRU, _mask_head, convex_upsample,
                   _f32, coords_grid, mask_logits)


class RelPosEmb(nn.Module):
    def __init__(self, max_pos_size, dim_head):
        super().__init__()
        self.rel_height = nn.Embedding(2 * max_pos_size - 1, dim_head)
        self.rel_width = nn.Embedding(2 * max_pos_size - 1, dim_head)
        deltas = torch.arange(max_pos_size).view(1, -1) - torch.arange(max_pos_size).view(-1, 1)
        self.register_buffer('rel_ind', deltas + max_pos_size - 1)

    def forward(self, q):
        # q: [b, heads, h, w, c] -> scores [b, heads, h, w, h, w]
        b, heads, h, w, c = q.shape
        height_emb = self.rel_height(self.rel_ind[:h, :h].reshape(-1)).view(h, h, 1, c)  # x u () d
        width_emb = self.rel_width(self.rel_ind[:w, :w].reshape(-1)).view(w, 1, w, c)    # y () v d
        height_score = torch.einsum('bhxyd,xuvd->bhxyuv', q, height_emb)
        width_score = torch.einsum('bhxyd,yuvd->bhxyuv', q, width_emb)
        return height_score + width_score


class Attention(nn.Module):
    def __init__(self, *, args, dim, max_pos_size=100, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_qk = nn.Conv2d(dim, heads * dim_head * 2, 1, bias=False)
        self.pos_emb = RelPosEmb(max_pos_size, dim_head)

    def forward(self, fmap):
        b, c, h, w = fmap.shape
        heads = self.heads
        q, k = self.to_qk(fmap).chunk(2, dim=1)
        # 'b (h d) x y -> b h (x y) d'
        q = q.reshape(b, heads, -1, h * w).transpose(-1, -2)
        k = k.reshape(b, heads, -1, h * w).transpose(-1, -2)
        o = ops.get()
        if (hasattr(o, "attention_softmax") and not getattr(self.args, "position_only", False)
                and not getattr(self.args, "position_and_content", False)):
            # similarity product on the fp32 matrix cores, row softmax with one read + one write of the [N, N] matrix
            # (pcfa_gemm_f32 + pcfa_softmax_rows_*); scale applied to the product instead of to q (gma.py:59)
            return o.attention_softmax(q.contiguous(), k.contiguous(), self.scale, gemm=cfg(self).gma_gemm)
        q = self.scale * q
        if getattr(self.args, "position_only", False):
            sim = self.pos_emb(q.reshape(b, heads, h, w, -1)).reshape(b, heads, h * w, h * w)
        elif getattr(self.args, "position_and_content", False):
            sim = torch.matmul(q, k.transpose(-1, -2))
            sim = sim + self.pos_emb(q.reshape(b, heads, h, w, -1)).reshape(b, heads, h * w, h * w)
        else:
            sim = torch.matmul(q, k.transpose(-1, -2))
        return sim.softmax(dim=-1)


class _SharedAttnGrad:
    """Book-keeping for one attention matrix that several _AttnTimesValue nodes multiply (once per refinement
    iteration): its gradient is accumulated in ONE buffer by the GEMMs themselves (beta = 1) and handed to autograd
    by whichever node runs last, instead of six [N, N] temporaries summed by five elementwise kernels of 3 x 198 MB
    traffic each (N = 7040 at 436x1024)."""

    def __init__(self):
        self.pending = 0
        self.buf = None


class _AttnTimesValue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn, v, shared):
        ctx.save_for_backward(attn, v)
        ctx.shared = shared
        shared.pending += 1
        return torch.matmul(attn, v)

    @staticmethod
    def backward(ctx, g):
        attn, v = ctx.saved_tensors
        sh = ctx.shared
        dv = torch.matmul(attn.transpose(-1, -2), g) if ctx.needs_input_grad[1] else None
        d_attn = None
        if ctx.needs_input_grad[0]:
            n, d = attn.shape[-1], v.shape[-1]
            if sh.buf is None:
                sh.buf = torch.matmul(g, v.transpose(-1, -2))
            else:
                sh.buf.view(-1, n, n).baddbmm_(g.reshape(-1, n, d), v.reshape(-1, n, d).transpose(-1, -2))
            if sh.pending <= 0:
                # a second backward through the same forward (retain_graph) would hand autograd a partial sum
                raise RuntimeError("GMA attention gradient: backward re-entered after the shared buffer was released; "
                                   "run a fresh forward (retain_graph is not supported on this path)")
            sh.pending -= 1
            if sh.pending == 0:
                d_attn, sh.buf = sh.buf, None
        return d_attn, dv, None


class Aggregate(nn.Module):
    def __init__(self, args, dim, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.scale = dim_head ** -0.5
        inner_dim = heads * dim_head
        self.to_v = nn.Conv2d(dim, inner_dim, 1, bias=False)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.project = nn.Conv2d(inner_dim, dim, 1, bias=False) if dim != inner_dim else None

    def forward(self, attn, fmap, shared=None):
        b, c, h, w = fmap.shape
        heads = self.heads
        v = self.to_v(fmap).reshape(b, heads, -1, h * w).transpose(-1, -2)  # b h (x y) d
        o = ops.get()
        if shared is not None and torch.is_grad_enabled() and attn.requires_grad:
            if hasattr(o, "AttnGradShare") and isinstance(shared, o.AttnGradShare):
                out = o.attn_times_value(attn, v, shared)                    # b h (x y) d, hand-written GEMM path
            else:
                out = _AttnTimesValue.apply(attn, v.contiguous(), shared)
        else:
            out = torch.matmul(attn, v)
        out = out.transpose(-1, -2).reshape(b, -1, h, w)                     # b (h d) x y
        if self.project is not None:
            out = self.project(out)
        return fmap + self.gamma * out


class GMAUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dim=128):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(4, 4)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = _mask_head()
        self._mask_cache = {}
        self.aggregator = Aggregate(args=args, dim=128, dim_head=128, heads=args.num_heads)

    def forward(self, net, inp, corr, flow, attention, want_mask=True, gru_ctx=None, attn_grad=None):
        motion_features = self.encoder(flow, corr)
        motion_features_global = self.aggregator(attention, motion_features, attn_grad)
        if gru_ctx is not None:
            net = self.gru.step(net, gru_ctx, torch.cat([motion_features, motion_features_global], dim=1))
        else:
            net = self.gru(net, torch.cat([inp, motion_features, motion_features_global], dim=1))
        delta_flow = self.flow_head(net)
        mask = mask_logits(self.mask, net, self._mask_cache) if want_mask else None
        return net, mask, delta_flow


class RAFTGMA(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        args.corr_levels = 4
        args.corr_radius = 4
        if not hasattr(args, 'dropout'):
            args.dropout = 0
        self.fnet = BasicEncoder(output_dim=256, norm_fn='instance', dropout=args.dropout)
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn='batch', dropout=args.dropout)
        self.update_block = GMAUpdateBlock(args, hidden_dim=hdim)
        self.att = Attention(args=args, dim=cdim, heads=args.num_heads, max_pos_size=160, dim_head=cdim)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        # both images normalised in one launch: the feature encoder's batch of two and the context encoder's input
        images12, image1 = ops.get().pm1_pair(image1, image2)
        hdim, cdim = self.hidden_dim, self.context_dim

        fmap1, fmap2 = self.fnet(images12, split=image1.shape[0])
        corr_fn = ops.get().CorrBlock(_f32(fmap1), _f32(fmap2), num_levels=4, radius=self.args.corr_radius,
                                      bwd_windows=cfg(self).pyramid_bwd_windows)

        net, inp = torch.split(self.cnet(image1), [hdim, cdim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        attention = self.att(inp)

        N, _, H, W = image1.shape
        coords0 = coords_grid(N, H // 8, W // 8, image1.device, _f32(image1).dtype)
        coords1 = coords_grid(N, H // 8, W // 8, image1.device, _f32(image1).dtype)
        if flow_init is not None:
            coords1 = coords1 + flow_init

        gru = self.update_block.gru
        gru_ctx = gru.per_iteration(gru.precompute(inp), iters) if gru.frozen() else None
        flow_predictions = []
        flow_up = None
        o = ops.get()   # the gradient of `attention` (used `iters` times) is formed once, by the last node that runs
        attn_grad = o.AttnGradShare(cfg(self).gma_gemm) if hasattr(o, "AttnGradShare") else _SharedAttnGrad()
        flow_cur = coords1 - coords0
        for itr in range(iters):
            coords1 = coords1.detach()
            corr = LookupRef(corr_fn, coords1)
            flow = flow_cur.detach()
            need_up = (not test_mode) or itr == iters - 1
            net, up_mask, delta_flow = self.update_block(net, inp, corr, flow, attention, want_mask=need_up,
                                                         gru_ctx=None if gru_ctx is None else gru_ctx[itr],
                                                         attn_grad=attn_grad)
            coords1, flow_cur = o.flow_step(coords1, delta_flow, coords0)   # see nets/raft.py
            if need_up:
                flow_up = convex_upsample(flow_cur, up_mask)
                flow_predictions.append(flow_up)
        if test_mode:
            return flow_cur, flow_up
        return flow_predictions
