"""PWC-DC-Net (Sun et al. 2018) for the PCFA hot path, written against pcfa_amd.ops.

Behavioural reference (cv-stuttgart/PCFA): models/PWCNet/PWCNet.py
    :29-43  conv / predict_flow / deconv builders      :45-58   correlate
    :65-164 layer table                                :166-206 warp
    :227-330 forward
Parameter names follow pwc_net_chairs.pth.tar.

The five 9x9 cost volumes run in the HIP spatial-correlation kernel on the
device the features live on; the reference's default bounces every one of them
through a CPU-only sampler (config_paths.py:30, PWCNet.py:18-21).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..config import cfg


# Build switches (sub-grid form of the dilated layers, own stride-2 / deconv kernels, deferred LeakyReLU masks ...) come
# from the frozen pcfa_amd.config.Config the model was built with: cfg(module).<switch>.


def _regrid(x, d_from, d_to, batch):
    """x in sub-grid layout d_from -- [batch * d_from^2, C, H / d_from, W / d_from], image (gy, gx) = the pixels with
    (y mod d_from, x mod d_from) = (gy, gx); d = 1 is the plain tensor -- to layout d_to, as ONE strided copy where d_to is
    a multiple of d_from or 1 (consecutive dilated layers: 1 -> 2 -> 4 -> 8 -> 16 -> 1 in the context network)."""
    if d_from == d_to:
        return x
    _, C, h, w = x.shape
    # the full-resolution tensor as a view: [B, C, H/d_from, d_from, W/d_from, d_from], y = sy * d_from + gy
    full = x.reshape(batch, d_from, d_from, C, h, w).permute(0, 3, 4, 1, 5, 2)
    if d_to == 1:
        return full.reshape(batch, C, h * d_from, w * d_from)
    if d_to % d_from:
        return _regrid(full.reshape(batch, C, h * d_from, w * d_from), 1, d_to, batch)
    r = d_to // d_from   # y mod d_to = (sy mod r) * d_from + gy,  y div d_to = sy div r
    v = full.reshape(batch, C, h // r, r, d_from, w // r, r, d_from).permute(0, 3, 4, 6, 7, 1, 2, 5)
    return v.reshape(batch * d_to * d_to, C, h // r, w // r)


class _ConvLeaky(nn.Sequential):
    """conv() of the reference (PWCNet.py:29-35): Conv2d + LeakyReLU(0.1); same parameter names ("0.weight", "0.bias").
    Frozen 3x3 / stride 1 / pad 1 instances run as ops.conv3x3 (Winograd on the fp32 matrix cores, bias and
    LeakyReLU in the epilogue), 3x3 / stride 2 ones as ops.conv_s2, dilated ones as d*d plain convolutions on sub-grids.

    Chains of such layers (pyramid level: conv_a -> conv_aa -> conv_b; context network: dc_conv1 .. dc_conv6) defer the
    LeakyReLU backward of a layer whose output has ONE consumer into that consumer's data-gradient epilogue
    (`grad_premasked` on the producer, `mask_input_grad` on the consumer -- the contract of ops.conv3x3): the caller
    sets the flags only where `kind()` says both sides run on the operator table."""

    def _frozen(self):
        c = self[0]
        return not (c.weight.requires_grad or (c.bias is not None and c.bias.requires_grad))

    def grid(self, x):
        """The sub-grid layout this layer computes in: its dilation on the "subgrid" path, else 1."""
        return self[0].dilation[0] if self.kind(x) == "subgrid" else 1

    def _bgr_weight(self):
        """The weight with its three input channels reversed: conv(x[:, [2, 1, 0]], w) = conv(x, w[:, [2, 1, 0]]) -- the
        RGB -> BGR stack of PWCNet.py:231-232 (two concatenations forward, six fills / copies / adds backward at full
        image size) costs nothing this way.  Cached per weight version (the frozen path)."""
        w = self[0].weight
        hit = getattr(self, "_bgr_cache", None)
        if hit is None or hit[0] != w._version or hit[1].device != w.device or hit[1].dtype != w.dtype:
            hit = (w._version, w.detach().flip(1).contiguous())
            object.__setattr__(self, "_bgr_cache", hit)
        return hit[1]

    def kind(self, x):
        """Which path forward(x) takes: "conv3x3" | "subgrid" | "s2" | None (the library modules).  x: the input at full
        resolution (or a meta tensor of that shape)."""
        c = self[0]
        if not self._frozen() or c.kernel_size != (3, 3) or c.groups != 1:
            return None
        if c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) and c.out_channels >= 16:
            return "conv3x3"
        d = c.dilation[0]
        conf = cfg(self)
        if (conf.dilated_as_subgrids and d in conf.dilated_as_subgrids and c.stride == (1, 1) and c.dilation == (d, d)
                and c.padding == (d, d) and c.out_channels >= 16 and x.shape[-2] % d == 0 and x.shape[-1] % d == 0):
            return "subgrid"
        if (conf.conv_s2 and c.stride == (2, 2) and c.padding == (1, 1) and c.dilation == (1, 1)
                and ops.get().conv_s2_supported(x, c.weight)):
            return "s2"
        return None

    def forward(self, x, grad_premasked=False, mask_input_grad=False, grid_in=1, grid_out=1, bgr_input=False, skip=False):
        """grid_in / grid_out: the sub-grid layout (_regrid) x arrives in / the result is wanted in -- _chain passes the
        next dilated layer's dilation so that two such layers exchange ONE copy; bgr_input: x is the RGB image and the
        layer is to see it as BGR (PWCNet.py:231-232); skip: also return an alias of x for its other consumer (ops.conv3x3:
        the gradient arriving there is added in this layer's data-gradient epilogue) -- plain conv3x3 layers only."""
        c, slope = self[0], self[1].negative_slope
        batch = x.shape[0] // (grid_in * grid_in)
        probe = x if grid_in == 1 else torch.empty((batch, x.shape[1], x.shape[2] * grid_in, x.shape[3] * grid_in),
                                                   device="meta")
        kind = self.kind(probe)
        if skip:
            if kind != "conv3x3" or grid_in != 1:
                raise RuntimeError("skip needs a plain 3x3 layer on the operator table")
            y, alias = ops.get().conv3x3(x, c.weight, c.bias, False, slope, skip=True, grad_premasked=grad_premasked,
                                         mask_input_grad=mask_input_grad, input_slope=slope if mask_input_grad else 0.)
            return _regrid(y, 1, grid_out, batch), alias
        if kind is None:
            if grad_premasked or mask_input_grad:
                raise RuntimeError("deferred LeakyReLU masks need the operator-table path on both sides (kind() is None)")
            x = _regrid(x, grid_in, 1, batch)
            return _regrid(super().forward(x.flip(1) if bgr_input else x), 1, grid_out, batch)
        kw = dict(grad_premasked=grad_premasked)
        weight = self._bgr_weight() if bgr_input else c.weight
        if kind == "s2":
            # the pyramid's stride-2 layers (PWCNet.py:87-104): direct fp32-MFMA kernel, bias + LeakyReLU in the epilogue
            if mask_input_grad:
                raise RuntimeError("conv_s2 does not apply an input mask")
            y = ops.get().conv_s2(_regrid(x, grid_in, 1, batch), weight, c.bias, leaky_slope=slope,
                                  own_bwd=cfg(self).conv_s2_bwd, **kw)
            return _regrid(y, 1, grid_out, batch)
        kw.update(mask_input_grad=mask_input_grad, input_slope=slope if mask_input_grad else 0.)
        # A 3x3 convolution with dilation d (the context network, PWCNet.py:160-166) never mixes pixels of different
        # (y mod d, x mod d): it is d*d independent plain 3x3 convolutions on the sub-sampled grids, zero padding
        # included (H, W multiples of d).  Sub-grids go to the batch axis and run on the Winograd kernel.
        d = c.dilation[0] if kind == "subgrid" else 1
        ys = ops.get().conv3x3(_regrid(x, grid_in, d, batch).contiguous(), weight, c.bias, False, slope, **kw)
        return _regrid(ys, d, grid_out, batch)


def _chain(layers, x, last_premasked=False, bgr_input=False, skip_first=False):
    """x -> layers[0] -> layers[1] -> ...: every intermediate output has exactly one consumer, so its LeakyReLU backward is
    applied by that consumer's data-gradient kernel wherever both layers run on the operator table (and the consumer is
    a stride-1 convolution, whose kernel has the mask epilogue)."""
    if not (layers and cfg(layers[0]).defer_leaky):
        alias = None
        for i, layer in enumerate(layers):
            x = layer(x, bgr_input=bgr_input and i == 0, skip=skip_first and i == 0)
            if skip_first and i == 0:
                x, alias = x
        return (x, alias) if skip_first else x
    masked_in = False
    regrid = cfg(layers[0]).pwc_fold_glue
    grid = 1           # the sub-grid layout x is in
    hw = x.shape[-2:]  # its full-resolution size
    batch = x.shape[0]
    for i, layer in enumerate(layers):
        # (the real tensor where it is at full resolution: conv_s2_supported looks at more than the shape)
        k = layer.kind(x if grid == 1 else torch.empty((batch, x.shape[1]) + tuple(hw), device="meta"))
        nxt = layers[i + 1] if i + 1 < len(layers) else None
        # the next layer's path depends on its input only through the spatial size (sub-grid divisibility), which a
        # stride-1 / stride-2 3x3 layer fixes: decide on a shape probe
        premask = False
        c = layer[0]
        hw = tuple((n + 2 * c.padding[j] - c.dilation[j] * (c.kernel_size[j] - 1) - 1) // c.stride[j] + 1
                   for j, n in enumerate(hw))   # this layer's output size
        nxt_grid = 1
        if k is not None and nxt is not None:
            probe = torch.empty((batch, c.out_channels) + hw, device="meta")
            premask = nxt.kind(probe) in ("conv3x3", "subgrid")
            # hand the result over in the layout the next layer computes in (one copy instead of two); a deferred mask
            # is element-wise, so producer and consumer only have to agree on the layout -- they do, it is this tensor
            nxt_grid = nxt.grid(probe) if regrid else 1
        x = layer(x, grad_premasked=premask, mask_input_grad=masked_in, grid_in=grid, grid_out=nxt_grid,
                  bgr_input=bgr_input and i == 0, skip=skip_first and i == 0)
        if skip_first and i == 0:
            x, alias = x
        grid = nxt_grid
        masked_in = premask
    return (x, alias) if skip_first else x


def conv(in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1):
    return _ConvLeaky(
        nn.Conv2d(int(in_planes), int(out_planes), kernel_size=kernel_size, stride=stride, padding=padding,
                  dilation=dilation, bias=True),
        nn.LeakyReLU(0.1))


class _PredictFlow(nn.Conv2d):
    """A 3x3 flow-prediction layer (2 output channels).  Frozen weights stream through ops.conv3x3_fewout (an
    HBM-bound kernel) instead of a library convolution padded to a matrix-core tile; parameter names unchanged."""

    def forward(self, x, skip=False):
        """skip: also return an alias of x for its other consumer (ops.conv3x3_fewout)."""
        if not (self.weight.requires_grad or (self.bias is not None and self.bias.requires_grad)):
            return ops.get().conv3x3_fewout(x, self.weight, self.bias, skip=skip)
        y = super().forward(x)
        return (y, x) if skip else y


def predict_flow(in_planes):
    return _PredictFlow(int(in_planes), 2, kernel_size=3, stride=1, padding=1, bias=True)


class _Deconv(nn.ConvTranspose2d):
    """deconv() of the reference (PWCNet.py:42-43).  Frozen 4x4 / stride 2 / pad 1 instances with at most 4 output
    channels (every one PWC-Net has) stream through ops.deconv4s2_fewout: fixed summation order, where the library
    picks implicit-GEMM / GEMM + col2im / first-call fallback kernels per process; parameter names unchanged."""

    def forward(self, x):
        if (self.kernel_size == (4, 4) and self.stride == (2, 2) and self.padding == (1, 1)
                and self.output_padding == (0, 0) and self.dilation == (1, 1) and self.groups == 1
                and self.out_channels <= 4 and cfg(self).deconv_fewout
                and not (self.weight.requires_grad or (self.bias is not None and self.bias.requires_grad))):
            return ops.get().deconv4s2_fewout(x, self.weight, self.bias)
        return super().forward(x)


def deconv(in_planes, out_planes, kernel_size=4, stride=2, padding=1):
    return _Deconv(int(in_planes), int(out_planes), kernel_size, stride, padding, bias=True)


def correlate(input1, input2):
    """9x9 cost volume averaged over channels (PWCNet.py:45-58)."""
    out = ops.get().spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=9, stride=1)
    b, ph, pw, h, w = out.size()
    return out.view(b, ph * pw, h, w) / input1.size(1)


_PYRAMID = [(3, 16), (16, 32), (32, 64), (64, 96), (96, 128), (128, 196)]


class PWCDCNet(nn.Module):
    def __init__(self, md=4):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=4, mode='bilinear')
        for lvl, (cin, cout) in enumerate(_PYRAMID, start=1):
            # level 6 names its stride-2 conv "aa" and the next one "a" (checkpoint quirk, PWCNet.py:90-92)
            first, second = ("aa", "a") if lvl == 6 else ("a", "aa")
            setattr(self, "conv%d%s" % (lvl, first), conv(cin, cout, kernel_size=3, stride=2))
            setattr(self, "conv%d%s" % (lvl, second), conv(cout, cout, kernel_size=3, stride=1))
            setattr(self, "conv%db" % lvl, conv(cout, cout, kernel_size=3, stride=1))
        self.corr = correlate
        self.leakyRELU = nn.LeakyReLU(0.1)

        nd = (2 * md + 1) ** 2
        dd = np.cumsum([128, 128, 96, 64, 32])
        for lvl, extra in ((6, 0), (5, 128 + 4), (4, 96 + 4), (3, 64 + 4), (2, 32 + 4)):
            od = nd + extra
            for i, (cin, cout) in enumerate(((od, 128), (od + dd[0], 128), (od + dd[1], 96), (od + dd[2], 64),
                                             (od + dd[3], 32))):
                setattr(self, "conv%d_%d" % (lvl, i), conv(cin, cout, kernel_size=3, stride=1))
            setattr(self, "predict_flow%d" % lvl, predict_flow(od + dd[4]))
            setattr(self, "deconv%d" % lvl, deconv(2, 2, kernel_size=4, stride=2, padding=1))
            if lvl > 2:
                setattr(self, "upfeat%d" % lvl, deconv(od + dd[4], 2, kernel_size=4, stride=2, padding=1))
        od = nd + 32 + 4
        self.dc_conv1 = conv(od + dd[4], 128, kernel_size=3, stride=1, padding=1, dilation=1)
        self.dc_conv2 = conv(128, 128, kernel_size=3, stride=1, padding=2, dilation=2)
        self.dc_conv3 = conv(128, 128, kernel_size=3, stride=1, padding=4, dilation=4)
        self.dc_conv4 = conv(128, 96, kernel_size=3, stride=1, padding=8, dilation=8)
        self.dc_conv5 = conv(96, 64, kernel_size=3, stride=1, padding=16, dilation=16)
        self.dc_conv6 = conv(64, 32, kernel_size=3, stride=1, padding=1, dilation=1)
        self.dc_conv7 = predict_flow(32)

        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight.data, mode='fan_in')
                if m.bias is not None:
                    m.bias.data.zero_()

    def warp(self, x, flo, scale=1.0):
        """Backward-warp x by scale * flo with a validity mask (PWCNet.py:166-206): one fused launch instead of the
        meshgrid / normalise / two grid_sample / compare / multiply sequence."""
        if scale != 1.0 and not cfg(self).pwc_fold_glue:
            flo, scale = flo * scale, 1.0
        return ops.get().pwc_warp(x, flo, 0.0001, deterministic=cfg(self).warp_bwd_deterministic, flow_scale=scale)

    def _cost_volume(self, f1, f2):
        """leakyRELU(corr(f1, f2)) (PWCNet.py:249,264,278,292,308) in one launch per direction."""
        return ops.get().pwc_cost_volume(f1, f2, self.leakyRELU.negative_slope)

    def _flow_head(self, lvl, x):
        """(predict_flow(x), x for upfeat): with Config.pwc_fold_glue the second is predict_flow's alias of x, so that
        upfeat's gradient is added inside predict_flow's data-gradient kernel instead of by an autograd add."""
        head = getattr(self, "predict_flow%d" % lvl)
        if cfg(self).pwc_fold_glue:
            return head(x, skip=True)
        return head(x), x

    def _decode(self, lvl, *parts):
        """The level's DenseNet block on cat(parts, 1) (PWCNet.py:250-255,265-270, ...)."""
        blocks = [getattr(self, "conv%d_%d" % (lvl, i)) for i in range(5)]
        if parts[0].shape[0] == 1 and not any(p.requires_grad for blk in blocks for p in blk.parameters()):
            # one pre-allocated buffer: the parts are written behind each other at its end, every convolution reads its
            # channel suffix in place and writes in front of it
            x = parts if len(parts) > 1 and cfg(self).pwc_fold_glue else (parts[0] if len(parts) == 1 else torch.cat(parts, 1))
            return ops.get().dense_block(x, [(blk[0].weight, blk[0].bias) for blk in blocks], blocks[0][1].negative_slope,
                                         fused_masks=cfg(self).dense_block_fused_masks)
        x = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        for blk in blocks:
            x = torch.cat((blk(x), x), 1)
        return x

    def forward(self, im1, im2):
        # RGB -> BGR (PWCNet.py:231-232): in conv1a's weights where that layer runs on the operator table
        fold_bgr = cfg(self).pwc_fold_glue and self.conv1a.kind(im1) is not None
        if not fold_bgr:
            im1 = torch.stack((im1[:, 2, :, :], im1[:, 1, :, :], im1[:, 0, :, :]), 1)
            im2 = torch.stack((im2[:, 2, :, :], im2[:, 1, :, :], im2[:, 0, :, :]), 1)

        def pyramid(im):
            feats, x = [], im
            for lvl in range(1, 7):
                first, second = ("aa", "a") if lvl == 6 else ("a", "aa")
                x = _chain([getattr(self, "conv%d%s" % (lvl, first)), getattr(self, "conv%d%s" % (lvl, second)),
                            getattr(self, "conv%db" % lvl)], x, bgr_input=fold_bgr and lvl == 1)
                feats.append(x)
            return feats

        if cfg(self).pwc_fold_glue and fold_bgr and im1.shape == im2.shape:
            # both pyramids in one batch (the layers are per-sample, so nothing is shared but the launches: half as
            # many, each with twice the workgroups -- the coarse levels are a handful of workgroups per image)
            b = im1.shape[0]
            c1, c2 = zip(*[ops.get().split_batch(f, b) for f in pyramid(torch.cat((im1, im2), 0))])
        else:
            c1, c2 = pyramid(im1), pyramid(im2)  # index 0 = level 1 ... index 5 = level 6

        corr6 = self._cost_volume(c1[5], c2[5])
        x = self._decode(6, corr6)
        flow, x = self._flow_head(6, x)
        flows = {6: flow}
        up_flow, up_feat = self.deconv6(flow), self.upfeat6(x)

        for lvl, scale in ((5, 0.625), (4, 1.25), (3, 2.5), (2, 5.0)):
            f1, f2 = c1[lvl - 1], c2[lvl - 1]
            warped = self.warp(f2, up_flow, scale)
            corr = self._cost_volume(f1, warped)
            x = self._decode(lvl, corr, f1, up_flow, up_feat)
            if lvl > 2:
                # (predict_flow hands upfeat an alias of x: upfeat's gradient is added in predict_flow's data-gradient kernel)
                flow, x = self._flow_head(lvl, x)
                flows[lvl] = flow
                up_flow = getattr(self, "deconv%d" % lvl)(flow)
                up_feat = getattr(self, "upfeat%d" % lvl)(x)

        # level 2's decoder output feeds predict_flow2 and the context network (PWCNet.py:311-316).  On the operator table
        # dc_conv1 hands back an alias of its input for the other consumer: that one's gradient is then added in
        # dc_conv1's data-gradient epilogue instead of by an autograd add over [565, H/4, W/4]
        dc = [self.dc_conv1, self.dc_conv2, self.dc_conv3, self.dc_conv4, self.dc_conv5, self.dc_conv6]
        if cfg(self).pwc_fold_glue and cfg(self).defer_leaky and self.dc_conv1.kind(x) == "conv3x3":
            ctx_feat, x = _chain(dc, x, skip_first=True)
            flows[2] = self.predict_flow2(x)
        else:
            flows[2] = self.predict_flow2(x)
            ctx_feat = _chain(dc, x)
        flow2 = flows[2] + self.dc_conv7(ctx_feat)

        if cfg(self).deconv_fewout and not self.training:
            return ops.get().upsample_bilinear(flow2, 4, 20.0)   # 20 * self.upsample(flow2), gather backward
        flow2 = 20 * self.upsample(flow2)
        if self.training:
            return (flow2,) + tuple(20 * self.upsample(flows[l]) for l in (3, 4, 5, 6))
        return flow2
