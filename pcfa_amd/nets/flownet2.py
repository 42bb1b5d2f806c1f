"""FlowNet2 (Ilg et al. 2017) for the PCFA hot path, written against pcfa_amd.ops.

Behavioural reference (cv-stuttgart/PCFA):
    models/FlowNet/FlowNet2.py:21-98 (stack), :115-177 (forward)
    models/FlowNet/FlowNetC.py:14-128      models/FlowNet/FlowNetS.py:14-95
    models/FlowNet/FlowNetSD.py:11-106     models/FlowNet/FlowNetFusion.py:11-67
    models/FlowNet/submodules.py:7-36 (conv / i_conv / predict_flow / deconv builders)
Module and parameter names follow FlowNet2_checkpoint.pth.tar (`flownetc.conv1.0.weight`, ...), so the public
checkpoint loads with `load_state_dict`.  The reference hard-codes batchNorm=False, fp16=False, rgb_max=255,
div_flow=20 (helper_functions/ownutilities.py:147-155); so does this file.

The three CUDA-only extensions of the reference (correlation_cuda, resample2d_cuda, channelnorm_cuda) are the HIP
operators `flownet_correlation`, `resample2d`, `channelnorm` of pcfa_amd.ops; frozen 3x3 / stride-1 convolutions
run as ops.conv3x3 (Winograd on the fp32 matrix cores, bias + LeakyReLU in the epilogue).
"""
import torch
import torch.nn as nn
from torch.nn import init

from .. import ops

# ops.conv3x3 works on 8x16-pixel tiles x 32 output channels; below this many pixels the grid cannot fill the chip
_CONV3X3_MIN_PIXELS = 1024


def _hip_conv_ok(c, x):
    return (c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1)
            and c.out_channels >= 16 and x.shape[-1] * x.shape[-2] >= _CONV3X3_MIN_PIXELS
            and not c.weight.requires_grad and not (c.bias is not None and c.bias.requires_grad))


class _ConvLeaky(nn.Sequential):
    """conv(batchNorm=False, ...) of submodules.py:7-19: Conv2d + LeakyReLU(0.1), parameter names "0.weight"/"0.bias"."""

    def forward(self, x):
        c = self[0]
        if _hip_conv_ok(c, x):
            return ops.get().conv3x3(x, c.weight, c.bias, False, self[1].negative_slope)
        return super().forward(x)


class _ConvLinear(nn.Sequential):
    """i_conv(batchNorm=False, ...) of submodules.py:21-31: a bare Conv2d inside a Sequential."""

    def forward(self, x):
        c = self[0]
        if _hip_conv_ok(c, x):
            return ops.get().conv3x3(x, c.weight, c.bias, False)
        return super().forward(x)


def conv(in_planes, out_planes, kernel_size=3, stride=1):
    return _ConvLeaky(
        nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=(kernel_size - 1) // 2,
                  bias=True),
        nn.LeakyReLU(0.1))


def i_conv(in_planes, out_planes, kernel_size=3, stride=1, bias=True):
    return _ConvLinear(
        nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=(kernel_size - 1) // 2,
                  bias=bias))


class _PredictFlow(nn.Conv2d):
    """A 3x3 flow-prediction layer (2 output channels).  Frozen weights stream through ops.conv3x3_fewout (an
    HBM-bound kernel) instead of a library convolution padded to a matrix-core tile; parameter names unchanged."""

    def forward(self, x):
        if not (self.weight.requires_grad or (self.bias is not None and self.bias.requires_grad)):
            return ops.get().conv3x3_fewout(x, self.weight, self.bias)
        return super().forward(x)


def predict_flow(in_planes):
    return _PredictFlow(in_planes, 2, kernel_size=3, stride=1, padding=1, bias=True)


def deconv(in_planes, out_planes):
    return nn.Sequential(nn.ConvTranspose2d(in_planes, out_planes, kernel_size=4, stride=2, padding=1, bias=True),
                         nn.LeakyReLU(0.1))


def _init(module):
    """Initialisation loop shared by every sub-network (e.g. FlowNetC.py:59-69)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if m.bias is not None:
                init.uniform_(m.bias)
            init.xavier_uniform_(m.weight)


class _Refinement:
    """The decoder every FlowNetC/S shares (FlowNetC.py:101-123 == FlowNetS.py:66-90)."""

    def _make_decoder(self, flow_bias):
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(1026)
        self.predict_flow4 = predict_flow(770)
        self.predict_flow3 = predict_flow(386)
        self.predict_flow2 = predict_flow(194)
        for name in ("6_to_5", "5_to_4", "4_to_3", "3_to_2"):
            setattr(self, "upsampled_flow" + name, nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=flow_bias))

    def _decode(self, out_conv2, out_conv3, out_conv4, out_conv5, out_conv6):
        flow6 = self.predict_flow6(out_conv6)
        concat5 = torch.cat((out_conv5, self.deconv5(out_conv6), self.upsampled_flow6_to_5(flow6)), 1)
        flow5 = self.predict_flow5(concat5)
        concat4 = torch.cat((out_conv4, self.deconv4(concat5), self.upsampled_flow5_to_4(flow5)), 1)
        flow4 = self.predict_flow4(concat4)
        concat3 = torch.cat((out_conv3, self.deconv3(concat4), self.upsampled_flow4_to_3(flow4)), 1)
        flow3 = self.predict_flow3(concat3)
        concat2 = torch.cat((out_conv2, self.deconv2(concat3), self.upsampled_flow3_to_2(flow3)), 1)
        flow2 = self.predict_flow2(concat2)
        if self.training:
            return flow2, flow3, flow4, flow5, flow6
        return flow2,


class FlowNetC(nn.Module, _Refinement):
    """FlowNetC.py:14-128."""

    def __init__(self, div_flow=20):
        super().__init__()
        self.div_flow = div_flow
        self.conv1 = conv(3, 64, kernel_size=7, stride=2)
        self.conv2 = conv(64, 128, kernel_size=5, stride=2)
        self.conv3 = conv(128, 256, kernel_size=5, stride=2)
        self.conv_redir = conv(256, 32, kernel_size=1, stride=1)
        self.corr_args = dict(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
        self.corr_activation = nn.LeakyReLU(0.1)
        self.conv3_1 = conv(473, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self._make_decoder(flow_bias=True)
        _init(self)

    def forward(self, x):
        x1 = x[:, 0:3, :, :]
        x2 = x[:, 3::, :, :]
        out_conv2a = self.conv2(self.conv1(x1))
        out_conv3a = self.conv3(out_conv2a)
        out_conv3b = self.conv3(self.conv2(self.conv1(x2)))
        out_corr = self.corr_activation(ops.get().flownet_correlation(out_conv3a, out_conv3b, **self.corr_args))
        in_conv3_1 = torch.cat((self.conv_redir(out_conv3a), out_corr), 1)
        out_conv3_1 = self.conv3_1(in_conv3_1)
        out_conv4 = self.conv4_1(self.conv4(out_conv3_1))
        out_conv5 = self.conv5_1(self.conv5(out_conv4))
        out_conv6 = self.conv6_1(self.conv6(out_conv5))
        return self._decode(out_conv2a, out_conv3_1, out_conv4, out_conv5, out_conv6)


class FlowNetS(nn.Module, _Refinement):
    """FlowNetS.py:14-95."""

    def __init__(self, input_channels=12):
        super().__init__()
        self.conv1 = conv(input_channels, 64, kernel_size=7, stride=2)
        self.conv2 = conv(64, 128, kernel_size=5, stride=2)
        self.conv3 = conv(128, 256, kernel_size=5, stride=2)
        self.conv3_1 = conv(256, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self._make_decoder(flow_bias=False)
        _init(self)

    def forward(self, x):
        out_conv2 = self.conv2(self.conv1(x))
        out_conv3 = self.conv3_1(self.conv3(out_conv2))
        out_conv4 = self.conv4_1(self.conv4(out_conv3))
        out_conv5 = self.conv5_1(self.conv5(out_conv4))
        out_conv6 = self.conv6_1(self.conv6(out_conv5))
        return self._decode(out_conv2, out_conv3, out_conv4, out_conv5, out_conv6)


class FlowNetSD(nn.Module):
    """FlowNetSD.py:11-106."""

    def __init__(self):
        super().__init__()
        self.conv0 = conv(6, 64)
        self.conv1 = conv(64, 64, stride=2)
        self.conv1_1 = conv(64, 128)
        self.conv2 = conv(128, 128, stride=2)
        self.conv2_1 = conv(128, 128)
        self.conv3 = conv(128, 256, stride=2)
        self.conv3_1 = conv(256, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.inter_conv5 = i_conv(1026, 512)
        self.inter_conv4 = i_conv(770, 256)
        self.inter_conv3 = i_conv(386, 128)
        self.inter_conv2 = i_conv(194, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(512)
        self.predict_flow4 = predict_flow(256)
        self.predict_flow3 = predict_flow(128)
        self.predict_flow2 = predict_flow(64)
        for name in ("6_to_5", "5_to_4", "4_to_3", "3_to_2"):
            setattr(self, "upsampled_flow" + name, nn.ConvTranspose2d(2, 2, 4, 2, 1))
        _init(self)

    def forward(self, x):
        out_conv0 = self.conv0(x)
        out_conv1 = self.conv1_1(self.conv1(out_conv0))
        out_conv2 = self.conv2_1(self.conv2(out_conv1))
        out_conv3 = self.conv3_1(self.conv3(out_conv2))
        out_conv4 = self.conv4_1(self.conv4(out_conv3))
        out_conv5 = self.conv5_1(self.conv5(out_conv4))
        out_conv6 = self.conv6_1(self.conv6(out_conv5))

        flow6 = self.predict_flow6(out_conv6)
        concat5 = torch.cat((out_conv5, self.deconv5(out_conv6), self.upsampled_flow6_to_5(flow6)), 1)
        flow5 = self.predict_flow5(self.inter_conv5(concat5))
        concat4 = torch.cat((out_conv4, self.deconv4(concat5), self.upsampled_flow5_to_4(flow5)), 1)
        flow4 = self.predict_flow4(self.inter_conv4(concat4))
        concat3 = torch.cat((out_conv3, self.deconv3(concat4), self.upsampled_flow4_to_3(flow4)), 1)
        flow3 = self.predict_flow3(self.inter_conv3(concat3))
        concat2 = torch.cat((out_conv2, self.deconv2(concat3), self.upsampled_flow3_to_2(flow3)), 1)
        flow2 = self.predict_flow2(self.inter_conv2(concat2))
        if self.training:
            return flow2, flow3, flow4, flow5, flow6
        return flow2,


class FlowNetFusion(nn.Module):
    """FlowNetFusion.py:11-67."""

    def __init__(self):
        super().__init__()
        self.conv0 = conv(11, 64)
        self.conv1 = conv(64, 64, stride=2)
        self.conv1_1 = conv(64, 128)
        self.conv2 = conv(128, 128, stride=2)
        self.conv2_1 = conv(128, 128)
        self.deconv1 = deconv(128, 32)
        self.deconv0 = deconv(162, 16)
        self.inter_conv1 = i_conv(162, 32)
        self.inter_conv0 = i_conv(82, 16)
        self.predict_flow2 = predict_flow(128)
        self.predict_flow1 = predict_flow(32)
        self.predict_flow0 = predict_flow(16)
        self.upsampled_flow2_to_1 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        _init(self)

    def forward(self, x):
        out_conv0 = self.conv0(x)
        out_conv1 = self.conv1_1(self.conv1(out_conv0))
        out_conv2 = self.conv2_1(self.conv2(out_conv1))
        flow2 = self.predict_flow2(out_conv2)
        concat1 = torch.cat((out_conv1, self.deconv1(out_conv2), self.upsampled_flow2_to_1(flow2)), 1)
        flow1 = self.predict_flow1(self.inter_conv1(concat1))
        concat0 = torch.cat((out_conv0, self.deconv0(concat1), self.upsampled_flow1_to_0(flow1)), 1)
        return self.predict_flow0(self.inter_conv0(concat0))


class FlowNet2(nn.Module):
    """FlowNet2.py:21-177 with batchNorm=False, fp16=False (the configuration import_and_load hard-codes)."""

    def __init__(self, rgb_max=255.0, div_flow=20.):
        super().__init__()
        self.div_flow = div_flow
        self.rgb_max = rgb_max
        self.flownetc = FlowNetC(div_flow=div_flow)
        self.upsample1 = nn.Upsample(scale_factor=4, mode='bilinear')
        self.flownets_1 = FlowNetS()
        self.upsample2 = nn.Upsample(scale_factor=4, mode='bilinear')
        self.flownets_2 = FlowNetS()
        self.flownets_d = FlowNetSD()
        self.upsample3 = nn.Upsample(scale_factor=4, mode='nearest')
        self.upsample4 = nn.Upsample(scale_factor=4, mode='nearest')
        self.flownetfusion = FlowNetFusion()
        _init(self)

    def _warp_block(self, x, flow):
        """Warp image 2 towards image 1 and take the brightness error (FlowNet2.py:128-134)."""
        o = ops.get()
        resampled = o.resample2d(x[:, 3:, :, :], flow)
        norm_diff = o.channelnorm(x[:, :3, :, :] - resampled)
        return resampled, norm_diff

    def forward(self, inputs):
        o = ops.get()
        rgb_mean = inputs.contiguous().view(inputs.size()[:2] + (-1,)).mean(dim=-1).view(
            inputs.size()[:2] + (1, 1, 1,))
        x = (inputs - rgb_mean) / self.rgb_max
        x = torch.cat((x[:, :, 0, :, :], x[:, :, 1, :, :]), dim=1)

        flownetc_flow = self.upsample1(self.flownetc(x)[0] * self.div_flow)
        resampled_img1, norm_diff_img0 = self._warp_block(x, flownetc_flow)
        concat1 = torch.cat((x, resampled_img1, flownetc_flow / self.div_flow, norm_diff_img0), dim=1)

        flownets1_flow = self.upsample2(self.flownets_1(concat1)[0] * self.div_flow)
        resampled_img1, norm_diff_img0 = self._warp_block(x, flownets1_flow)
        concat2 = torch.cat((x, resampled_img1, flownets1_flow / self.div_flow, norm_diff_img0), dim=1)

        flownets2_flow = self.upsample4(self.flownets_2(concat2)[0] * self.div_flow)
        norm_flownets2_flow = o.channelnorm(flownets2_flow)
        _, diff_flownets2_img1 = self._warp_block(x, flownets2_flow)

        flownetsd_flow = self.upsample3(self.flownets_d(x)[0] / self.div_flow)
        norm_flownetsd_flow = o.channelnorm(flownetsd_flow)
        _, diff_flownetsd_img1 = self._warp_block(x, flownetsd_flow)

        concat3 = torch.cat((x[:, :3, :, :], flownetsd_flow, flownets2_flow, norm_flownetsd_flow,
                             norm_flownets2_flow, diff_flownetsd_img1, diff_flownets2_img1), dim=1)
        return self.flownetfusion(concat3)
