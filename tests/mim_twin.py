"""Test infrastructure: the MIM decoder (reference libs/vl_heads.py:107-165, `ITGHead`) as a plain PyTorch-ROCm module -- MIOpen
convolutions and ATen ops under autograd -- with the product model's parameter names.  It is the A/B reference of
tests/test_model_gpu.py::test_mim_decoder_hip_vs_torch_twin and is fed by the product model's own
`forward_pyramid_features_vl`; the product (mvlt_amd/) has exactly one backend, the HIP schedule of mvlt_amd/mim.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_bn(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout))


class MimTwin(nn.Module):
    def __init__(self, dims=(64, 128, 320, 512), ch=64):
        super().__init__()
        self.reduction1 = _conv_bn(dims[1], ch)
        self.reduction2 = _conv_bn(dims[2], ch)
        self.reduction3 = _conv_bn(dims[3], ch)
        self.conv_upsample1 = _conv_bn(ch, ch)
        self.conv_upsample2 = _conv_bn(ch, ch)
        self.conv_upsample3 = _conv_bn(ch, ch)
        self.conv_upsample4 = _conv_bn(ch, ch)
        self.conv_upsample5 = _conv_bn(2 * ch, 2 * ch)
        self.conv_concat2 = _conv_bn(2 * ch, 2 * ch)
        self.conv_concat3 = _conv_bn(3 * ch, 3 * ch)
        self.conv4 = _conv_bn(3 * ch, 3 * ch)
        self.score = nn.Sequential(nn.Conv2d(3 * ch, 3, 1))

    def forward(self, f1, f2, f3, conv_dtype):
        """conv3x3 operands in `conv_dtype` (bf16 MFMA on MIOpen), everything else -- BatchNorm statistics and normalisation, the
        align_corners bilinear resizes, the three-way feature products -- in fp32 (the split the HIP schedule uses)."""
        def cb(seq, t):
            y = F.conv2d(t.to(conv_dtype), seq[0].weight.to(conv_dtype), None, padding=1).float()
            return seq[1](y)
        up = lambda t, s=2: F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=True)
        low, mid, high = cb(self.reduction1, f1), cb(self.reduction2, f2), cb(self.reduction3, f3)
        a = cb(self.conv_upsample1, up(high)) * mid
        b = cb(self.conv_upsample2, up(mid)) * cb(self.conv_upsample3, up(a)) * low
        c = cb(self.conv_concat2, torch.cat((a, cb(self.conv_upsample4, up(high))), 1))
        d = cb(self.conv_concat3, torch.cat((b, cb(self.conv_upsample5, up(c))), 1))
        e = cb(self.conv4, d)
        sc = self.score[0]
        s = F.conv2d(e.to(conv_dtype), sc.weight.to(conv_dtype), None).float() + sc.bias.view(1, -1, 1, 1)
        return up(s, 8)
