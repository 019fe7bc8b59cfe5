"""PyTorch-ROCm / MIOpen ResNet-50 with mmdet's state-dict names: the COMPARISON leg that
bench.py and tests/test_backbone_gpu.py run beside the native `ResNet50Hip`
(depth 50, style='pytorch', out_indices (0,1,2,3), frozen BN --
configs/mask2former/pairnet.py:9-19).  Not part of the product package."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        return F.relu(self.bn3(self.conv3(x)) + idt)


class ResNet50(nn.Module):
    """depth 50, style='pytorch', out_indices (0,1,2,3), frozen BN
    (configs/mask2former/pairnet.py:9-19)."""

    def __init__(self, **unused):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
            layers = []
            for b in range(blocks):
                layers.append(_Bottleneck(cin, planes, 2 if (b == 0 and i > 0) else 1, b == 0))
                cin = planes * 4
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*layers))
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        outs = []
        for i in range(4):
            x = getattr(self, "layer%d" % (i + 1))(x)
            outs.append(x.contiguous())
        return tuple(outs)
