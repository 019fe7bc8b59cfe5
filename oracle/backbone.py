"""CPU restatement of the reference's backbone: mmdet `ResNet(depth=50, num_stages=4,
out_indices=(0,1,2,3), style="pytorch", norm_cfg=BN, norm_eval=True)` as configured in
configs/mask2former/pairnet.py:9-19.  TEST INFRASTRUCTURE.

mmdet 2.25.1 is not vendored in /root/reference and not installed, so this restates its
published ResNet-50 (identical in structure and parameter names to torchvision's: stride on
the 3x3 convolution of the first block of stages 2-4, 1x1 stride-s projection shortcut,
BatchNorm in eval mode with eps 1e-5) from torch.nn primitives.  Unpinned against mmdet
itself; pinned against an independent implementation of the same network, HuggingFace
`transformers.ResNetBackbone`, with identical weights
(tests/test_oracle.py::test_resnet50_oracle_matches_transformers_resnet).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
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
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        return F.relu(self.bn3(self.conv3(out)) + identity)


class OracleResNet50(nn.Module):
    """depth 50 (default) or 101: mmdet's bottleneck arch_settings (3,4,6,3) / (3,4,23,3)."""
    BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth=50):
        super().__init__()
        self.STAGES = tuple(zip((64, 128, 256, 512), self.BLOCKS[depth]))
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (planes, blocks) in enumerate(self.STAGES):
            layers = []
            for b in range(blocks):
                layers.append(Bottleneck(cin, planes, 2 if (b == 0 and i > 0) else 1, b == 0))
                cin = planes * 4
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*layers))
        self.eval()

    @torch.no_grad()
    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), kernel_size=3, stride=2, padding=1)
        outs = []
        for i in range(4):
            x = getattr(self, "layer%d" % (i + 1))(x)
            outs.append(x)
        return tuple(outs)


def seeded_backbone_state(seed, depth=50):
    """Deterministic ResNet-50 state dict (numpy PCG64): He-style conv weights, BatchNorm
    statistics away from the identity so that folding is exercised."""
    import numpy as np
    rng = np.random.default_rng(seed)
    sd = OracleResNet50(depth).state_dict()
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy(rng.normal(0, 0.1, shape).astype(np.float32))
        elif ".bn" in k or k.startswith("bn1") or "downsample.1" in k:
            lo, hi = (0.5, 1.5) if k.endswith("weight") else (-0.1, 0.1)
            out[k] = torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32))
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            out[k] = torch.from_numpy(
                (rng.normal(0, 1, shape) * (1.0 / fan_in) ** 0.5).astype(np.float32))
    return out
