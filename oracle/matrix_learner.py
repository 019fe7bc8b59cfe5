"""Matrix Learner restatement (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows pairnet/models/frameworks/cnn_factory.py:6-53 (`ConvTiny`) and :186-195
(`creat_cnn`): three 7x7 convolutions with padding 3, 1 -> mid -> mid -> 1
channels, ReLU after the first two, applied to the (B, Q, Q) importance matrix.
State-dict names: conv_layers.{0,1,2}.0.{weight,bias}.  Pinned against the
reference module imported by path (tests/test_oracle.py, this container only).
"""
import torch.nn as nn
import torch.nn.functional as F


class MatrixLearnerTiny(nn.Module):
    def __init__(self, mid_channels=64, kernel_size=7):
        super().__init__()
        chans = [1, mid_channels, mid_channels, 1]
        self.conv_layers = nn.ModuleList(
            nn.Sequential(nn.Conv2d(chans[i], chans[i + 1], kernel_size,
                                    padding=3))
            for i in range(3))

    def forward(self, importance):
        x = importance[:, None]
        for i, block in enumerate(self.conv_layers):
            x = block[0](x)
            if i < 2:
                x = F.relu(x)
        return x[:, 0]


def build_matrix_learner(name):
    if name != "conv_tiny":
        raise NotImplementedError(
            "only mapper='conv_tiny' is on the north-star path "
            "(configs/mask2former/pairnet.py:26)")
    return MatrixLearnerTiny()
