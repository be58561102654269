"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's VggLoss (reference src/loss.py:29-49).

PARITY UNPINNED: the reference class cannot be constructed here (torchvision is not installed and
vgg19(pretrained=True), loss.py:33, is a weight download), so this restates the published VGG19 'E'
configuration's first 27 feature modules (conv3x3+ReLU x2 @64, pool, x2 @128, pool, x4 @256, pool, x4 @512 =
through relu4_4; 10 585 152 parameters as SURVEY.md section 8 a10 counted) with torchvision's state-dict keys,
and the loss expression of loss.py:43-47.  Weights are name-seeded random.  Only tests/ import this file.
"""
import math
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512)


def param_shapes():
    s, idx, cin = OrderedDict(), 0, 3
    for v in CFG:
        if v == "M":
            idx += 1
        else:
            s["features.%d.weight" % idx] = (v, cin, 3, 3)
            s["features.%d.bias" % idx] = (v,)
            cin = v
            idx += 2
    return s


def test_params(seed=0):
    out = {}
    for name, shape in param_shapes().items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7fffffff)
        u = torch.rand(shape, generator=g) * 2 - 1
        out[name] = u * math.sqrt(6.0 / (shape[1] * 9)) if len(shape) == 4 else u * 0.1
    return out


def features(p, x):
    idx = 0
    for v in CFG:
        if v == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, p["features.%d.weight" % idx], p["features.%d.bias" % idx], padding=1))
            idx += 2
    return x


def vgg_loss(p, output, target):                # loss.py:43-47
    return (features(p, output) - features(p, target)).abs().mean()
