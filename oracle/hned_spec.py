"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's frozen HED edge detector (forward).

Functional torch-CPU restatement of reference src/models/hned.py:9-105 over a {state_dict key: tensor}
dict.  PINNED against outputs of the reference module itself (tests/golden/hned_*.npz, written by
oracle/make_golden.py) - with name-seeded random weights, because the trained weights live at an
author-local path (reference src/trainer.py:97) and are not in the repository: structure and arithmetic
are pinned, the learned function is not.  Only tests/ import this file.
"""
import math
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

STAGES = (("moduleVggOne", 3, 64, (0, 2)), ("moduleVggTwo", 64, 128, (1, 3)), ("moduleVggThr", 128, 256, (1, 3, 5)),
          ("moduleVggFou", 256, 512, (1, 3, 5)), ("moduleVggFiv", 512, 512, (1, 3, 5)))
SCORES = ("moduleScoreOne", "moduleScoreTwo", "moduleScoreThr", "moduleScoreFou", "moduleScoreFiv")


def param_shapes():
    s = OrderedDict()
    for name, cin, cout, idx in STAGES:          # hned.py:13-57
        c = cin
        for i in idx:
            s["%s.%d.weight" % (name, i)] = (cout, c, 3, 3)
            s["%s.%d.bias" % (name, i)] = (cout,)
            c = cout
    for name, (_, _, cout, _) in zip(SCORES, STAGES):   # hned.py:60-64
        s[name + ".weight"] = (1, cout, 1, 1)
        s[name + ".bias"] = (1,)
    s["moduleCombine.0.weight"] = (1, 5, 1, 1)           # hned.py:66-69
    s["moduleCombine.0.bias"] = (1,)
    return s


def test_params(seed=0):
    out = {}
    for name, shape in param_shapes().items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7fffffff)
        fan_in = shape[1] * shape[2] * shape[3] if len(shape) == 4 else 64
        u = torch.rand(shape, generator=g) * 2 - 1
        if name.startswith("moduleScore") and len(shape) == 4:
            out[name] = u * (0.012 * math.sqrt(6.0 / fan_in))   # features are O(100): keep the scores O(1), sigmoids unsaturated
        elif len(shape) == 4:
            out[name] = u * math.sqrt(6.0 / fan_in)              # He-style: the 13-conv ReLU stack neither vanishes nor explodes
        else:
            out[name] = u * 0.5
    return out


def forward(p, x):
    """hned.py:73-105.  x (b,3,H,W) in [0,1] -> (d1, d2, d3, d4, d5, fuse), each (b,1,H,W)."""
    t = torch.cat([x[:, 0:1] * 255.0 - 104.00698793, x[:, 1:2] * 255.0 - 116.66876762, x[:, 2:3] * 255.0 - 122.67891434], 1)
    H, W = t.shape[2], t.shape[3]
    feats = []
    for si, (name, cin, cout, idx) in enumerate(STAGES):
        if si > 0:
            t = F.max_pool2d(t, kernel_size=2, stride=2)
        for i in idx:
            t = F.relu(F.conv2d(t, p["%s.%d.weight" % (name, i)], p["%s.%d.bias" % (name, i)], padding=1))
        feats.append(t)
    scores = [F.interpolate(F.conv2d(f, p[n + ".weight"], p[n + ".bias"]), size=(H, W), mode="bilinear", align_corners=False)
              for n, f in zip(SCORES, feats)]
    fuse = torch.sigmoid(F.conv2d(torch.cat(scores, 1), p["moduleCombine.0.weight"], p["moduleCombine.0.bias"]))
    return tuple(torch.sigmoid(s) for s in scores) + (fuse,)
