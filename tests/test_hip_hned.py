"""GPU: the reference's frozen HED edge detector (reference src/models/hned.py:9-105), forward, against all six
outputs captured from the reference module itself (tests/golden/hned.npz; name-seeded weights - the trained ones
are at an author-local path, trainer.py:97, so structure and arithmetic are pinned, the learned function is not)
and against the CPU restatement on one more shape.  fp32, 1e-4."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close
from oracle import hned_spec as HS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["a", "b"])
def test_hned_matches_reference_outputs(dev, tag):
    from vlg.hned import HNEDHIP
    z = np.load(os.path.join(GOLDEN, "hned.npz"))
    x = torch.from_numpy(z[tag + "_x"])
    net = HNEDHIP(x.shape[0], x.shape[2], x.shape[3], dev)
    assert net.reference_shapes() == {k: tuple(v) for k, v in HS.param_shapes().items()}
    net.load_state_dict(HS.test_params(0))
    out = net.forward(x.to(dev))
    assert_close(out, torch.from_numpy(z[tag + "_out"]), rtol=1e-4, atol=1e-5, what="HED d1..d5, fuse")


def test_hned_256_against_restatement(dev):
    from vlg.hned import HNEDHIP
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(1))
    p = HS.test_params(3)
    net = HNEDHIP(1, 256, 256, dev)
    net.load_state_dict(p)
    want = torch.stack([o[:, 0] for o in HS.forward(p, x)])
    assert_close(net.forward(x.to(dev)), want, rtol=1e-4, atol=1e-5, what="HED 256x256")
