#!/usr/bin/env python
"""Generates tests/golden/addcoords.npz from the reference's own AddCoords module (golden set item 3).

Run in the build container only (needs /root/reference):   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_addcoords.py

Source: models.modules.AddCoords (reference src/models/modules.py:65-96).  Its constructor calls .cuda() on the coordinate
buffers (modules.py:69-70), so it is constructed with Tensor.cuda patched to a no-op; the buffers are hard-coded to
256 x 256.  Stored: the two appended channels in full for one image (they are batch-independent), and a pass-through
check of the data channels.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from models import modules as ref_modules
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        m = ref_modules.AddCoords()
        x = torch.randn(2, 5, 256, 256, generator=torch.Generator().manual_seed(11))
        y = m(x)
    finally:
        torch.Tensor.cuda = orig_cuda
    assert tuple(y.shape) == (2, 7, 256, 256)
    assert torch.equal(y[:, :5], x), "data channels pass through"
    assert torch.equal(y[0, 5:], y[1, 5:]), "coordinate channels are batch-independent"
    np.savez_compressed(os.path.join(OUT, "addcoords.npz"), coord_channels=y[0, 5:].numpy(),
                        first_varies_along=np.array("H" if float((y[0, 5, 1, 0] - y[0, 5, 0, 0]).abs()) > 0 else "W"))
    print("wrote addcoords.npz", y[0, 5, :2, :2].tolist(), y[0, 6, :2, :2].tolist())


if __name__ == "__main__":
    main()
