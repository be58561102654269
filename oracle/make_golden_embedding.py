#!/usr/bin/env python
"""Generates tests/golden/embedding_simple.npz from the ONLY embedding the reference contains: the per-pixel class-id
table of models.simple.Simple (reference src/models/simple.py:23 `torch.nn.Embedding(num_embeddings=30, embedding_dim)`,
used at :41-42 `seg[mask] = self.n_classes; x2 = self.embedding(seg)`), i.e. "classes + 1 reserved id" row-gather
semantics - what BASELINE.json's object-slot embedding honours for its class table (last paragraph).

Build container only (needs /root/reference):   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_embedding.py
The module's constructor calls .cuda() on a constant (simple.py:19); Tensor.cuda is patched to a no-op for the
construction, exactly as oracle/make_golden.py does for CoordGridNet.  Stored: ids before / after
the reserved-id rule, mask, the table, the gathered rows, and the table gradient under sum(out * r) (duplicate ids
accumulate)."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        import models.simple as ref_simple
        torch.manual_seed(7)
        m = ref_simple.Simple(n_classes=29, embedding_dim=64, model_name="encoder_decoder")     # simple.py:14-28
    finally:
        torch.Tensor.cuda = orig_cuda
    g = torch.Generator().manual_seed(11)
    seg_gt = torch.randint(0, 29, (2, 6, 7), generator=g)
    mask = torch.rand(2, 6, 7, generator=g) < 0.25
    with torch.no_grad():                                       # simple.py:39-41, verbatim
        seg = seg_gt.clone().long()
        seg[mask] = m.n_classes
    x2 = m.embedding(seg)                                       # simple.py:42
    r = torch.randn(x2.shape, generator=g)
    (x2 * r).sum().backward()
    np.savez_compressed(os.path.join(OUT, "embedding_simple.npz"), seg_gt=seg_gt.numpy(), mask=mask.numpy(), ids=seg.numpy(),
                        table=m.embedding.weight.detach().numpy(), out=x2.detach().numpy(), r=r.numpy(),
                        dtable=m.embedding.weight.grad.numpy())
    print("written", os.path.join(OUT, "embedding_simple.npz"), tuple(x2.shape))


if __name__ == "__main__":
    main()
