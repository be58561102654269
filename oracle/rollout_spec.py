"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's autoregressive rollout (pixel model).

Follows reference src/trainer.py:453-476 (generate_sequence) line by line with torch-CPU ops on top of
oracle/gridnet_spec.py and oracle/hned_spec.py (both bit-identical to the reference modules on the committed
fixtures), with SURVEY.md Appendix A-10 repaired the way vlg/image_engine.py:FrameRollout states it: the
10-channel input gets the two edge channels the network was trained with (trainer.py:190-197), computed as
trainer.py:214-216 computes the edge map of a prediction.  PARITY UNPINNED as a whole (the reference method cannot
run: 8-channel input into a 10-channel net, undefined self.netG); its pieces are pinned.  Only tests/ import this.
"""
import torch

from oracle import gridnet_spec as G
from oracle import hned_spec as HS

IMG_MEAN = torch.tensor([0.485, 0.456, 0.406])[None, :, None, None]      # trainer.py:123
IMG_STD = torch.tensor([0.229, 0.224, 0.225])[None, :, None, None]       # trainer.py:122
MEAN_ARR = torch.tensor([-0.03, -0.088, -0.188])[None, :, None, None]    # trainer.py:120
STD_ARR = torch.tensor([0.448, 0.448, 0.450])[None, :, None, None]       # trainer.py:121


def edge(hed_params, img_normalised):
    """e = hed(img * img_std + img_mean)[5]   (trainer.py:214-216, fused map per Appendix A-3)"""
    return HS.forward(hed_params, img_normalised * IMG_STD + IMG_MEAN)[5]


def step(p, hed_params, coord, img_a, img_b, seg_a, seg_b):
    """One iteration of trainer.py:460-469.  Returns (seg logits, normalised img_next, seg_next ids as float)."""
    x = torch.cat([edge(hed_params, img_a), seg_a, img_a, img_b, seg_b, edge(hed_params, img_b)], dim=1)   # :461 (+ A-10)
    seg_logits, img_next = G.forward(p, x, coord)                                                          # :464
    img_next = (img_next - MEAN_ARR) / STD_ARR                                                             # :466
    seg_next = torch.argmax(seg_logits, dim=1).unsqueeze_(1).float()                                       # :467
    return seg_logits, img_next, seg_next


def generate_sequence(p, hed_params, coord, img1, img2, seg1, seg2, steps=8):
    img, seg = [img1, img2], [seg1, seg2]                                   # :454-458
    with torch.no_grad():
        for _ in range(steps):                                               # :460
            _, img_next, seg_next = step(p, hed_params, coord, img[-2], img[-1], seg[-2], seg[-1])
            img.append(img_next)                                             # :468
            seg.append(seg_next)                                             # :469
    return torch.cat(img, dim=1), torch.cat(seg, dim=1)                      # :470-471
