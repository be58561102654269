"""Test doubles.  OracleEngine has LayoutEngine's interface but computes with the CPU oracle, so
the host logic around the kernels (Trainer loop, checkpoints, sharding, gradient buckets over
gloo) can be exercised in a container without a GPU.  It lives under tests/ on purpose: the
product has no CPU path."""
import argparse
import logging
import math

import torch

from oracle import layout_spec as O
from vlg.spec import ADAM_BETA1, ADAM_BETA2, ADAM_EPS, ADAM_LR, LayoutConfig, param_layout, param_shapes


class OracleEngine:
    def __init__(self, cfg: LayoutConfig, seed: int = 1024, lr: float = ADAM_LR, beta1: float = ADAM_BETA1):
        self.cfg, self.device = cfg, torch.device("cpu")
        self.lr, self.beta1 = lr, beta1
        self.layout, self.n_params = param_layout(cfg)
        self.params = torch.zeros(self.n_params)
        self.grads_ext = torch.zeros(self.n_params + 4)
        self.grads = self.grads_ext[:self.n_params]
        self.loss_out = self.grads_ext[self.n_params:]
        self.exp_avg = torch.zeros(self.n_params)
        self.exp_avg_sq = torch.zeros(self.n_params)
        self.step_count = 0
        self.load_params(O.init_params(param_shapes(cfg), seed=seed))

    def _view(self, flat, name):
        off, shape = self.layout[name]
        return flat[off:off + math.prod(shape)].view(shape)

    def named_params(self):
        return {n: self._view(self.params, n) for n in self.layout}

    def named_grads(self):
        return {n: self._view(self.grads, n) for n in self.layout}

    def load_params(self, tensors):
        for n in self.layout:
            self._view(self.params, n).copy_(tensors[n])

    def optimizer_state(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.step_count,
                "lr": self.lr, "beta1": self.beta1}

    def load_optimizer(self, st):
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        self.step_count = int(st["step"])

    def forward(self, batch):
        p = self.named_params()
        self._logits, self._raw = O.forward(p, batch["slot_class"], batch["slot_box"], self.cfg.n_layers)
        parts = O.losses(self._logits, self._raw, batch["tgt_class"], batch["tgt_box"], batch["valid"])
        self.loss_out.copy_(torch.stack([x.detach() for x in parts]))
        return self.loss_out

    def outputs_btn(self):
        return self._logits.detach(), self._raw.detach()

    def forward_backward(self, batch, reducer=None):
        parts, grads = O.loss_and_grads(self.named_params(), batch, self.cfg.n_layers)
        self.loss_out.copy_(torch.tensor(parts))
        for n, g in grads.items():
            self._view(self.grads, n).copy_(g)
        if reducer is not None:                          # same completion order as LayoutEngine.backward
            reducer.ready("head")
            for l in reversed(range(self.cfg.n_layers)):
                reducer.ready("l%d" % l)
            reducer.ready("embed")
        return self.loss_out

    def adam_step(self, grad_scale: float = 1.0, lo: int = 0, hi=None, advance: bool = True):
        hi = self.n_params if hi is None else hi
        if advance:
            self.step_count += 1
        O.adam_step(self.params[lo:hi], self.grads[lo:hi] * grad_scale, self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                    self.step_count, lr=self.lr, beta1=self.beta1, beta2=ADAM_BETA2, eps=ADAM_EPS)

    def train_step(self, batch, reducer=None):
        """Same schedule as LayoutEngine.train_step: everything but the embedding range is updated while the last
        bucket's all-reduce is still in flight."""
        loss = self.forward_backward(batch, reducer)
        if reducer is not None:
            split = self.layout["l0.ln1_g"][0]
            reducer.wait(keep=("embed",))
            self.adam_step(reducer.grad_scale, lo=split, hi=self.n_params)
            reducer.wait()
            self.adam_step(reducer.grad_scale, lo=0, hi=split, advance=False)
        else:
            self.adam_step()
        return loss


def oracle_factory(cfg, args):
    return OracleEngine(cfg, seed=int(args.seed), lr=float(args.lr), beta1=float(args.beta1))


def reference_args(path, rank=0, gpus=1, **over):
    """Namespace with the fields reference src/main.py:86-160 parses (defaults from there) plus the ones
    main.worker injects (main.py:51-52,164,173,183)."""
    logger = logging.getLogger("vlg-test-%d" % rank)
    logger.setLevel(logging.DEBUG)
    a = argparse.Namespace(
        dataset="cityscape", train_dir="/data/agong/train", val_dir="/data/agong/val", test_dir="/data/agong/test",
        validate=False, edge=False, val_interval=1, arch="CoordGridNet", discriminator="NLayerDiscriminator",
        generator="ResnetGenerator", batch_size=32, epochs=10, resume=None, img1=None, img2=None, seg1=None,
        seg2=None, workers=4, port=None, seed=1024, print_freq=10, path=str(path), ckpt=None, start_epoch=1,
        disp_interval=10, optimizer="adamax", lr=0.0002, beta1=0.5, lr_decay_step=5, lr_decay_gamma=0.1,
        input_nc=8, output_nc=3, ngf=64, ndf=64, netD="basic", netG="resnet_9blocks", n_layers_D=3,
        norm="instance", init_type="normal", init_gain=0.02, no_dropout=False, gan_mode="lsgan",
        logger=logger, rank=rank, gpus=gpus)
    for k, v in over.items():
        setattr(a, k, v)
    return a
