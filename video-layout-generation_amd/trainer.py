"""Drop-in `trainer` module: same surface the reference's src/main.py consumes.

    from trainer import Trainer            # reference src/main.py:14
    trainer = Trainer(args)                # reference src/main.py:62   (after init_process_group + seeding)
    trainer.set_epoch(epoch); trainer.train(); metrics = trainer.validate()
    trainer.save_checkpoint(metrics)       # reference src/main.py:76-82

`args` is the argparse Namespace of reference src/main.py:86-160 plus the fields main.worker
injects (`logger`, `rank`, `gpus`, `path`, `port`; main.py:51-52,164,173,183).  The step behind
it is the MI355X-native layout-token step (vlg.engine.LayoutEngine -> libvlg_hip.so); the
reference's own step is a 2-D CNN that BASELINE.json does not ask for.

Shape knobs the reference CLI does not have are read with getattr/env defaults so main.py stays
byte-for-byte unchanged:  VLG_FRAMES (T, 16), VLG_SLOTS (N, 64), VLG_DMODEL (d, 256),
VLG_LAYERS (4), VLG_TRAIN_CLIPS (1024), VLG_VAL_CLIPS (256), VLG_VARIABLE_N (0), VLG_PRECISION (fp32 | fp32x3 | bf16 |
bf16_mfma: projection arithmetic / activation storage, vlg/engine.py).

VLG_MODEL=gridnet switches the step to the reference's OWN model and losses (vlg/image_engine.py;
VLG_WITH_HED / VLG_WITH_VGG = 1 add the frozen edge net and the VGG19 term, VLG_HED_CKPT / VLG_VGG_CKPT their weights;
--train_dir / --val_dir in the reference's Cityscapes layout are read by vlg/cityscapes.py, else frames are synthetic:
--arch GridNet|CoordGridNet on the conv3x3 MFMA kernels, 40 L1 + 20 (GradientLoss + SSIM) + 10 CE, reference
src/trainer.py:193-258) on synthetic frame triplets of VLG_IMG_SIZE (256) pixels; the default (layout) is the
token step BASELINE.json's metric is quoted on.

Repairs of reference defects, all stated: gradients are overwritten each
step (A-5 zero_grad), the train log line uses the `loss` key (A-6), one checkpoint schema
{'epoch','arch','gridnet','optimizer'} for save/--ckpt/--resume (A-1,A-8,A-9), `.model` exists
(A-11), validate() moves every input to the device (A-13) and all-reduces a size-weighted SUM then
divides by the GLOBAL clip count (the reference divides by the local count, trainer.py:336-339,
which returns world x the mean), visualisation is not run every step (A-15).
"""
from __future__ import annotations

import os
import random
import shutil
from time import time
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from vlg.data import BATCH_KEYS, BucketedClipLoader, ClipLoader, synthetic_clips, to_device
from vlg.dp import GradReducer, bucket_ranges
from vlg.spec import ADAM_BETA1, ADAM_LR, SEED, LayoutConfig


class AverageMeter(object):
    """Running weighted mean; same arithmetic as reference src/utils.py:1-16."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class _ScalarLog:
    """SummaryWriter stand-in (tensorboardX is not installed here): same add_scalar calls as
    reference src/trainer.py:166,281,377, appended to <path>/scalars.tsv."""

    def __init__(self, path: str):
        os.makedirs(path, exist_ok=True)
        self.f = open(os.path.join(path, "scalars.tsv"), "a")

    def add_scalar(self, tag, value, step):
        self.f.write("%s\t%d\t%.8g\n" % (tag, int(step), float(value)))
        self.f.flush()

    def add_image(self, *a, **k):      # image grids (trainer.py:282-286) are out of scope
        pass


def _make_writer(path: str):
    try:
        from tensorboardX import SummaryWriter      # reference src/trainer.py:17,141-142
        return SummaryWriter(path)
    except Exception:
        return _ScalarLog(path)


def _knob(args, name: str, env: str, default: int) -> int:
    v = getattr(args, name, None)
    return int(v) if v is not None else int(os.environ.get(env, default))


def layout_config(args) -> LayoutConfig:
    """Per-GPU batch = batch_size // gpus, as reference src/trainer.py:148 sizes its loaders."""
    gpus = max(int(getattr(args, "gpus", 1) or 1), 1)
    return LayoutConfig(B=max(int(args.batch_size) // gpus, 1), T=_knob(args, "n_frames", "VLG_FRAMES", 16),
                        N=_knob(args, "n_slots", "VLG_SLOTS", 64), d=_knob(args, "d_model", "VLG_DMODEL", 256),
                        n_layers=_knob(args, "n_layers", "VLG_LAYERS", 4),
                        # "slot" (default): causal attention along T per slot; "clip": block-causal over all slots of a clip
                        attention=str(getattr(args, "attention", None) or os.environ.get("VLG_ATTENTION", "slot")))


def get_layout_engine(args, cfg: Optional[LayoutConfig] = None, engine_factory: Optional[Callable] = None):
    """Model + optimiser state in one object (counterpart of get_gridnet, reference src/trainer.py:81-94:
    build on args.rank's device, Adam(lr=args.lr, betas=(args.beta1, 0.999)), optional --ckpt load)."""
    cfg = cfg or layout_config(args)
    if engine_factory is None:
        from vlg.engine import LayoutEngine          # HIP only; raises without a GPU / without the .so
        device = torch.device("cuda", int(args.rank))
        precision = str(getattr(args, "precision", None) or os.environ.get("VLG_PRECISION", "fp32"))
        engine = LayoutEngine(cfg, device, seed=int(getattr(args, "seed", SEED)),
                              lr=float(getattr(args, "lr", ADAM_LR)), beta1=float(getattr(args, "beta1", ADAM_BETA1)),
                              precision=precision,
                              padded_slots=bool(_knob(args, "variable_n", "VLG_VARIABLE_N", 0)))   # fixed-N feeds hold no padded slots
    else:
        engine = engine_factory(cfg, args)
    ckpt_path = getattr(args, "ckpt", None)
    if ckpt_path is not None:
        args.logger.info("Loading from ckpt %s" % ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=True)
        if "gridnet" in ckpt:
            engine.load_params(ckpt["gridnet"])
        if "optimizer" in ckpt:
            engine.load_optimizer(ckpt["optimizer"])
    return engine


get_gridnet = get_layout_engine     # reference name (src/trainer.py:81)


def _load_frozen(net, env: str, args) -> None:
    """Weights of a frozen net (HED / VGG19).  `env` names a state_dict file in the reference's / torchvision's key
    format; the authors' HED checkpoint keeps it under 'generator' (reference src/trainer.py:99), other tools under
    'state_dict'.  Without a file the net gets torch's default conv initialisation from the shared seed (identical on
    every rank) - announced at WARNING level, because the edge maps / perceptual term then mean nothing."""
    path = os.environ.get(env)
    if path:
        sd_ = torch.load(path, map_location="cpu", weights_only=True)
        for key in ("generator", "state_dict"):
            if isinstance(sd_, dict) and key in sd_ and isinstance(sd_[key], dict):
                sd_ = sd_[key]
                break
        net.load_state_dict(sd_)
        args.logger.info("%s: loaded %s" % (type(net).__name__, path))
        return
    gi = torch.Generator().manual_seed(int(getattr(args, "seed", SEED)) + 17)
    shp_ = net.reference_shapes()
    net.load_state_dict({k: (torch.rand(v, generator=gi) * 2 - 1) / float(max(1, (v[1] * v[2] * v[3]) if len(v) == 4 else 64)) ** 0.5
                         for k, v in shp_.items()})
    args.logger.warning("%s not set: %s runs with RANDOMLY INITIALISED frozen weights (the trained ones are not in the "
                        "reference repository: src/trainer.py:97 is an author-local path, vgg19(pretrained=True) a download)"
                        % (env, type(net).__name__))


class _ImageModel:
    """Adapter giving vlg.image_engine.ImageEngine the few methods Trainer uses on the layout engine."""

    def __init__(self, args, batch: int):
        from vlg.image_engine import IMAGE_KEYS, ImageEngine
        size = _knob(args, "img_size", "VLG_IMG_SIZE", 256)
        self.size = size
        self.args = args
        self.device = torch.device("cuda", int(args.rank))
        arch = args.arch if args.arch in ("GridNet", "CoordGridNet") else "CoordGridNet"
        from vlg.cityscapes import is_dataset_root
        self.on_disk = is_dataset_root(getattr(args, "train_dir", None))        # reference layout (src/folder.py:14-46)
        # the frozen edge net feeds two of the ten input channels (trainer.py:190-197): needed as soon as frames come
        # from files; the VGG19 term of CombinedLoss (loss.py:61-62) is opt-in - neither net's trained weights ship
        # with the reference (trainer.py:97 is an author-local path, vgg19(pretrained=True) a download)
        with_hed = self.on_disk or bool(_knob(args, "with_hed", "VLG_WITH_HED", 0))
        with_vgg = bool(_knob(args, "with_vgg", "VLG_WITH_VGG", 0))
        self.keys = tuple(k for k in IMAGE_KEYS if not (with_hed and k in ("e1", "e2")))   # HED makes the edge maps itself
        self.engine = ImageEngine(batch, size, size, self.device, arch=arch, lr=float(getattr(args, "lr", ADAM_LR)),
                                  beta1=float(getattr(args, "beta1", ADAM_BETA1)), with_hed=with_hed, with_vgg=with_vgg)
        for net, env in ((self.engine.hed, "VLG_HED_CKPT"), (self.engine.vgg, "VLG_VGG_CKPT")):
            if net is not None:
                _load_frozen(net, env, args)
        # torch's default initialisers for Conv2d (U(+-1/sqrt(fan_in)) for weight and bias) and PReLU (0.25), from one
        # generator seeded like every rank's (main.py:57-60) so replicas start identical without a broadcast
        g = torch.Generator().manual_seed(int(getattr(args, "seed", SEED)))
        shapes = self.engine.net.reference_shapes()
        sd = {}
        for k, shp in shapes.items():
            if shp == (1,):
                sd[k] = torch.full(shp, 0.25)
            else:
                fan_in = (shp[1] if len(shp) == 4 else shapes[k[:-len("bias")] + "weight"][1]) * 9
                sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / fan_in ** 0.5
        self.engine.load_state_dict(sd)
        ckpt_path = getattr(args, "ckpt", None)                       # get_gridnet's --ckpt, reference src/trainer.py:85-92
        if ckpt_path is not None:
            args.logger.info("Loading from ckpt %s" % ckpt_path)
            ckpt = torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=True)
            if "gridnet" in ckpt:
                self.load_params(ckpt["gridnet"])                     # (the reference wrote `generator.` here: Appendix A-1)
            if "optimizer" in ckpt:
                self.load_optimizer(ckpt["optimizer"])
        self.world = max(int(getattr(args, "gpus", 1) or 1), 1)
        self.step_count = 0
        self._rollouts = {}
        self._hed_master = None

    def named_params(self):
        return self.engine.state_dict()

    def load_params(self, sd):
        self.engine.load_state_dict(sd)

    def optimizer_state(self):
        return self.engine.optimizer_state()

    def load_optimizer(self, st):
        self.engine.load_optimizer(st)

    def train_step(self, batch, flip: bool, reducer=None):
        return self.engine.train_step(batch, flip, reducer).reshape(1)

    def eval_loss(self, batch):
        self.engine.forward(batch, flip=False, want_grads=False)
        return self.engine.total().reshape(1)

    def _edge_net(self):
        """The frozen HED net whose weights the rollout's twin reads: the training step's own, or (synthetic-frame
        training feeds edge maps as inputs) one built for the purpose."""
        if self.engine.hed is not None:
            return self.engine.hed
        if self._hed_master is None:
            from vlg.hned import HNEDHIP
            self._hed_master = HNEDHIP(1, 16, 16, self.device)        # weights only; twins are sized per call
            _load_frozen(self._hed_master, "VLG_HED_CKPT", self.args)
        return self._hed_master

    def rollout(self, img1, img2, seg1, seg2, steps: int = 8):
        """reference src/trainer.py:453-476 on the HIP nets (vlg.image_engine.FrameRollout)."""
        from vlg.image_engine import FrameRollout
        b, _, H, W = img1.shape
        key = (int(b), int(H), int(W))
        if key not in self._rollouts:
            self._rollouts[key] = FrameRollout(self.engine, self._edge_net(), *key)
        return self._rollouts[key].run(img1, img2, seg1, seg2, steps)


class _ModelHandle:
    """`trainer.model` / `trainer.gridnet`: main.py:65 calls trainer.model.eval()."""

    def __init__(self, engine):
        self.engine, self.training = engine, True

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def state_dict(self):
        return {k: v.detach().cpu().clone() for k, v in self.engine.named_params().items()}

    def load_state_dict(self, sd):
        self.engine.load_params(sd)


class Trainer:

    def __init__(self, args, engine_factory: Optional[Callable] = None):
        args.logger.info("Initializing trainer")
        # reference src/trainer.py:107-108 creates ../predict relative to the working directory (src/); the default is kept,
        # args.predict_dir / VLG_PREDICT_DIR move it (tests, read-only working directories)
        self.predict_dir = str(getattr(args, "predict_dir", None) or os.environ.get("VLG_PREDICT_DIR") or "../predict")
        if not os.path.isdir(self.predict_dir):
            os.makedirs(self.predict_dir, exist_ok=True)
        self.args = args
        self.world = max(int(getattr(args, "gpus", 1) or 1), 1)
        self.distributed = dist.is_available() and dist.is_initialized() and self.world > 1
        if torch.cuda.is_available() and engine_factory is None:
            torch.cuda.set_device(int(args.rank))               # reference src/trainer.py:110
        self.cfg = layout_config(args)
        self.image_mode = engine_factory is None and os.environ.get("VLG_MODEL", "layout").lower() == "gridnet"
        self.reducer = None
        if self.image_mode:
            self.engine = _ImageModel(args, self.cfg.B)
            if self.distributed:                                # replaces DDP(gridnet), trainer.py:113: bucket per grid column
                net = self.engine.engine.net
                self.reducer = GradReducer(net.grads_ext, net.bucket_ranges())
        else:
            self.engine = get_layout_engine(args, self.cfg, engine_factory)
            if self.distributed:                                # replaces the DDP wrappers, trainer.py:113,115
                self.reducer = GradReducer(self.engine.grads_ext,
                                           bucket_ranges(self.engine.layout, self.engine.n_params, self.cfg.n_layers))
        self.device = self.engine.device
        self.gridnet = self.model = _ModelHandle(self.engine)   # reference attr `gridnet`; main.py:65 wants `.model`
        self.global_step = 0
        self.epoch = 0
        if getattr(args, "resume", None) is not None:           # reference src/trainer.py:138-139
            self.load_checkpoint(args.resume)
        self.writer = _make_writer(args.path) if int(args.rank) == 0 else None   # trainer.py:141-142

        seed = int(getattr(args, "seed", SEED))
        variable_n = bool(_knob(args, "variable_n", "VLG_VARIABLE_N", 0))
        n_train = _knob(args, "train_clips", "VLG_TRAIN_CLIPS", 1024)
        n_val = _knob(args, "val_clips", "VLG_VAL_CLIPS", 256)
        common = dict(batch=self.cfg.B, rank=int(args.rank), world=self.world, seed=seed)
        if self.device.type == "cuda":       # device batches, pinned staging + one batch ahead on a side stream (vlg/data.py)
            common["device"] = self.device
        if self.image_mode and self.engine.on_disk:
            # the reference's own data: frame triplets from <train_dir>/{deeplab256_label,leftImg256}/<city>/ (folder.py)
            from vlg.cityscapes import TripletFolder, TripletLoader
            val_dir = getattr(args, "val_dir", None) or args.train_dir
            common["device"] = self.device
            self.train_loader = TripletLoader(TripletFolder(args.train_dir), shuffle=True, **common)
            self.val_loader = TripletLoader(TripletFolder(val_dir), shuffle=False, **common)
            args.logger.debug("Finish init trainer")
            return
        if self.image_mode:
            from vlg.image_engine import synthetic_frames
            sz = self.engine.size
            train_clips, val_clips = synthetic_frames(n_train, sz, sz, seed=seed), synthetic_frames(n_val, sz, sz, seed=seed + 1)
            Loader, common["keys"] = ClipLoader, self.engine.keys
        else:
            mk = dict(T=self.cfg.T, N=self.cfg.N, n_classes=self.cfg.n_classes, variable_n=variable_n)
            train_clips = synthetic_clips(n_train, seed=seed, **mk)      # get_dataset(), trainer.py:144
            val_clips = synthetic_clips(n_val, seed=seed + 1, **mk)
            Loader = BucketedClipLoader if variable_n else ClipLoader
        self.train_loader = Loader(train_clips, shuffle=True, **common)   # DistributedSampler + DataLoader, :145-152
        self.val_loader = Loader(val_clips, shuffle=False, **common)
        args.logger.debug("Finish init trainer")

    # ----------------------------------------------------------------- epoch control
    def set_epoch(self, epoch):
        """0-based in, 1-based stored, samplers reseeded (reference src/trainer.py:158-162)."""
        self.args.logger.info("Start of epoch %d" % (epoch + 1))
        self.epoch = epoch + 1
        self.train_loader.set_epoch(epoch)
        self.val_loader.set_epoch(epoch)

    # ------------------------------------------------------------------------- train
    def _flip(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Horizontal flip of a layout = cx -> 1 - cx on inputs AND targets.  Drawn from the shared-seed
        `random` stream exactly like reference src/trainer.py:200-206, so all ranks flip together."""
        if random.random() < 0.5:
            batch = dict(batch)
            for k in ("slot_box", "tgt_box"):
                b = batch[k].clone()
                b[..., 0] = 1.0 - b[..., 0]
                batch[k] = b
        return batch

    def train(self):
        self.args.logger.info("Training started")
        self.gridnet.train()
        end = time()
        n_batches = len(self.train_loader)
        for i, batch in enumerate(self.train_loader):
            if self.image_mode:                                   # flip is done by vlg_prep_input (trainer.py:200-206)
                flip = random.random() < 0.5
                batch = to_device(batch, self.device, self.engine.keys)
            else:
                batch = to_device(self._flip(batch), self.device)     # H2D, reference src/trainer.py:184-187
            load_time = time() - end
            end = time()
            self.global_step += 1
            # forward, 40/20/10 loss, backward, bucketed all-reduce, Adam: reference src/trainer.py:209-258
            if self.image_mode:
                loss = self.engine.train_step(batch, flip, self.reducer)
            else:
                loss = self.engine.train_step(batch, self.reducer)
            if int(self.args.rank) == 0 and i % int(self.args.print_freq) == 0:
                # with a reducer the loss floats were summed over ranks inside a gradient bucket (the first one of
                # the token step, the last one of the pixel step): logged value = cross-rank mean, as sync() gave the
                # reference (trainer.py:256), without a collective of its own
                value = float(loss[0].item()) / (self.world if self.reducer is not None else 1)
                comp_time = time() - end
                self.args.logger.info(
                    "Epoch [{epoch:d}/{tot_epoch:d}][{cur_batch:d}/{tot_batch:d}] "
                    "load [{load_time:.3f}s] comp [{comp_time:.3f}s] "
                    "loss [{loss:.4f}]".format(epoch=self.epoch, tot_epoch=int(self.args.epochs), cur_batch=i + 1,
                                               tot_batch=n_batches, load_time=load_time, comp_time=comp_time,
                                               loss=value))
                self.writer.add_scalar("train/gen loss GAN", value, self.global_step)   # tag: trainer.py:281
            end = time()

    # ---------------------------------------------------------------------- validate
    def validate(self):
        self.args.logger.info("Validation started")
        self.gridnet.eval()
        val_loss = AverageMeter()
        end = time()
        n_batches = len(self.val_loader)
        for i, batch in enumerate(self.val_loader):
            keys = self.engine.keys if self.image_mode else BATCH_KEYS
            batch = to_device(batch, self.device, keys)
            load_time = time() - end
            end = time()
            if self.image_mode:
                loss = self.engine.eval_loss(batch).clone()
            else:
                loss = self.engine.forward(batch)[0:1].clone()      # forward only, reference src/trainer.py:320-333
            size = torch.tensor([float(batch[keys[0]].shape[0])], device=loss.device)
            loss.mul_(size)
            self.sync([loss, size], mean=False)                     # size-weighted SUM, trainer.py:336-338
            loss.div_(size)                                         # / GLOBAL clip count (see module docstring)
            val_loss.update(loss.item(), size.item())
            comp_time = time() - end
            end = time()
            if int(self.args.rank) == 0 and i % int(self.args.print_freq) == 0:
                self.args.logger.info(
                    "Epoch [{epoch:d}/{tot_epoch:d}][{cur_batch:d}/{tot_batch:d}] "
                    "load [{load_time:.3f}s] comp [{comp_time:.3f}s]".format(
                        epoch=self.epoch, tot_epoch=int(self.args.epochs), cur_batch=i + 1, tot_batch=n_batches,
                        load_time=load_time, comp_time=comp_time))
        if int(self.args.rank) == 0:
            self.args.logger.info("Epoch [{epoch:d}/{tot_epoch:d}] loss [{loss:.4f}] ".format(
                epoch=self.epoch, tot_epoch=int(self.args.epochs), loss=val_loss.avg))
            self.writer.add_scalar("val/loss", val_loss.avg, self.epoch)                # trainer.py:377
        return {"loss": val_loss.avg}                                                   # trainer.py:379

    def sync(self, tensors, mean=True):
        """Synchronize all tensors given using mean or sum (reference src/trainer.py:381-386)."""
        if not self.distributed:
            return
        for tensor in tensors:
            dist.all_reduce(tensor)
            if mean:
                tensor.div_(self.world)

    # ------------------------------------------------------------------- checkpoints
    def save_checkpoint(self, metrics):
        """Rank 0 only (reference src/main.py:81-82): ../checkpoint/%03d.pth + latest.pth
        (reference src/trainer.py:390-402), schema {'epoch','arch','gridnet','optimizer'}."""
        self.args.logger.info("Saving checkpoint..")
        prefix = "../checkpoint"
        os.makedirs(prefix, exist_ok=True)
        torch.save({"epoch": self.epoch, "arch": self.args.arch, "gridnet": self.gridnet.state_dict(),
                    "optimizer": self.engine.optimizer_state(), "metrics": dict(metrics or {})},
                   "%s/%03d.pth" % (prefix, self.epoch))
        shutil.copy("%s/%03d.pth" % (prefix, self.epoch), "%s/latest.pth" % prefix)

    def load_checkpoint(self, resume):
        """--resume (reference src/trainer.py:404-414): arch must match; restores epoch, weights, Adam state."""
        self.args.logger.info("Resuming checkpoint %s" % resume)
        ckpt = torch.load(resume, map_location=torch.device("cpu"), weights_only=True)
        assert ckpt["arch"] == self.args.arch, ("Architecture mismatch: ckpt %s, config %s"
                                                % (ckpt["arch"], self.args.arch))
        self.epoch = ckpt["epoch"]
        self.gridnet.load_state_dict(ckpt["gridnet"])
        self.engine.load_optimizer(ckpt["optimizer"])
        self.args.logger.info("Checkpoint loaded")

    # ----------------------------------------------------------------------- rollout
    def generate_sequence(self, *inputs, steps: int = 8):
        """Autoregressive rollout, `steps` predictions (8 in reference src/trainer.py:460).

        VLG_MODEL=gridnet - the reference's own method: generate_sequence(img1, img2, seg1, seg2) with two
        ImageNet-normalised frames (b,3,H,W) and two segmentation-id maps (b,1,H,W) (what eval_generate_sequence passes,
        trainer.py:440-451).  Runs trainer.py:453-476 on the HIP nets (Appendix A-10 repaired: vlg/image_engine.py
        FrameRollout), writes ../predict/val_<time>_{img,seg}.npy like trainer.py:474-476 and ALSO returns the two
        arrays (b, 3*(steps+2), H, W), (b, steps+2, H, W).

        Layout mode: generate_sequence(slot_class (B,T,N), slot_box (B,T,N,4)): predict the frame after the clip,
        append it, slide the T-frame window; returns classes (B,steps,N) and boxes (B,steps,N,4) on the CPU."""
        if self.image_mode:
            if len(inputs) != 4:
                raise TypeError("generate_sequence(img1, img2, seg1, seg2) in VLG_MODEL=gridnet mode")
            import numpy as np
            p, q = self.engine.rollout(*inputs, steps=steps)
            p, q = p.cpu().numpy(), q.cpu().numpy()
            t = time()
            os.makedirs(self.predict_dir, exist_ok=True)
            np.save(os.path.join(self.predict_dir, "val_" + str(t) + "_img.npy"), p)     # trainer.py:474-476
            np.save(os.path.join(self.predict_dir, "val_" + str(t) + "_seg.npy"), q)
            return p, q
        if len(inputs) != 2:
            raise TypeError("generate_sequence(slot_class, slot_box) in layout mode")
        slot_class, slot_box = inputs
        cls = slot_class.clone().to(self.device)
        box = slot_box.clone().to(self.device)
        B, T, N = cls.shape
        out_c, out_b = [], []
        dummy = {"tgt_class": torch.zeros_like(cls), "tgt_box": torch.zeros_like(box)}
        for _ in range(steps):
            # valid: padded slots carry the reserved class id (vlg/data.py) - with attention = "clip" they must not be attended
            # to; the loss the forward also evaluates against the dummy targets is not used
            valid = (cls < self.cfg.n_classes).to(torch.float32).contiguous()
            self.engine.forward(dict(dummy, slot_class=cls.contiguous(), slot_box=box.contiguous(), valid=valid))
            logits, raw = self.engine.outputs_btn()
            nc = torch.argmax(logits[:, -1], dim=-1)                 # (B,N), as trainer.py:467
            nb = torch.sigmoid(raw[:, -1])
            out_c.append(nc.cpu())
            out_b.append(nb.cpu())
            cls = torch.cat([cls[:, 1:], nc[:, None]], dim=1)
            box = torch.cat([box[:, 1:], nb[:, None]], dim=1)
        return torch.stack(out_c, dim=1), torch.stack(out_b, dim=1)

    def eval_generate_sequence(self, img1, img2, seg1, seg2):
        """main.py:64-67 entry (reference src/trainer.py:429-451): read two frames and two segmentation-id maps from
        files, resize the maps to 256 x 256 nearest-neighbour (:439-440; cv2.INTER_NEAREST index rule, vlg/cityscapes.py),
        ToTensor + ImageNet-normalise the frames (:443-447), then generate_sequence.  PIL decodes the files (cv2 is not
        installed here).  An unreadable path logs and returns like :436-438.  A layout-token model has no pixel
        inputs: there the method logs an error and returns None, so that `main.py --img1 ...` (main.py:64-67) ends the
        way the reference's entry does for unusable inputs - a log line, not a traceback."""
        if not self.image_mode:
            self.args.logger.error("eval_generate_sequence reads image files: run with VLG_MODEL=gridnet "
                                   "(layout mode rolls out with generate_sequence(slot_class, slot_box))")
            return None
        from vlg.cityscapes import _load_rgb, _load_seg
        from vlg.spec import IMG_MEAN, IMG_STD
        for pth in (img1, img2, seg1, seg2):
            if not (isinstance(pth, str) and os.path.isfile(pth)):
                self.args.logger.debug("path name not exists")                    # trainer.py:436-438
                return None
        size = self.engine.size
        frames = [_load_rgb(pth) for pth in (img1, img2)]
        segs = [_load_seg(pth, (size, size)).float()[None, None] for pth in (seg1, seg2)]
        for f in frames:
            if tuple(f.shape[1:]) != (size, size):
                raise ValueError("frames must be %d x %d like the reference's pre-scaled leftImg256 (got %s)"
                                 % (size, size, tuple(f.shape[1:])))
        mean = torch.tensor(IMG_MEAN)[:, None, None]
        std = torch.tensor(IMG_STD)[:, None, None]
        frames = [((f - mean) / std)[None] for f in frames]                       # transforms.Normalize, :443-447
        self.args.logger.debug(segs[0].shape)
        self.args.logger.debug(frames[0].shape)
        return self.generate_sequence(frames[0], frames[1], segs[0], segs[1])
