"""On-disk frame-triplet feed in the reference's own Cityscapes layout.

Restates reference src/folder.py:14-46 (make_dataset) and :85-104 (DatasetFolder.__getitem__) without cv2 /
torchvision (absent here; PIL decodes the same PNGs):

    <root>/deeplab256_label/<city>/<city>_<snippet6>_<frame6>_gtFine_myseg_id.png     class-id maps, 0..19(20)
    <root>/leftImg256/<city>/<city>_<snippet6>_<frame6>_leftImg8bit.png               RGB frames

A sample is three frames 3 apart, (i, i+3, i+6), taken from every run of consecutive frame numbers of a snippet, for
i in range(first, last - 6) - the reference's bound, which stops one triplet short of the run's end (folder.py:33-34);
kept as is.  Cities and snippets are visited in sorted order (the reference iterates os.listdir / a set, i.e. in
no defined order; only the ORDER differs, and a sampler shuffles it anyway).

__getitem__ returns what folder.py:97-104 returns after transforms.ToTensor (data.py:33-36):
frames float32 (3,H,W) in [0,1]; seg1, seg2 float32 (1,H,W) holding ids; seg3 int64 (H,W).  Segmentation maps are
resized with nearest neighbour to the frame size when they differ (folder.py:133 resizes them to 256 x 256).
There is no dataset in this environment: tests build a miniature tree with PIL.
"""
from __future__ import annotations

import os
import threading
import queue
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from .data import ClipLoader

SEG_DIR, IMG_DIR = "deeplab256_label", "leftImg256"
SEG_SUFFIX, IMG_SUFFIX = "_gtFine_myseg_id.png", "_leftImg8bit.png"
FRAME_KEYS = ("frame1", "seg1", "frame2", "seg2", "frame3", "seg3")


def is_dataset_root(path: Optional[str]) -> bool:
    return bool(path) and os.path.isdir(os.path.join(os.path.expanduser(path), SEG_DIR)) and \
        os.path.isdir(os.path.join(os.path.expanduser(path), IMG_DIR))


def _consecutive_runs(numbers: List[int]) -> List[Tuple[int, int]]:
    """(first, last) of every maximal run of consecutive integers in a sorted list."""
    runs: List[Tuple[int, int]] = []
    for n in numbers:
        if runs and n == runs[-1][1] + 1:
            runs[-1] = (runs[-1][0], n)
        elif not runs or n != runs[-1][1]:
            runs.append((n, n))
    return runs


def make_dataset(root: str) -> List[Tuple[List[str], List[str]]]:
    """[(three segmentation paths, three frame paths)], the enumeration of reference src/folder.py:14-46."""
    root = os.path.expanduser(root)
    seg_root, img_root = os.path.join(root, SEG_DIR), os.path.join(root, IMG_DIR)
    samples: List[Tuple[List[str], List[str]]] = []
    for city in sorted(os.listdir(seg_root)):
        if not os.path.isdir(os.path.join(seg_root, city)):
            continue
        by_snippet: Dict[int, List[int]] = {}
        for name in os.listdir(os.path.join(seg_root, city)):
            if name.endswith(".png"):
                parts = name.split("_")                                    # <city>_<snippet>_<frame>_gtFine_myseg_id.png
                by_snippet.setdefault(int(parts[1]), []).append(int(parts[2]))
        for snippet in sorted(by_snippet):
            stem = "%s_%06d" % (city, snippet)
            for first, last in _consecutive_runs(sorted(by_snippet[snippet])):
                for i in range(first, last - 6):                            # the reference's bound (folder.py:33-34), kept
                    names = [os.path.join(city, "%s_%06d" % (stem, i + 3 * j)) for j in range(3)]
                    samples.append(([os.path.join(seg_root, n + SEG_SUFFIX) for n in names],
                                    [os.path.join(img_root, n + IMG_SUFFIX) for n in names]))
    return samples


def _load_rgb(path: str) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div_(255.0)         # ToTensor: HWC uint8 -> CHW [0,1]


def resize_nearest_cv2(a: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(a, dsize, interpolation=cv2.INTER_NEAREST) (reference src/folder.py:133, src/trainer.py:439-440) by its
    published index rule - destination pixel (y, x) takes source (min(floor(y * H/h), H-1), min(floor(x * W/w), W-1)) -
    which is NOT PIL's NEAREST (that samples at pixel centres, floor((x + 0.5) * W/w)).  cv2 is absent here, so this
    restatement is unpinned against cv2 itself; an 8x downscale picks rows 0, 8, 16 ... as OpenCV's resizeNN does.
    The scale is formed as OpenCV forms it - ify = 1.0 / (dst / src) in double precision, not src / dst - so that
    non-dyadic ratios round the same way."""
    H, W = a.shape[:2]
    h, w = size
    ify, ifx = 1.0 / (float(h) / float(H)), 1.0 / (float(w) / float(W))
    ys = np.minimum(np.floor(np.arange(h, dtype=np.float64) * ify).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w, dtype=np.float64) * ifx).astype(np.int64), W - 1)
    return a[ys][:, xs]


def _load_seg(path: str, size: Tuple[int, int]) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("L"), dtype=np.uint8)
    if a.shape != tuple(size):
        a = resize_nearest_cv2(a, size)                                            # folder.py:133 (INTER_NEAREST)
    return torch.from_numpy(a.copy())


class TripletFolder:
    """Dataset over make_dataset(root); item = dict of FRAME_KEYS (reference src/folder.py:85-104)."""

    def __init__(self, root: str):
        self.root = root
        self.samples = make_dataset(root)
        if not self.samples:
            raise RuntimeError("Found 0 frame triplets under %s (expected %s/ and %s/ with <city> folders)"
                               % (root, SEG_DIR, IMG_DIR))                         # folder.py:72-74

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, index: int) -> Dict[str, torch.Tensor]:
        seg_paths, img_paths = self.samples[index]
        imgs = [_load_rgb(p) for p in img_paths]
        hw = tuple(imgs[0].shape[1:])
        segs = [_load_seg(p, hw) for p in seg_paths]
        return {"frame1": imgs[0], "frame2": imgs[1], "frame3": imgs[2],
                "seg1": segs[0].float().unsqueeze(0), "seg2": segs[1].float().unsqueeze(0), "seg3": segs[2].long()}


class TripletLoader(ClipLoader):
    """ClipLoader semantics (DistributedSampler sharding, set_epoch reshuffle, ragged tail dropped) over a TripletFolder.
    A background thread decodes the next batch while the GPU works on the current one (the reference used
    DataLoader workers, src/trainer.py:149-152)."""

    def __init__(self, folder: TripletFolder, batch: int, rank: int = 0, world: int = 1, seed: int = 1024,
                 shuffle: bool = True, device: Optional[torch.device] = None, prefetch: int = 2):
        self.folder, self.batch, self.rank, self.world = folder, batch, rank, world
        self.seed, self.shuffle, self.device, self.epoch = seed, shuffle, device, 0
        self.keys, self.n, self.prefetch = FRAME_KEYS, len(folder), prefetch

    def _collate(self, items: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        b = {k: torch.stack([it[k] for it in items]).contiguous() for k in FRAME_KEYS}
        if self.device is not None:
            b = {k: v.to(self.device, non_blocking=True) for k, v in b.items()}
        return b

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx = self._indices()
        batches = [idx[i * self.batch:(i + 1) * self.batch] for i in range(len(self))]
        if self.prefetch <= 0:
            for ids in batches:
                yield self._collate([self.folder[i] for i in ids])
            return
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                for ids in batches:
                    if not put(("ok", [self.folder[i] for i in ids])):           # decode off the training thread
                        return
                put(("end", None))
            except Exception as e:                                                # surfaced in the consumer
                put(("err", e))

        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                tag, items = q.get()
                if tag == "end":
                    break
                if tag == "err":
                    raise items
                yield self._collate(items)
        finally:                                                                  # also when the consumer stops early
            stop.set()
            t.join()
