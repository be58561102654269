"""The on-disk feed in the reference's Cityscapes layout (vlg/cityscapes.py) against the semantics of reference
src/folder.py:14-46,85-104, on a miniature tree written with PIL (there is no dataset in this environment)."""
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def write_tree(root, city, snippet, frames, hw=(8, 12), seg_hw=None, seed=0):
    rng = np.random.RandomState(seed)
    seg_dir, img_dir = os.path.join(root, "deeplab256_label", city), os.path.join(root, "leftImg256", city)
    os.makedirs(seg_dir, exist_ok=True)
    os.makedirs(img_dir, exist_ok=True)
    out = {}
    for f in frames:
        stem = "%s_%06d_%06d" % (city, snippet, f)
        img = rng.randint(0, 256, size=hw + (3,), dtype=np.uint8)
        seg = rng.randint(0, 20, size=seg_hw or hw, dtype=np.uint8)
        Image.fromarray(img, "RGB").save(os.path.join(img_dir, stem + "_leftImg8bit.png"))
        Image.fromarray(seg, "L").save(os.path.join(seg_dir, stem + "_gtFine_myseg_id.png"))
        out[f] = (img, seg)
    return out


def test_make_dataset_enumerates_triplets_like_the_reference(tmp_path):
    from vlg.cityscapes import is_dataset_root, make_dataset
    root = str(tmp_path / "train")
    # snippet 7 of 'aachen': frames 0..11 and, after a gap, 20..29; snippet 9: too short for any triplet
    write_tree(root, "aachen", 7, list(range(0, 12)) + list(range(20, 30)))
    write_tree(root, "aachen", 9, list(range(0, 6)))
    write_tree(root, "bonn", 1, list(range(3, 11)))
    assert is_dataset_root(root) and not is_dataset_root(str(tmp_path))
    samples = make_dataset(root)
    got = [tuple(os.path.basename(p)[:-len("_leftImg8bit.png")] for p in imgs) for _, imgs in samples]
    # folder.py:33-34: for every run r of consecutive frames, i in range(r[0], r[-1] - 6) -> (i, i+3, i+6)
    want = []
    for city, snip, runs in (("aachen", 7, [(0, 11), (20, 29)]), ("aachen", 9, [(0, 5)]), ("bonn", 1, [(3, 10)])):
        for first, last in runs:
            for i in range(first, last - 6):
                want.append(tuple("%s_%06d_%06d" % (city, snip, i + 3 * j) for j in range(3)))
    assert got == want and len(got) == 5 + 3 + 0 + 1
    for segs, imgs in samples:                      # every path exists and the two lists name the same frames
        assert all(os.path.exists(p) for p in segs + imgs)
        assert [os.path.basename(p).split("_gtFine")[0] for p in segs] == [os.path.basename(p).split("_leftImg")[0] for p in imgs]


def test_items_follow_folder_getitem(tmp_path):
    from vlg.cityscapes import TripletFolder
    root = str(tmp_path / "d")
    data = write_tree(root, "ulm", 3, list(range(0, 8)), hw=(8, 12))
    ds = TripletFolder(root)
    assert len(ds) == 1                            # range(0, 7 - 6) = {0}: frames (0, 3, 6)
    it = ds[0]
    for j, f in enumerate((0, 3, 6)):
        img, seg = data[f]
        fr = it["frame%d" % (j + 1)]
        assert fr.dtype == torch.float32 and tuple(fr.shape) == (3, 8, 12)
        assert torch.equal(fr, torch.from_numpy(img).permute(2, 0, 1).float() / 255.0)     # transforms.ToTensor()
        sg = it["seg%d" % (j + 1)]
        if j < 2:                                  # folder.py:99-100: float, channel dim
            assert sg.dtype == torch.float32 and tuple(sg.shape) == (1, 8, 12) and torch.equal(sg[0], torch.from_numpy(seg).float())
        else:                                      # folder.py:101: long, no channel dim (the CE target)
            assert sg.dtype == torch.int64 and tuple(sg.shape) == (8, 12) and torch.equal(sg, torch.from_numpy(seg).long())
    with pytest.raises(RuntimeError):
        os.makedirs(str(tmp_path / "empty" / "deeplab256_label" / "x"))
        os.makedirs(str(tmp_path / "empty" / "leftImg256" / "x"))
        TripletFolder(str(tmp_path / "empty"))


def test_segmentation_is_resized_nearest_to_the_frame(tmp_path):
    from vlg.cityscapes import TripletFolder
    root = str(tmp_path / "d")
    data = write_tree(root, "ulm", 3, list(range(0, 8)), hw=(8, 12), seg_hw=(16, 24))      # label maps at twice the size
    it = TripletFolder(root)[0]
    assert tuple(it["seg1"].shape) == (1, 8, 12) and tuple(it["seg3"].shape) == (8, 12)
    # cv2.INTER_NEAREST's index rule (folder.py:133): destination (y, x) <- source (floor(y * 2), floor(x * 2)) for a 2x
    # downscale, i.e. the even rows / columns - NOT PIL's centre sampling, which would take the odd ones
    want = data[0][1][0::2, 0::2]
    assert torch.equal(it["seg1"][0], torch.from_numpy(want.copy()).float())
    assert not np.array_equal(want, np.asarray(Image.fromarray(data[0][1], "L").resize((12, 8), Image.NEAREST)))


def test_nearest_resize_follows_the_cv2_index_rule():
    from vlg.cityscapes import resize_nearest_cv2
    a = np.arange(7 * 5, dtype=np.uint8).reshape(7, 5)
    up = resize_nearest_cv2(a, (14, 10))                         # upscale 2x: every source pixel twice
    assert np.array_equal(up, np.repeat(np.repeat(a, 2, 0), 2, 1))
    down = resize_nearest_cv2(np.arange(64, dtype=np.uint8).reshape(8, 8), (3, 3))   # ratio 8/3: rows floor(0, 2.67, 5.33)
    assert np.array_equal(down, np.arange(64).reshape(8, 8)[[0, 2, 5]][:, [0, 2, 5]])
    assert np.array_equal(resize_nearest_cv2(a, a.shape), a)


def test_loader_shards_like_the_clip_loader(tmp_path):
    from vlg.cityscapes import FRAME_KEYS, TripletFolder, TripletLoader
    from vlg.data import shard_indices
    root = str(tmp_path / "d")
    write_tree(root, "ulm", 3, list(range(0, 20)))                                        # 13 triplets
    ds = TripletFolder(root)
    seen = []
    for rank in range(2):
        ld = TripletLoader(ds, batch=2, rank=rank, world=2, seed=5, shuffle=True, prefetch=2 if rank else 0)
        ld.set_epoch(3)
        idx = shard_indices(len(ds), rank, 2, 3, 5, True)
        assert len(ld) == len(idx) // 2
        batches = list(ld)
        assert len(batches) == len(ld)
        for bi, b in enumerate(batches):
            assert tuple(b.keys()) == FRAME_KEYS and tuple(b["frame1"].shape) == (2, 3, 8, 12) and b["seg3"].dtype == torch.int64
            for j in range(2):
                assert torch.equal(b["frame2"][j], ds[idx[bi * 2 + j]]["frame2"])
        seen += idx[:len(ld) * 2]
    assert len(set(seen)) >= 12                    # the two ranks cover the epoch (one wrapped duplicate at most)
    it = iter(TripletLoader(ds, batch=2, prefetch=2))                                     # abandoning the iterator must not hang
    next(it)
    it.close()
