"""the DAVIS 2016 loader and the sample transforms, against a small DAVIS-shaped tree written by the
test itself (no dataset ships offline) - file-list semantics of src/dataloaders/davis_2016.py:40-99, sample layout of
:101-134, transform behaviour of src/dataloaders/custom_transforms.py:63-133."""
import os
import random
import sys
import zlib

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

from dataloaders import custom_transforms as T  # noqa: E402
from dataloaders.davis_2016 import DAVIS2016, MEANVAL  # noqa: E402

H, W = 24, 40
SEQS = {"bear": 3, "camel": 2, "dog": 2}
SPLIT = {"train": ["bear", "camel"], "val": ["dog"]}


def _frame(seq, k):
    rng = np.random.RandomState(zlib.crc32(("%s/%d" % (seq, k)).encode()))
    return rng.randint(0, 256, size=(H, W, 3)).astype(np.uint8)  # RGB as stored


def _mask(seq, k):
    m = np.zeros((H, W), dtype=np.uint8)
    m[4 + k:14 + k, 10:25] = 255
    return m


@pytest.fixture(scope="module")
def davis_root(tmp_path_factory):
    root = tmp_path_factory.mktemp("davis")
    lines = {"train": [], "val": []}
    for seq, n in SEQS.items():
        (root / "JPEGImages" / "480p" / seq).mkdir(parents=True)
        (root / "Annotations" / "480p" / seq).mkdir(parents=True)
        for k in range(n):
            Image.fromarray(_frame(seq, k)).save(str(root / "JPEGImages" / "480p" / seq / ("%05d.jpg" % k)), quality=100,
                                                 subsampling=0)
            Image.fromarray(_mask(seq, k)).save(str(root / "Annotations" / "480p" / seq / ("%05d.png" % k)))
            line = "/JPEGImages/480p/%s/%05d.jpg /Annotations/480p/%s/%05d.png \n" % (seq, k, seq, k)
            for split, members in SPLIT.items():
                if seq in members:
                    lines[split].append(line)
    (root / "ImageSets" / "480p").mkdir(parents=True)
    for split in ("train", "val"):
        (root / "ImageSets" / "480p" / (split + ".txt")).write_text("".join(lines[split]))
    (root / "ImageSets" / "480p" / "trainval.txt").write_text("".join(lines["train"] + lines["val"]))
    return root


def test_split_lists(davis_root):
    tr = DAVIS2016(mode="train", db_root_dir=str(davis_root))
    te = DAVIS2016(mode="test", db_root_dir=str(davis_root))
    assert len(tr) == 5 and len(te) == 2
    assert tr.seq_list == ["bear"] * 3 + ["camel"] * 2 and te.seq_list == ["dog", "dog"]
    assert tr.fname_list[:3] == ["00000", "00001", "00002"]
    assert all(l is not None for l in tr.labels)
    with pytest.raises(Exception):
        DAVIS2016(mode="val", db_root_dir=str(davis_root))


def test_sequence_mode_hides_all_but_the_first_annotation(davis_root):
    tr = DAVIS2016(mode="train", db_root_dir=str(davis_root), seq_name="dog")
    assert len(tr) == 1 and tr.labels[0] is not None and tr.fname_list == ["00000"]  # the one-shot frame
    te = DAVIS2016(mode="test", db_root_dir=str(davis_root), seq_name="bear")
    assert len(te) == 3 and te.labels[0] is not None and te.labels[1:] == [None, None]
    s0, s1 = te[0], te[1]
    assert s0["gt"].max() == 1.0 and not s1["gt"].any() and s1["gt"].shape == (H, W)
    with pytest.raises(RuntimeError):
        DAVIS2016(mode="test", db_root_dir=str(davis_root), seq_name="no-such-sequence")


def test_sample_layout_bgr_mean_and_gt_scale(davis_root):
    ds = DAVIS2016(mode="train", db_root_dir=str(davis_root))
    s = ds[1]
    assert set(s) == {"image", "gt", "seq_name", "fname"} and s["seq_name"] == "bear" and s["fname"] == "00001"
    img, gt = s["image"], s["gt"]
    assert img.dtype == np.float32 and img.shape == (H, W, 3) and gt.dtype == np.float32 and gt.shape == (H, W)
    with Image.open(os.path.join(str(davis_root), "JPEGImages", "480p", "bear", "00001.jpg")) as im:
        rgb = np.asarray(im.convert("RGB")).astype(np.float32)
    want = rgb[:, :, ::-1] - np.asarray(MEANVAL, dtype=np.float32)  # blue first, dataset mean off
    assert np.array_equal(img, want)
    assert np.array_equal(gt, (_mask("bear", 1) / 255.0).astype(np.float32)) and set(np.unique(gt)) == {0.0, 1.0}
    assert ds.get_img_size() == [H, W]


def test_input_res_resize(davis_root):
    ds = DAVIS2016(mode="train", db_root_dir=str(davis_root), inputRes=(12, 20))
    s = ds[0]
    assert s["image"].shape == (12, 20, 3) and s["gt"].shape == (12, 20)
    assert set(np.unique(s["gt"])) <= {0.0, 1.0}  # nearest: no blended label values


def test_resize_sizes_identity_and_nearest():
    img = np.random.RandomState(0).rand(480, 854, 3).astype(np.float32)
    gt = (np.random.RandomState(1).rand(480, 854) > 0.5).astype(np.float32)
    assert T.resize(img, 1, 1).shape == img.shape and np.array_equal(T.resize(img, 1, 1), img)
    assert T.resize(img, 0.8, 0.8).shape == (384, 683, 3)   # the 384x683 frames of the reference's fine-tune
    assert T.resize(img, 0.5, 0.5).shape == (240, 427, 3)
    g = T.resize(gt, 0.5, 0.5)
    assert g.shape == (240, 427) and np.array_equal(g, gt[::2, ::2][:, :427])  # floor(dst / scale)
    assert T.resize(np.zeros((3, 4), np.float32), 0.5, 0.5).shape == (2, 2)    # cvRound(1.5) = 2


def test_cubic_resize_is_exact_on_linear_ramps():
    ys, xs = np.meshgrid(np.arange(40, dtype=np.float32), np.arange(64, dtype=np.float32), indexing="ij")
    ramp = np.stack([2 * xs + 1, 3 * ys - 5, xs + ys], axis=-1)
    out = T.resize(ramp, 0.5, 0.5)
    oy, ox = np.meshgrid(np.arange(20), np.arange(32), indexing="ij")
    sx, sy = (ox + 0.5) / 0.5 - 0.5, (oy + 0.5) / 0.5 - 0.5          # pixel-centre mapping
    want = np.stack([2 * sx + 1, 3 * sy - 5, sx + sy], axis=-1)
    inner = (slice(1, -1), slice(1, -1))                              # borders are replicated, not extrapolated
    assert np.allclose(out[inner], want[inner], atol=1e-4)
    w = T._cubic_weights(np.array([0.0, 0.25, 0.5], dtype=np.float32))
    assert np.allclose(w.sum(-1), 1.0) and np.allclose(w[0], [0, 1, 0, 0])
    assert np.allclose(w[2], [-0.09375, 0.59375, 0.59375, -0.09375])  # a = -0.75 at the half-way point


def test_flip_and_to_tensor():
    random.seed(0)
    img = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)
    gt = np.arange(6, dtype=np.float32).reshape(2, 3)
    flips = 0
    for _ in range(40):
        s = T.RandomHorizontalFlip()({"image": img.copy(), "gt": gt.copy(), "fname": "f", "seq_name": "s"})
        flipped = np.array_equal(s["image"], img[:, ::-1])
        assert flipped or np.array_equal(s["image"], img)
        assert np.array_equal(s["gt"], gt[:, ::-1] if flipped else gt)  # frame and mask flip together
        flips += flipped
    assert 8 <= flips <= 32
    t = T.ToTensor()({"image": img.copy(), "gt": gt.copy(), "fname": "f", "seq_name": "s"})
    assert t["image"].shape == (3, 2, 3) and t["gt"].shape == (1, 2, 3) and t["fname"] == "f"
    assert torch.equal(t["image"][1], torch.from_numpy(img[:, :, 1]))


def test_scale_n_rotate_identity_and_quarter_turn():
    random.seed(1)
    img = np.random.RandomState(2).rand(9, 9, 3).astype(np.float32)
    gt = (np.random.RandomState(3).rand(9, 9) > 0.5).astype(np.float32)
    same = T.ScaleNRotate(rots=[0], scales=[1.0])({"image": img.copy(), "gt": gt.copy()})
    assert np.allclose(same["image"], img, atol=1e-5) and np.array_equal(same["gt"], gt)
    M = T.rotation_matrix((4.0, 4.0), 90.0, 1.0)      # odd size, centre on a pixel: an exact quarter turn
    assert np.array_equal(T.warp_affine(gt, M), np.rot90(gt))


def test_training_pipeline_through_the_factories(davis_root):
    from util import io_helper
    random.seed(3)
    dl = io_helper.get_data_loader_train(davis_root, 1, seq_name="camel", resident=False)  # (the DataLoader itself)
    assert len(dl) == 1
    batch = dl.dataset[0]  # (the factory's worker processes are not needed to check the composed transforms)
    assert batch["image"].dim() == 3 and batch["image"].shape[0] == 3 and batch["gt"].shape[0] == 1
    assert batch["image"].shape[1:] == batch["gt"].shape[1:] and batch["image"].dtype == torch.float32
    assert batch["image"].shape[1] in (12, 19, 24)    # scales 0.5 / 0.8 / 1 of 24 rows (cvRound(19.2) = 19)
    te = io_helper.get_data_loader_test(davis_root, 1, seq_name="camel").dataset
    assert len(te) == 2 and te[1]["gt"].sum() == 0


def test_resident_one_shot_loader_equals_the_per_iteration_pipeline(davis_root):
    """A sequence run's training loader (src/util/io_helper.py:62-70 on the ONE sample of src/dataloaders/davis_2016.py:72-83):
    the device-resident loader (six flip x scale variants built once) against the per-iteration DataLoader (worker process,
    decode, flip, rescale every epoch) under the same torch seed - the same (shape, flip) sequence and the same tensors, bit
    for bit, epoch after epoch, and torch's default generator left in the same state."""
    from util import io_helper
    from dataloaders.resident import ResidentOneShotLoader
    n_epochs = 12
    runs = {}
    for tag, resident in (("pipeline", False), ("resident", True)):
        torch.manual_seed(77)
        loader = io_helper.get_data_loader_train(str(davis_root), 1, "bear", resident=resident)
        assert isinstance(loader, ResidentOneShotLoader) == resident and len(loader) == 1
        seen = []
        for _ in range(n_epochs):
            batches = list(loader)
            assert len(batches) == 1
            seen.append(batches[0])
        runs[tag] = (seen, torch.rand(3))  # (the generator's state afterwards shows in the next draw)
    assert torch.equal(runs["pipeline"][1], runs["resident"][1])
    shapes = set()
    for a, b in zip(runs["pipeline"][0], runs["resident"][0]):
        assert a["seq_name"] == b["seq_name"] == ["bear"] and a["fname"] == b["fname"]
        for key in ("image", "gt"):
            assert a[key].dtype == b[key].dtype and tuple(a[key].shape) == tuple(b[key].shape), key
            assert torch.equal(a[key], b[key].cpu()), key
        shapes.add(tuple(a["image"].shape))
    assert len(shapes) >= 2  # the scale really changes between epochs under this seed
    res = io_helper.get_data_loader_train(str(davis_root), 1, "bear")
    assert len(res.variants) == 6 and {tuple(v["image"].shape[2:]) for v in res.variants.values()} == {(12, 20), (19, 32), (24, 40)}
    # the whole training set (no sequence) keeps the DataLoader
    assert not isinstance(io_helper.get_data_loader_train(str(davis_root), 1), ResidentOneShotLoader)
