"""DAVIS 2016 dataset (reference: src/dataloaders/davis_2016.py:21-138), restated without cv2.

On-disk layout (``db_root_dir``):
    ImageSets/480p/{train,val,trainval}.txt   lines "/JPEGImages/480p/<seq>/<f>.jpg /Annotations/480p/<seq>/<f>.png"
    JPEGImages/480p/<seq>/<f>.jpg             frames
    Annotations/480p/<seq>/<f>.png            masks (0 / 255)

Same constructor, attributes (``seq_list``, ``fname_list``, ``img_list``, ``labels``) and sample dict as the reference:
``image`` float32 H x W x 3 in BGR order minus ``meanval``, ``gt`` float32 H x W scaled to [0, 1] (all zeros where the
sequence mode hides the annotation), ``seq_name``, ``fname``.  Decoding goes through PIL (the reference uses
``cv2.imread``: BGR channel order, colour masks converted to luminance - both reproduced here); the optional ``inputRes``
resize follows ``scipy.misc.imresize`` (= PIL bilinear for frames, nearest for masks).
"""
import os
from pathlib import Path as P

import numpy as np
from PIL import Image
from torch.utils.data import Dataset

from util.logger import get_logger

log = get_logger(__file__)

MEANVAL = (104.00699, 116.66877, 122.67892)  # BGR, src/dataloaders/davis_2016.py:28


def read_bgr(path: str) -> np.ndarray:
    """uint8 H x W x 3, blue first (what cv2.imread returns)."""
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def read_gray(path: str) -> np.ndarray:
    """uint8 H x W (cv2.imread(path, 0): palette / colour files become luminance)."""
    with Image.open(path) as im:
        return np.asarray(im.convert('L'))


def _imresize(arr: np.ndarray, size, nearest: bool = False) -> np.ndarray:
    """scipy.misc.imresize(arr, size[, interp='nearest']) for uint8 input: size = (rows, cols) or a float factor."""
    if isinstance(size, (int, float)) and not isinstance(size, bool):
        rows, cols = int(arr.shape[0] * float(size)), int(arr.shape[1] * float(size))
    else:
        rows, cols = int(size[0]), int(size[1])
    im = Image.fromarray(arr)
    return np.asarray(im.resize((cols, rows), resample=Image.NEAREST if nearest else Image.BILINEAR))


class DAVIS2016(Dataset):
    """mode 'train' reads ImageSets/480p/train.txt, 'test' reads val.txt; with ``seq_name`` both read trainval.txt,
    keep that sequence only, hide every annotation except the first frame's, and 'train' keeps the first frame only
    (the one-shot setting)."""

    def __init__(self, mode='train', inputRes=None, db_root_dir='/path/to/DAVIS-2016', transform=None,
                 meanval=MEANVAL, seq_name=None):
        self.mode = mode.lower()
        self.inputRes = inputRes
        self.db_root_dir = str(db_root_dir)
        self.transform = transform
        self.meanval = meanval
        self.seq_name = seq_name
        list_of = {'train': 'train', 'test': 'val'}
        if self.mode not in list_of:
            raise Exception("Mode {} does not exist. Must be one of ['train', 'test']".format(mode))
        fname = 'trainval' if seq_name is not None else list_of[self.mode]
        root = P(self.db_root_dir)
        entries = []
        with open(str(root / 'ImageSets' / '480p' / (fname + '.txt'))) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2:
                    entries.append((parts[0], parts[1]))
        if not entries:
            raise RuntimeError('empty image set file: ' + fname + '.txt')
        rows = []
        for img_rel, lab_rel in entries:
            pieces = [p for p in img_rel.split('/') if p]
            rows.append((pieces[-2], pieces[-1].split('.')[0], str(root.joinpath(*pieces)),
                         str(P(*[p for p in lab_rel.split('/') if p]))))
        if seq_name is not None:
            rows = [r for r in rows if r[0] == seq_name]
            if not rows:
                raise RuntimeError('sequence {} is not listed in {}.txt'.format(seq_name, fname))
            rows = [(s, f, i, l if k == 0 else None) for k, (s, f, i, l) in enumerate(rows)]
            if self.mode == 'train':
                rows = rows[:1]
        self.seq_list = [r[0] for r in rows]
        self.fname_list = [r[1] for r in rows]
        self.img_list = [r[2] for r in rows]
        self.labels = [r[3] for r in rows]
        log.info('Done initializing ' + fname + ' Dataset')

    def __len__(self):
        return len(self.img_list)

    def __getitem__(self, idx):
        img, gt = self.make_img_gt_pair(idx)
        sample = {'image': img, 'gt': gt, 'seq_name': self.seq_list[idx], 'fname': self.fname_list[idx]}
        if self.transform is not None:
            sample = self.transform(sample)
        return sample

    def make_img_gt_pair(self, idx):
        img = read_bgr(os.path.join(self.db_root_dir, self.img_list[idx]))
        label = None
        if self.labels[idx] is not None:
            label = read_gray(os.path.join(self.db_root_dir, self.labels[idx]))
        if self.inputRes is not None:
            img = _imresize(img, self.inputRes)
            if label is not None:
                label = _imresize(label, self.inputRes, nearest=True)
        img = np.asarray(img, dtype=np.float32) - np.asarray(self.meanval, dtype=np.float32)
        if label is not None:
            gt = np.asarray(label, dtype=np.float32)
            # (the reference's `gt / np.max([gt.max(), 1e-8])` promotes to float64; the values are the same and every
            # consumer casts to float32, so the mask stays float32 here)
            gt = gt / np.float32(max(float(gt.max()), 1e-8))
        else:
            gt = np.zeros(img.shape[:-1], dtype=np.float32)
        return img, gt

    def get_img_size(self):
        return list(read_bgr(os.path.join(self.db_root_dir, self.img_list[0])).shape[:2])
