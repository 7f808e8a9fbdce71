"""Sample transforms of the training pipeline (reference: src/dataloaders/custom_transforms.py:7-133),
restated on numpy (no cv2 in this image).  A sample is the dict DAVIS2016.__getitem__ yields; ``fname`` / ``seq_name``
pass through, 3-D arrays (frames) are resampled with cv2's bicubic kernel (a = -0.75), 2-D arrays (masks) with its
nearest-neighbour rule.

The resampling follows cv2's published algorithm (pixel-centre mapping ``src = (dst + 0.5) / scale - 0.5``, replicated
borders for resize, constant-0 borders for warpAffine, nearest = floor(dst / scale)); cv2 itself is not available here
to generate fixtures against, so these two functions are pinned by properties only (identity at scale 1, exactness on
linear ramps, output sizes): "parity unpinned" for the augmentation arithmetic.
"""
import math
import random

import numpy as np
import torch

_SKIP = ('fname', 'seq_name')


class Compose(object):
    """torchvision.transforms.Compose (torchvision is not installed here)."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


def _round_half_even(v: float) -> int:
    return int(np.rint(v))  # cvRound


def _cubic_weights(frac: np.ndarray) -> np.ndarray:
    """cv2's interpolateCubic: 4 taps at offsets -1..2 for fractional position ``frac`` (A = -0.75), float32."""
    a = np.float32(-0.75)
    x = frac.astype(np.float32)
    w0 = ((a * (x + 1) - 5 * a) * (x + 1) + 8 * a) * (x + 1) - 4 * a
    w1 = ((a + 2) * x - (a + 3)) * x * x + 1
    w2 = ((a + 2) * (1 - x) - (a + 3)) * (1 - x) * (1 - x) + 1
    w3 = np.float32(1) - w0 - w1 - w2
    return np.stack([w0, w1, w2, w3], axis=-1).astype(np.float32)


def _resize_axis_cubic(arr: np.ndarray, out_len: int, scale: float, axis: int) -> np.ndarray:
    n = arr.shape[axis]
    pos = (np.arange(out_len, dtype=np.float64) + 0.5) / scale - 0.5
    base = np.floor(pos).astype(np.int64)
    w = _cubic_weights((pos - base).astype(np.float32))          # [out_len, 4]
    idx = np.clip(base[:, None] + np.arange(-1, 3)[None, :], 0, n - 1)  # replicate border
    a = np.moveaxis(arr, axis, 0).astype(np.float32)
    out = np.zeros((out_len,) + a.shape[1:], dtype=np.float32)
    for k in range(4):
        out += a[idx[:, k]] * w[:, k].reshape((-1,) + (1,) * (a.ndim - 1))
    return np.moveaxis(out, 0, axis)


def resize(arr: np.ndarray, fx: float, fy: float) -> np.ndarray:
    """cv2.resize(arr, None, fx=fx, fy=fy, interpolation=INTER_NEAREST if arr.ndim == 2 else INTER_CUBIC)."""
    h, w = arr.shape[:2]
    ow, oh = _round_half_even(w * fx), _round_half_even(h * fy)
    if ow <= 0 or oh <= 0:
        raise ValueError('resize: empty output')
    if ow == w and oh == h:
        return arr.copy()
    if arr.ndim == 2:
        ys = np.minimum(np.floor(np.arange(oh) / fy).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(ow) / fx).astype(np.int64), w - 1)
        return np.ascontiguousarray(arr[ys][:, xs])
    out = _resize_axis_cubic(arr, ow, fx, axis=1)   # horizontal pass first, as cv2 does
    out = _resize_axis_cubic(out, oh, fy, axis=0)
    return np.ascontiguousarray(out.astype(arr.dtype if arr.dtype.kind == 'f' else np.float32))


def warp_affine(arr: np.ndarray, M: np.ndarray) -> np.ndarray:
    """cv2.warpAffine(arr, M, (w, h), flags=INTER_NEAREST if arr.ndim == 2 else INTER_CUBIC): M maps source to
    destination; pixels that fall outside the source are 0."""
    h, w = arr.shape[:2]
    A = np.vstack([np.asarray(M, dtype=np.float64), [0.0, 0.0, 1.0]])
    inv = np.linalg.inv(A)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    if arr.ndim == 2:
        xi, yi = np.rint(sx).astype(np.int64), np.rint(sy).astype(np.int64)
        ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
        out = np.zeros_like(arr)
        out[ok] = arr[yi[ok], xi[ok]]
        return out
    bx, by = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    wx, wy = _cubic_weights((sx - bx).astype(np.float32)), _cubic_weights((sy - by).astype(np.float32))
    src = arr.astype(np.float32)
    out = np.zeros(arr.shape, dtype=np.float32)
    for j in range(4):
        yy = by + j - 1
        for i in range(4):
            xx = bx + i - 1
            ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            val = np.where(ok[..., None], src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0.0)
            out += val * (wy[..., j] * wx[..., i])[..., None]
    return out


def rotation_matrix(center, angle_deg: float, scale: float) -> np.ndarray:
    """cv2.getRotationMatrix2D."""
    a = scale * math.cos(math.radians(angle_deg))
    b = scale * math.sin(math.radians(angle_deg))
    cx, cy = center
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)


class ScaleNRotate(object):
    """Random zoom and rotation about the centre; tuples = continuous ranges, lists = fixed choices
    (src/dataloaders/custom_transforms.py:7-60; not part of the default training pipeline)."""

    def __init__(self, rots=(-30, 30), scales=(.75, 1.25)):
        assert isinstance(rots, type(scales))
        self.rots, self.scales = rots, scales

    def __call__(self, sample):
        if isinstance(self.rots, tuple):
            rot = (self.rots[1] - self.rots[0]) * random.random() - (self.rots[1] - self.rots[0]) / 2
            sc = (self.scales[1] - self.scales[0]) * random.random() - (self.scales[1] - self.scales[0]) / 2 + 1
        else:
            rot = self.rots[random.randint(0, len(self.rots) - 1)]
            sc = self.scales[random.randint(0, len(self.scales) - 1)]
        for key in sample.keys():
            if key in _SKIP:
                continue
            tmp = sample[key]
            h, w = tmp.shape[:2]
            tmp = warp_affine(tmp, rotation_matrix((w / 2, h / 2), rot, sc))
            if tmp.min() < 0.0:
                tmp = tmp - tmp.min()
            if tmp.max() > 1.0:
                tmp = tmp / tmp.max()
            sample[key] = tmp
        return sample


class Resize(object):
    """Rescale frame and mask by one scale drawn from ``scales`` (src/dataloaders/custom_transforms.py:63-93)."""

    def __init__(self, scales=(0.5, 0.8, 1)):
        self.scales = list(scales)

    def __call__(self, sample):
        sc = self.scales[random.randint(0, len(self.scales) - 1)]
        for key in sample.keys():
            if key not in _SKIP:
                sample[key] = resize(sample[key], sc, sc)
        return sample


class RandomHorizontalFlip(object):
    """Mirror frame and mask left-right with probability 0.5 (src/dataloaders/custom_transforms.py:96-111)."""

    def __call__(self, sample):
        if random.random() < 0.5:
            for key in sample.keys():
                if key not in _SKIP:
                    sample[key] = np.ascontiguousarray(sample[key][:, ::-1])
        return sample


class ToTensor(object):
    """H x W[x C] arrays -> C x H x W tensors (src/dataloaders/custom_transforms.py:114-133)."""

    def __call__(self, sample):
        for key in sample.keys():
            if key in _SKIP:
                continue
            tmp = sample[key]
            if tmp.ndim == 2:
                tmp = tmp[:, :, np.newaxis]
            sample[key] = torch.from_numpy(np.ascontiguousarray(tmp.transpose((2, 0, 1))))
        return sample
