"""ORACLE (test infrastructure, never shipped or timed as the product).

CPU restatement (numpy, integer arithmetic) of the input pre-step in front of PatchEmbed:

    uint8 HxWx3 face crop --bicubic resize to 224x224 (uint8)--> ToTensor (/255) --> Normalize(.5,.5) --> 4x4 patches

Two flavours, because the reference resizes with two different libraries:

* **pil** -- the Aff-Wild2 path: `transforms.Resize(224, interpolation=BICUBIC)` on a PIL image *before* ToTensor
  (utils/util.py:43-52), i.e. Pillow's `Image.resize` (third-party; the reference's requirements.txt leaves the version
  open through torchvision==0.13.0; restated here from Pillow's published algorithm, src/libImaging/Resample.c:
  `precompute_coeffs` + `ImagingResampleHorizontal_8bpc` / `...Vertical_8bpc`): a = -0.5 cubic, support 2 taps each side
  for up-scaling, window truncated (not replicated) at the border and renormalised, coefficients rounded to 22-bit fixed
  point, horizontal pass first with its result rounded and clipped to uint8, then the vertical pass likewise.
  **PINNED**: Pillow (12.2.0) is importable in the development container and on the GPU image;
  `oracle/gen_golden.py` runs Pillow itself on hash-generated crops and commits the outputs (tests/golden/preproc.npz);
  tests/test_oracle_golden.py holds this restatement to them BIT-EXACTLY, and where Pillow is importable
  tests/test_preproc_cpu.py additionally compares against Pillow live on random crops of several sizes.
* **cv2** -- the MELD path: `cv2.resize(im, (224,224), interpolation=cv2.INTER_CUBIC)` on the uint8 BGR array
  (utils/dataset.py:54-57), i.e. OpenCV's 8-bit cubic resize (third-party `opencv-python`, version open in
  requirements.txt; restated from OpenCV's published algorithm, modules/imgproc/src/resize.cpp: `interpolateCubic`
  with A = -0.75 evaluated in float, coefficients scaled by 2^11 and rounded to short, replicate border, un-rounded
  int32 horizontal sums, vertical pass with 22-bit fixed-point rounding and saturation).
  **PARITY UNPINNED**: cv2 is in neither container, the reference holds no fixture for it, and OpenCV's vectorised
  vertical pass rounds through float, which may differ from this scalar form by one LSB on some pixels.  It is held
  only to properties (tests/test_preproc_cpu.py: constant images, flips, closeness to a float a=-0.75 evaluation).

ToTensor + Normalize follow torchvision (float32: v / 255, then (t - 0.5) / 0.5) and the patch order follows
PatchEmbed's Conv2d weight layout (Swin_Transformer.py:407,419): column = c*16 + dy*4 + dx.
"""
from __future__ import annotations

import math

import numpy as np

PIL_BITS = 22          # Pillow: PRECISION_BITS = 32 - 8 - 2
CV_BITS = 11           # OpenCV: INTER_RESIZE_COEF_BITS


def _pil_cubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_table(in_size: int, out_size: int):
    """Resample.c:precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter.  Returns (idx [out,4] int32,
    w [out,4] int32): out[x] = clip8((2^21 + sum_k src[idx[x,k]] * w[x,k]) >> 22).  Windows shorter than 4 taps
    (image border) are padded with weight-0 taps on a valid index."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    if math.ceil(support) * 2 + 1 > 5 and in_size >= out_size:
        raise ValueError("oracle.preproc: up-scaling only (the reference enlarges 112/160 -> 224)")
    idx = np.zeros((out_size, 4), np.int32)
    w = np.zeros((out_size, 4), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        n = xmax - xmin
        k = [_pil_cubic((x + xmin - center + 0.5) / fscale) for x in range(n)]
        ww = sum(k)
        k = [v / ww for v in k]
        assert n <= 4, (in_size, out_size, xx, n)
        for i in range(4):
            idx[xx, i] = xmin + min(i, n - 1)
            if i < n:
                w[xx, i] = int(k[i] * (1 << PIL_BITS) + (0.5 if k[i] >= 0 else -0.5))     # C cast truncates toward zero
    return idx, w


def _cv_round_short(v: np.float32) -> int:
    r = int(np.rint(np.float32(v)))                  # cvRound: round half to even
    return max(-32768, min(32767, r))


def cv2_table(in_size: int, out_size: int):
    """resize.cpp: fx = (dx + 0.5) * scale - 0.5 in float, sx = floor(fx), interpolateCubic(fx - sx) with A = -0.75
    in float, taps sx-1 .. sx+2 with replicate border, coefficients saturate_cast<short>(c * 2048)."""
    scale = 1.0 / (out_size / in_size)
    idx = np.zeros((out_size, 4), np.int32)
    w = np.zeros((out_size, 4), np.int32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    for dx in range(out_size):
        fx = np.float32((dx + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        x = np.float32(fx - np.float32(sx))
        c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
        c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
        c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
        c3 = one - c0 - c1 - c2
        for k, c in enumerate((c0, c1, c2, c3)):
            idx[dx, k] = min(max(sx - 1 + k, 0), in_size - 1)
            w[dx, k] = _cv_round_short(np.float32(c) * np.float32(1 << CV_BITS))
    return idx, w


def table(mode: str, in_size: int, out_size: int):
    if mode == "pil":
        return pil_table(in_size, out_size)
    if mode == "cv2":
        return cv2_table(in_size, out_size)
    raise ValueError(f"resize flavour {mode!r}: 'pil' (utils/util.py:45) or 'cv2' (utils/dataset.py:57)")


def resize_u8(img: np.ndarray, mode: str, out_size: int = 224) -> np.ndarray:
    """img (..., H, W, C) uint8, H == W -> (..., out, out, C) uint8."""
    assert img.dtype == np.uint8 and img.shape[-3] == img.shape[-2]
    idx, w = table(mode, img.shape[-2], out_size)
    a = img.astype(np.int64)
    if mode == "pil":
        h = np.full(a.shape[:-2] + (out_size, a.shape[-1]), 1 << (PIL_BITS - 1), np.int64)
        for k in range(4):
            h += a[..., :, idx[:, k], :] * w[:, k].astype(np.int64)[:, None]
        h = np.clip(h >> PIL_BITS, 0, 255)                               # uint8 intermediate
        v = np.full(a.shape[:-3] + (out_size, out_size, a.shape[-1]), 1 << (PIL_BITS - 1), np.int64)
        for k in range(4):
            v += h[..., idx[:, k], :, :] * w[:, k].astype(np.int64)[:, None, None]
        return np.clip(v >> PIL_BITS, 0, 255).astype(np.uint8)
    h = np.zeros(a.shape[:-2] + (out_size, a.shape[-1]), np.int64)       # cv2: un-rounded int32 row sums
    for k in range(4):
        h += a[..., :, idx[:, k], :] * w[:, k].astype(np.int64)[:, None]
    v = np.zeros(a.shape[:-3] + (out_size, out_size, a.shape[-1]), np.int64)
    for k in range(4):
        v += h[..., idx[:, k], :, :] * w[:, k].astype(np.int64)[:, None, None]
    return np.clip((v + (1 << (2 * CV_BITS - 1))) >> (2 * CV_BITS), 0, 255).astype(np.uint8)


def normalize_lut() -> np.ndarray:
    """ToTensor + Normalize(mean .5, std .5) per byte value, in float32 as torchvision computes it"""
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    return ((v - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def frames_from_u8(img: np.ndarray, mode: str) -> np.ndarray:
    """(n, S, S, 3) uint8 -> (n, 3, 224, 224) float32: the tensor the reference hands to the Swin model"""
    r = resize_u8(img, mode, 224)
    return np.ascontiguousarray(normalize_lut()[r].transpose(0, 3, 1, 2))


def patch_cols(frames: np.ndarray) -> np.ndarray:
    """(n, 3, 224, 224) -> (n*3136, 48) rows = patches in raster order, column = c*16 + dy*4 + dx"""
    n = frames.shape[0]
    x = frames.reshape(n, 3, 56, 4, 56, 4).transpose(0, 2, 4, 1, 3, 5)
    return np.ascontiguousarray(x.reshape(n * 3136, 48))


def patch_cols_from_u8(img: np.ndarray, mode: str) -> np.ndarray:
    return patch_cols(frames_from_u8(img, mode))
