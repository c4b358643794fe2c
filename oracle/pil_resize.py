"""Oracle (test infrastructure): Pillow's 8-bit BILINEAR `Image.resize`, restated in numpy.

The reference's `SamPredictor.set_image` (model/segment_anything/predictor.py:34-60) resizes the uint8 image with
`ResizeLongestSide.apply_image` (utils/transforms.py:27-35): `np.array(resize(to_pil_image(image), target_size))`, torchvision's thin wrapper
over `PIL.Image.resize(size, BILINEAR)`.  torchvision is absent from /root/reference's environment here; Pillow itself (12.2.0, the library that does
the arithmetic) is installed, so this restatement is PINNED against `PIL.Image.resize` directly (tests/test_oracle_golden.py).

Algorithm (Pillow src/libImaging/Resample.c, `precompute_coeffs` / `normalize_coeffs_8bpc` / `ImagingResampleHorizontal_8bpc` / `...Vertical_8bpc`):
  scale = in / out; filterscale = max(scale, 1); support = 1.0 * filterscale (bilinear); for output index xx:
    center = (xx + 0.5) * scale; xmin = max(0, int(center - support + 0.5)); xmax = min(in, int(center + support + 0.5)) - xmin
    w[x] = triangle((x + xmin - center + 0.5) / filterscale), normalised by their sum (double arithmetic)
    k[x] = int(w[x] * 2^22 + 0.5)                                   (PRECISION_BITS = 32 - 8 - 2; weights are never negative here)
  out = clip8((2^21 + sum_x in[xmin + x] * k[x]) >> 22), the horizontal pass first (uint8 intermediate), then the vertical pass; a pass whose
  size does not change is skipped.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """-> (xmin int32 [out], count int32 [out], k int32 [out, ksize])"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32); cnt = np.zeros(out_size, np.int32); kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(n, np.float64)
        ww = 0.0
        for x in range(n):
            a = (x + lo - center + 0.5) * ss
            a = -a if a < 0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        xmin[xx], cnt[xx] = lo, n
        kk[xx, :n] = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
    return xmin, cnt, kk


def _pass(img, out_size, axis):
    """img uint8 [H, W, C]; resample along `axis` (0 = vertical, 1 = horizontal)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    xmin, cnt, kk = coeffs(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt[xx]):
            acc += src[xmin[xx] + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C], bit-identical to `np.array(PIL.Image.fromarray(img).resize((out_w, out_h), BILINEAR))`."""
    h, w = img.shape[:2]
    x = img
    if out_w != w:
        x = _pass(x, out_w, 1)
    if out_h != h:
        x = _pass(x, out_h, 0)
    return x


def apply_image(img, long_side=1024):
    """utils/transforms.py:27-35 (`ResizeLongestSide.apply_image`)."""
    h, w = img.shape[:2]
    sc = long_side * 1.0 / max(h, w)
    return resize_bilinear_u8(img, int(h * sc + 0.5), int(w * sc + 0.5))
