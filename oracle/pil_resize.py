"""TEST INFRASTRUCTURE (oracle) -- not part of the product path.

CPU restatement of the reference's image pre-processing (dataset/transform_func.py:101-124, `make_transform`):

    Resize((S, S))          transform_func.py:19-31  -> torchvision.transforms.functional.resize(PIL image, (S, S),
                                                        Image.BILINEAR) = PIL `Image.resize((S, S), BILINEAR)`
    ToTensor()              transform_func.py:51-66  -> (uint8 HWC / 255) as float64, CHW
    Normalize(mean, std)    transform_func.py:91-99  -> (x - mean) / std   (float64; engine.py:25 casts to float32)

The resampling itself lives in a third-party dependency of the reference that is not vendored in /root/reference:
Pillow (any version >= 7; 12.2.0 is installed in this image), `src/libImaging/Resample.c`: `precompute_coeffs`,
`normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`, `ImagingResampleVertical_8bpc`.  Its published algorithm for
8-bit images, restated here:

  * per output coordinate xx: scale = in/out, filterscale = max(scale, 1), support = 1.0 * filterscale (triangle
    filter), center = (xx + 0.5) * scale, xmin = int(center - support + 0.5) clipped to 0, xmax = int(center + support
    + 0.5) clipped to in;  weights w_x = tri((x + xmin - center + 0.5) / filterscale), normalised by their sum;
  * fixed point: k_int = int(0.5 + k * 2^22)  (k >= 0 for the triangle filter);
  * horizontal pass into a uint8 intermediate, then vertical pass: out = clip8((2^21 + sum px * k_int) >> 22);
    a pass is skipped when that dimension does not change.

PINNED: tests/test_oracle_resize.py checks this restatement bit for bit against Pillow itself (when importable) over
random sizes, and against tests/golden/resize_*.npz, which oracle/gen_golden_resize.py produced with Pillow."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2

NORMALIZE = {"MNIST": ([0.1307], [0.3081]),                                       # transform_func.py:102-105
             "CUB200": ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
             "ConText": ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
             "ImageNet": ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])}


def precompute_coeffs(in_size, out_size):
    """-> (ksize, bounds [out,2] int32 (xmin, count), kk [out, ksize] int32 fixed-point weights)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.empty(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * float(1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, out_size, axis):
    """One resampling pass of a uint8 array [H, W, C] along `axis` (0 vertical, 1 horizontal)."""
    in_size = img.shape[axis]
    _, bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)                # [in, ...]
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """PIL `Image.resize((out_w, out_h), BILINEAR)` of a uint8 image [H, W] or [H, W, C]."""
    a = np.asarray(img, np.uint8)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    if a.shape[1] != out_w:
        a = _pass(a, out_w, 1)
    if a.shape[0] != out_h:
        a = _pass(a, out_h, 0)
    return a[:, :, 0] if squeeze else a


def to_tensor_normalize(u8, mean, std):
    """ToTensor + Normalize of transform_func.py, float64 in / float32 out as the engine casts (engine.py:25)."""
    a = np.asarray(u8)
    if a.ndim == 2:
        a = a[:, :, None]
    x = (a / 255).transpose(2, 0, 1)
    m = np.asarray(mean, np.float64)[:, None, None]
    s = np.asarray(std, np.float64)[:, None, None]
    return ((x - m) / s).astype(np.float32)


def normalize_lut(mean, std):
    """[C, 256] float32: every value ToTensor + Normalize can produce (the GPU kernel indexes it by the resized byte)."""
    v = np.arange(256)[None, :] / 255
    return ((v - np.asarray(mean, np.float64)[:, None]) / np.asarray(std, np.float64)[:, None]).astype(np.float32)


def transform(img_u8, size, dataset):
    mean, std = NORMALIZE[dataset]
    return to_tensor_normalize(resize_bilinear_u8(img_u8, size, size), mean, std)
