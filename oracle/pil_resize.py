"""ORACLE -- test infrastructure only.  numpy restatement of PIL's 8-bit bilinear / bicubic resize (libImaging/Resample.c of Pillow:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc, ImagingResampleVertical_8bpc) -- the resampling behind
torchvision's F.resized_crop / Resize, which the reference's input pipeline calls (utils/transforms.py:165-172, utils/dataflow.py:140-160).

Pinned against PIL itself (tests/test_input_pipeline.py, CPU: bit-identical on random images, boxes and sizes); the HIP kernel
atomnas_image_preprocess (csrc/preprocess.hip) is checked against PIL-generated fixtures and against this restatement."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


BILINEAR, BICUBIC = "bilinear", "bicubic"


def _filter(filt):
    """(support, weight function) of Resample.c: bilinear_filter (triangle, support 1), bicubic_filter (Keys cubic with a = -0.5,
    support 2) -- torchvision's InterpolationMode.BILINEAR / BICUBIC on PIL images"""
    if filt == BILINEAR:
        return 1.0, lambda x: 1.0 - x if x < 1.0 else 0.0
    if filt == BICUBIC:
        a = -0.5

        def w(x):
            if x < 1.0:
                return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
            if x < 2.0:
                return (((x - 5) * x + 8) * x - 4) * a
            return 0.0
        return 2.0, w
    raise ValueError(filt)


def coeffs(in_size, out_size, filt=BILINEAR):
    """per output position: (first input sample, integer coefficients)"""
    fsupport, fw = _filter(filt)
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = []
        for x in range(xmax):
            w.append(fw(abs((x + xmin - center + 0.5) * ss)))
        ww = 0.0
        for v in w:
            ww += v
        k = []
        for v in w:
            if ww != 0.0:
                v = v / ww
            k.append(int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)))
        out.append((xmin, np.asarray(k, dtype=np.int64)))
    return out


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255)


def resize_u8(img, out_h, out_w, filt=BILINEAR):
    """img: uint8 [H, W, C] -> uint8 [out_h, out_w, C]; horizontal pass, rounding to uint8, vertical pass (PIL's order)"""
    H, W, C = img.shape
    src = img.astype(np.int64)
    tmp = np.empty((H, out_w, C), dtype=np.int64)
    for xx, (xmin, k) in enumerate(coeffs(W, out_w, filt)):
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, xmin:xmin + len(k), :] * k[None, :, None]).sum(1)
        tmp[:, xx, :] = _clip8(acc)
    out = np.empty((out_h, out_w, C), dtype=np.int64)
    for yy, (ymin, k) in enumerate(coeffs(H, out_h, filt)):
        acc = (1 << (PRECISION_BITS - 1)) + (tmp[ymin:ymin + len(k), :, :] * k[:, None, None]).sum(0)
        out[yy] = _clip8(acc)
    return out.astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    return resize_u8(img, out_h, out_w, BILINEAR)


def crop_resize_flip(img, box, size, flip, filt=BILINEAR):
    """F.resized_crop(img, i, j, h, w, size, interpolation) followed by an optional horizontal flip, on a uint8 HWC array"""
    i, j, h, w = box
    r = resize_u8(img[i:i + h, j:j + w, :], size, size, filt)
    return r[:, ::-1, :].copy() if flip else r


def to_tensor_normalize(u8_hwc, mean, std):
    """transforms.ToTensor + Normalize in fp32: [H, W, 3] uint8 -> [3, H, W] float32"""
    t = u8_hwc.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (t - np.asarray(mean, dtype=np.float32)[:, None, None]) / np.asarray(std, dtype=np.float32)[:, None, None]
