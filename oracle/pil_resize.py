"""ORACLE -- test infrastructure only.  numpy restatement of PIL's 8-bit bilinear resize (libImaging/Resample.c of Pillow:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc, ImagingResampleVertical_8bpc) -- the resampling behind
torchvision's F.resized_crop / Resize, which the reference's input pipeline calls (utils/transforms.py:165-172, utils/dataflow.py:140-160).

Pinned against PIL itself (tests/test_input_pipeline.py, CPU: bit-identical on random images, boxes and sizes); the HIP kernel
atomnas_image_preprocess (csrc/preprocess.hip) is checked against PIL-generated fixtures and against this restatement."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """per output position: (first input sample, integer coefficients)"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
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
            a = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - a if a < 1.0 else 0.0)
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


def resize_bilinear_u8(img, out_h, out_w):
    """img: uint8 [H, W, C] -> uint8 [out_h, out_w, C]; horizontal pass, rounding to uint8, vertical pass (PIL's order)"""
    H, W, C = img.shape
    src = img.astype(np.int64)
    tmp = np.empty((H, out_w, C), dtype=np.int64)
    for xx, (xmin, k) in enumerate(coeffs(W, out_w)):
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, xmin:xmin + len(k), :] * k[None, :, None]).sum(1)
        tmp[:, xx, :] = _clip8(acc)
    out = np.empty((out_h, out_w, C), dtype=np.int64)
    for yy, (ymin, k) in enumerate(coeffs(H, out_h)):
        acc = (1 << (PRECISION_BITS - 1)) + (tmp[ymin:ymin + len(k), :, :] * k[:, None, None]).sum(0)
        out[yy] = _clip8(acc)
    return out.astype(np.uint8)


def crop_resize_flip(img, box, size, flip):
    """F.resized_crop(img, i, j, h, w, size, BILINEAR) followed by an optional horizontal flip, on a uint8 HWC array"""
    i, j, h, w = box
    r = resize_bilinear_u8(img[i:i + h, j:j + w, :], size, size)
    return r[:, ::-1, :].copy() if flip else r


def to_tensor_normalize(u8_hwc, mean, std):
    """transforms.ToTensor + Normalize in fp32: [H, W, 3] uint8 -> [3, H, W] float32"""
    t = u8_hwc.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (t - np.asarray(mean, dtype=np.float32)[:, None, None]) / np.asarray(std, dtype=np.float32)[:, None, None]
