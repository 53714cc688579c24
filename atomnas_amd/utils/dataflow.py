"""Input pipeline on the GPU: the reference's DataPrefetcher (utils/dataflow.py:13-58) with the per-sample pixel work
moved into a HIP kernel.

    reference:  DataLoader workers: PIL decode -> crop -> resize -> flip -> ToTensor -> Normalize (CPU)  -> prefetcher: H2D of fp32 batches
    here:       loader: decoded uint8 HWC images + (box, flip) decisions (atomnas_amd/utils/transforms.py) -> prefetcher: H2D of the
                uint8 pixels on a side stream (a quarter of the fp32 bytes), atomnas_image_preprocess on that stream -> fp32 NCHW batch

Same iterator protocol as the reference's DataPrefetcher (__iter__ / __next__ / __len__, batch k + 1 in flight while batch k trains,
`torch.cuda.current_stream().wait_stream(side)` at hand-over).  JPEG decoding and LMDB reading are out of scope in this image (no
decoder, no lmdb): loaders hand over decoded uint8 arrays, e.g. SyntheticDecodedImages below (bench.py --input-pipeline uint8).
"""
import ctypes
import random

import numpy as np
import torch

from .. import _lib
from . import transforms as T

DESC_DTYPE = np.dtype([("off", "<i8"), ("H", "<i4"), ("W", "<i4"), ("bi", "<i4"), ("bj", "<i4"), ("bh", "<i4"), ("bw", "<i4"), ("flip", "<i4"),
                       ("pad", "<i4")])   # struct atomnas_img_desc (include/atomnas_hip.h), 40 bytes
MAX_SCALE = 9.0   # the kernel's tap budget: a crop side may be at most 9x the output side


def preprocess(pool_dev, desc_dev, n, size, mean, std, out, out_mode=0, stream=None):
    """launches atomnas_image_preprocess: pool_dev uint8 device tensor, desc_dev uint8 device tensor holding n DESC_DTYPE records"""
    st = ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    _lib.call("atomnas_image_preprocess", ctypes.c_void_p(pool_dev.data_ptr()), ctypes.c_void_p(desc_dev.data_ptr()), int(n), int(size),
              ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
              ctypes.c_void_p(out.data_ptr()), int(out_mode), st)


def check_box(H, W, box, size):
    i, j, h, w = box
    if not (0 <= i and 0 <= j and h > 0 and w > 0 and i + h <= H and j + w <= W):
        raise ValueError("crop box %s outside a %d x %d image" % (box, H, W))
    if h > MAX_SCALE * size or w > MAX_SCALE * size:
        raise ValueError("crop box %s is more than %gx the output size %d" % (box, MAX_SCALE, size))


class DevicePrefetcher(object):
    """DataPrefetcher (utils/dataflow.py:13-58) for decoded uint8 samples.  `loader` yields batches (images, boxes, flips, targets):
    images = list of uint8 HWC tensors (pinned memory makes the copies asynchronous), boxes = list of (top, left, height, width),
    flips = list of bool, targets = int64 tensor.  Yields (input fp32 [N, 3, S, S] on the GPU, target on the GPU)."""

    def __init__(self, loader, image_size=224, mean=T.IMAGENET_MEAN, std=T.IMAGENET_STD, max_image_bytes=3 * 640 * 640):
        if not torch.cuda.is_available():
            raise _lib.AtomnasHipError("DevicePrefetcher needs the GPU (the preprocessing kernel has no CPU fallback)")
        self.loader_len = len(loader) if hasattr(loader, "__len__") else None
        self.loader = iter(loader)
        self.size, self.mean, self.std = int(image_size), tuple(mean), tuple(std)
        self.stream = torch.cuda.Stream()
        self.max_image_bytes = int(max_image_bytes)
        self.slots = [None, None]   # per slot: (device pool, device descriptors, pinned descriptors, output) sized on first use
        # per slot: event behind the last host-to-device copy that READ the slot's pinned descriptors.  The host rewrites them for the
        # batch after next; nothing else orders the host against that copy (the reference's prefetcher never reuses host staging
        # memory), and a caller that does not synchronise per step (graph replay) runs several steps ahead of the device.
        self.desc_read = [None, None]
        self.k = 0
        self.stop = False
        self.preload()

    def _slot(self, n, nbytes):
        s = self.slots[self.k & 1]
        if s is None or s[0].numel() < nbytes or s[3].shape[0] != n:
            cap = max(nbytes, n * self.max_image_bytes // 4)
            s = (torch.empty(cap, dtype=torch.uint8, device="cuda"), torch.empty(n * DESC_DTYPE.itemsize, dtype=torch.uint8, device="cuda"),
                 torch.empty(n * DESC_DTYPE.itemsize, dtype=torch.uint8).pin_memory(),
                 torch.empty(n, 3, self.size, self.size, dtype=torch.float32, device="cuda"))
            self.slots[self.k & 1] = s
        return s

    def preload(self):
        try:
            images, boxes, flips, target = next(self.loader)
        except StopIteration:
            self.stop = True
            self.next_input = self.next_target = None
            return
        n = len(images)
        sizes = [int(im.numel()) for im in images]
        offs = np.concatenate([[0], np.cumsum([(b + 15) // 16 * 16 for b in sizes])])
        # the slot's previous batch was handed out two iterations ago; its consumer ran on the current stream before this call
        self.stream.wait_stream(torch.cuda.current_stream())
        pool, desc_dev, desc_pin, out = self._slot(n, int(offs[-1]))
        ev = self.desc_read[self.k & 1]
        if ev is not None:
            ev.synchronize()   # the copy queued two batches ago has read desc_pin (normally long done: no wait in steady state)
        d = np.frombuffer(desc_pin.numpy(), dtype=DESC_DTYPE)
        for q, (im, box, fl) in enumerate(zip(images, boxes, flips)):
            H, W = int(im.shape[0]), int(im.shape[1])
            check_box(H, W, box, self.size)
            d[q] = (int(offs[q]), H, W, box[0], box[1], box[2], box[3], 1 if fl else 0, 0)
        with torch.cuda.stream(self.stream):
            for q, im in enumerate(images):
                pool[int(offs[q]):int(offs[q]) + sizes[q]].copy_(im.reshape(-1), non_blocking=True)
            desc_dev.copy_(desc_pin, non_blocking=True)
            ev = self.desc_read[self.k & 1] = self.desc_read[self.k & 1] or torch.cuda.Event()
            ev.record(self.stream)
            preprocess(pool, desc_dev, n, self.size, self.mean, self.std, out, 0, self.stream)
            self.next_target = target.cuda(non_blocking=True)
        self.next_input = out
        self.k += 1

    def __next__(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        if self.stop:
            raise StopIteration
        inp, tgt = self.next_input, self.next_target
        self.preload()
        return inp, tgt

    def __iter__(self):
        return self

    def __len__(self):
        return self.loader_len


class SyntheticDecodedImages(object):
    """A stand-in for the decoded ImageNet samples of the reference's loaders (utils/dataflow.py:173-236; JPEG / LMDB are out of scope):
    `pool_size` uint8 HWC images of ImageNet-like sizes in pinned memory, batches of `batch` samples with the boxes and flips of the
    'imagenet1k_mnas_bilinear' training transform (random.seed(seed) fixes them)."""
    SIZES = [(375, 500), (500, 375), (333, 500), (480, 640), (500, 500), (256, 341), (600, 400), (224, 224)]

    def __init__(self, batch, steps, num_classes=1000, image_size=224, pool_size=64, seed=0, train=True):
        g = torch.Generator().manual_seed(seed)
        self.images = []
        for q in range(pool_size):
            H, W = self.SIZES[q % len(self.SIZES)]
            self.images.append(torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, generator=g).pin_memory())
        self.batch, self.steps, self.num_classes = batch, steps, num_classes
        self.rng_state = random.Random(seed).getstate()
        tr, va = T.mnas_bilinear_transforms(image_size)
        self.crop, self.flip = tr if train else va
        self.g = g

    def __len__(self):
        return self.steps

    def __iter__(self):
        saved = random.getstate()
        random.setstate(self.rng_state)
        try:
            for _ in range(self.steps):
                idx = [random.randrange(len(self.images)) for _ in range(self.batch)]
                imgs = [self.images[i] for i in idx]
                boxes = [self.crop(im) for im in imgs]
                flips = [bool(self.flip()) if self.flip is not None else False for _ in imgs]
                target = torch.randint(0, self.num_classes, (self.batch,), generator=self.g).pin_memory()
                self.rng_state = random.getstate()
                random.setstate(saved)
                yield imgs, boxes, flips, target
                saved = random.getstate()
                random.setstate(self.rng_state)
        finally:
            random.setstate(saved)
