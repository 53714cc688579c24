"""Input pipeline on the GPU: the reference's DataPrefetcher (utils/dataflow.py:13-58) with the per-sample pixel work
moved into a HIP kernel.

    reference:  DataLoader workers: PIL decode -> crop -> resize -> flip -> ToTensor -> Normalize (CPU)  -> prefetcher: H2D of fp32 batches
    here:       loader: decoded uint8 HWC images + (box, flip) decisions (atomnas_amd/utils/transforms.py) -> prefetcher: H2D of the
                uint8 pixels on a side stream (a quarter of the fp32 bytes), atomnas_image_preprocess on that stream -> fp32 NCHW batch

Same iterator protocol as the reference's DataPrefetcher (__iter__ / __next__ / __len__, batch k + 1 in flight while batch k trains,
`torch.cuda.current_stream().wait_stream(side)` at hand-over).  Decoding stays on the host, as in the reference: `dataset: imagenet1k`
(image folders) decodes JPEG / PNG with PIL on loader threads (ImageFolderDecoded); LMDB records cannot be read here (no lmdb module).
Loaders hand over decoded uint8 arrays, e.g. SyntheticDecodedImages below (bench.py --input-pipeline uint8).
"""
import ctypes
import importlib
import random

import numpy as np
import torch

from .. import _lib
from . import transforms as T

DESC_DTYPE = np.dtype([("off", "<i8"), ("H", "<i4"), ("W", "<i4"), ("bi", "<i4"), ("bj", "<i4"), ("bh", "<i4"), ("bw", "<i4"), ("flip", "<i4"),
                       ("pad", "<i4")])   # struct atomnas_img_desc (include/atomnas_hip.h), 40 bytes
MAX_SCALE = 9.0   # the kernel's tap budget: a crop side may be at most 9x the output side


FILTERS = {"bilinear": 0, "bicubic": 1}   # the `filter` argument of atomnas_image_preprocess: PIL's BILINEAR / BICUBIC resamplers


def preprocess(pool_dev, desc_dev, n, size, mean, std, out, out_mode=0, stream=None, filter="bilinear"):
    """launches atomnas_image_preprocess: pool_dev uint8 device tensor, desc_dev uint8 device tensor holding n DESC_DTYPE records"""
    st = ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    _lib.call("atomnas_image_preprocess", ctypes.c_void_p(pool_dev.data_ptr()), ctypes.c_void_p(desc_dev.data_ptr()), int(n), int(size),
              ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
              ctypes.c_void_p(out.data_ptr()), int(out_mode), FILTERS[filter], st)


def check_box(H, W, box, size):
    i, j, h, w = box
    if not (0 <= i and 0 <= j and h > 0 and w > 0 and i + h <= H and j + w <= W):
        raise ValueError("crop box %s outside a %d x %d image" % (box, H, W))
    if h > MAX_SCALE * size or w > MAX_SCALE * size:
        raise ValueError("crop box %s is more than %gx the output size %d" % (box, MAX_SCALE, size))


class DevicePrefetcher(object):
    """DataPrefetcher (utils/dataflow.py:13-58) for decoded uint8 samples.  `loader` yields batches (images, boxes, flips, targets):
    images = list of uint8 HWC tensors (pinned memory makes the copies asynchronous), boxes = list of (top, left, height, width),
    flips = list of bool, targets = int64 tensor.  Yields (input fp32 [N, 3, S, S] on the GPU, target on the GPU).

    The host side of a batch -- drawing it from the loader (the crop / flip decisions of 256 samples), the descriptor table, 256 copy
    submissions: ~7 ms of Python -- runs in a worker thread (threaded=True, default), one batch ahead of the hand-over; the training
    thread only waits for an event.  Two slots (pixel pool, descriptors, output) alternate; a slot is refilled only after the consumer
    of its previous batch has been ordered behind an event on the consumer's stream."""

    def __init__(self, loader, image_size=224, mean=T.IMAGENET_MEAN, std=T.IMAGENET_STD, max_image_bytes=3 * 640 * 640, threaded=True,
                 filter="bilinear"):
        if not torch.cuda.is_available():
            raise _lib.AtomnasHipError("DevicePrefetcher needs the GPU (the preprocessing kernel has no CPU fallback)")
        self.loader_len = len(loader) if hasattr(loader, "__len__") else None
        self.loader = iter(loader)
        self.size, self.mean, self.std = int(image_size), tuple(mean), tuple(std)
        if filter not in FILTERS:
            raise NotImplementedError("resampling filter %r (atomnas_image_preprocess: %s)" % (filter, ", ".join(FILTERS)))
        self.filter = filter
        self.stream = torch.cuda.Stream()
        self.device = torch.cuda.current_device()
        self.max_image_bytes = int(max_image_bytes)
        self.slots = [None, None]   # per slot: (device pool, device descriptors, pinned descriptors, output) sized on first use
        # per slot: event behind the last host-to-device copy that READ the slot's pinned descriptors.  The host rewrites them for the
        # batch after next; nothing else orders the host against that copy (the reference's prefetcher never reuses host staging
        # memory), and a caller that does not synchronise per step (graph replay) runs several steps ahead of the device.
        self.desc_read = [None, None]
        self.consumed = [None, None]   # per slot: event on the consumer's stream behind the use of the slot's previous batch
        self.k = 0                     # batches submitted
        self.handed = 0                # batches handed out
        self.threaded = bool(threaded)
        if self.threaded:
            import queue
            import threading
            self._q = queue.Queue(maxsize=1)
            self._free = [threading.Semaphore(1), threading.Semaphore(1)]
            self._closed = False
            self._thread = threading.Thread(target=self._work, name="atomnas-prefetch", daemon=True)
            self._thread.start()
        else:
            self._pending = self._submit()

    def _slot(self, q, n, nbytes):
        s = self.slots[q]
        if s is None or s[0].numel() < nbytes or s[3].shape[0] != n:
            cap = max(nbytes, n * self.max_image_bytes // 4)
            s = (torch.empty(cap, dtype=torch.uint8, device="cuda"), torch.empty(n * DESC_DTYPE.itemsize, dtype=torch.uint8, device="cuda"),
                 torch.empty(n * DESC_DTYPE.itemsize, dtype=torch.uint8).pin_memory(),
                 torch.empty(n, 3, self.size, self.size, dtype=torch.float32, device="cuda"))
            self.slots[q] = s
        return s

    def _submit(self):
        """draws the next batch and queues its copies and the preprocessing launch on the side stream -> (input, target, ready event)
        or None at the end of the loader"""
        try:
            images, boxes, flips, target = next(self.loader)
        except StopIteration:
            return None
        q = self.k & 1
        n = len(images)
        sizes = [int(im.numel()) for im in images]
        offs = np.concatenate([[0], np.cumsum([(b + 15) // 16 * 16 for b in sizes])])
        pool, desc_dev, desc_pin, out = self._slot(q, n, int(offs[-1]))
        ev = self.desc_read[q]
        if ev is not None:
            ev.synchronize()   # the copy queued two batches ago has read desc_pin (normally long done: no wait in steady state)
        d = np.frombuffer(desc_pin.numpy(), dtype=DESC_DTYPE)
        for i, (im, box, fl) in enumerate(zip(images, boxes, flips)):
            H, W = int(im.shape[0]), int(im.shape[1])
            check_box(H, W, box, self.size)
            d[i] = (int(offs[i]), H, W, box[0], box[1], box[2], box[3], 1 if fl else 0, 0)
        with torch.cuda.stream(self.stream):
            if self.consumed[q] is not None:
                self.stream.wait_event(self.consumed[q])   # the slot's previous batch has been consumed (handed out two batches ago)
            for i, im in enumerate(images):
                pool[int(offs[i]):int(offs[i]) + sizes[i]].copy_(im.reshape(-1), non_blocking=True)
            desc_dev.copy_(desc_pin, non_blocking=True)
            ev = self.desc_read[q] = self.desc_read[q] or torch.cuda.Event()
            ev.record(self.stream)
            preprocess(pool, desc_dev, n, self.size, self.mean, self.std, out, 0, self.stream, self.filter)
            tgt = target.cuda(non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        self.k += 1
        return out, tgt, ready

    def _work(self):
        torch.cuda.set_device(self.device)
        try:
            while not self._closed:
                self._free[self.k & 1].acquire()
                if self._closed:
                    break
                item = self._submit()
                self._q.put(item)
                if item is None:
                    break
        except BaseException as e:   # handed to the consumer, which re-raises it
            self._q.put(e)

    def __next__(self):
        cur = torch.cuda.current_stream()
        if self.handed > 0:
            # everything the caller has queued on its stream so far -- the use of the batch handed out last time included -- is in front
            # of this event: the slot of that batch may be refilled behind it
            q = (self.handed - 1) & 1
            ev = self.consumed[q] = self.consumed[q] or torch.cuda.Event()
            ev.record(cur)
            if self.threaded:
                self._free[q].release()
        if self.threaded:
            item = self._q.get()
            if isinstance(item, BaseException):
                raise item
        else:
            item, self._pending = self._pending, None
        if item is None:
            raise StopIteration
        out, tgt, ready = item
        cur.wait_event(ready)
        self.handed += 1
        if not self.threaded:
            self._pending = self._submit_after_consume()
        return out, tgt

    def _submit_after_consume(self):
        # synchronous mode: the next batch is submitted right away (it fills the OTHER slot; its own slot's consumer event, recorded
        # one hand-over ago, is waited for on the side stream)
        return self._submit()

    def close(self):
        if self.threaded:
            self._closed = True
            for s in self._free:
                s.release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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


# ---------------------------------------------------------------------------------------------- the reference's factories
# data_transforms / dataset / data_loader (utils/dataflow.py:92-267): same names, signatures and FLAGS keys, so that
# `train.py app:<yml>` reaches the GPU input pipeline from the yaml.  What differs is WHERE the pixel work happens: a transform here
# only decides (crop box, flip) per sample, the dataset hands out decoded uint8 images with those decisions, and DevicePrefetcher
# runs crop / resize / flip / ToTensor / Normalize in one kernel per batch.  'imagenet1k' decodes image folders with PIL on loader
# threads; 'imagenet1k_lmdb' raises (no lmdb module in this image).
class DeviceTransform(object):
    """What data_transforms returns per split: the deciders of a transform chain whose pixel work is atomnas_image_preprocess.
    transform(img) -> ((top, left, height, width), flip) for a decoded image (HWC array / tensor, PIL image or (width, height))."""

    def __init__(self, crop, flip, size, mean, std, filter="bilinear"):
        self.crop, self.flip, self.size, self.mean, self.std, self.filter = crop, flip, int(size), tuple(mean), tuple(std), filter

    def __call__(self, img):
        box = self.crop(img)   # random draws in the reference's order: the crop's, then the flip's
        return box, (bool(self.flip()) if self.flip is not None else False)

    def __repr__(self):
        return "DeviceTransform({}, {}, size={})".format(self.crop, self.flip, self.size)


def data_transforms(FLAGS):
    """Get transform of dataset (utils/dataflow.py:92-170) -> (train_transforms, val_transforms, test_transforms)."""
    name = FLAGS.data_transforms
    if name in ('imagenet1k_mnas_bilinear', 'imagenet1k_mnas_bicubic'):
        size = int(FLAGS.get('image_size', 224)) if hasattr(FLAGS, 'get') else 224
        filt = name.rsplit('_', 1)[1]   # Image.BILINEAR / Image.BICUBIC of the reference: the `filter` of atomnas_image_preprocess
        (crop, flip), (vcrop, _) = T.mnas_transforms(size, filt)
        train = DeviceTransform(crop, flip, size, T.IMAGENET_MEAN, T.IMAGENET_STD, filt)
        val = DeviceTransform(vcrop, None, size, T.IMAGENET_MEAN, T.IMAGENET_STD, filt)
        return train, val, val
    if name in ('imagenet1k_basic', 'imagenet1k_inception', 'imagenet1k_mobile'):
        raise NotImplementedError("data_transforms '{}': ColorJitter / Lighting have no device kernel here".format(name))
    try:
        transforms_lib = importlib.import_module(name)
        return transforms_lib.data_transforms()
    except ImportError:
        raise NotImplementedError('Data transform {} is not yet implemented.'.format(name))


class FakeData(object):
    """utils/dataflow.py:61-89: `size` samples of one all-zero image with label 0 (the reference's smoke data source)."""

    def __init__(self, size=1000, image_size=(3, 224, 224), num_classes=10):
        self.size, self.image_size, self.num_classes = size, image_size, num_classes
        self.img = torch.zeros(image_size)
        self.target = 0

    def __getitem__(self, index):
        if index >= len(self):
            raise IndexError("{} index out of range".format(self.__class__.__name__))
        return self.img, self.target

    def __len__(self):
        return self.size


class DecodedFakeData(object):
    """`size` decoded samples for the GPU input pipeline: a pool of uint8 HWC images of ImageNet-like sizes in pinned memory (sample i
    is image i mod pool) with seeded labels; __getitem__ applies the split's DeviceTransform and returns (image, box, flip, target).
    Stand-in for ImageFolder / ImageFolderLMDB + a JPEG decoder, which this image cannot provide."""

    def __init__(self, size, transform, num_classes=1000, pool_size=64, seed=0):
        g = torch.Generator().manual_seed(seed)
        pin = torch.cuda.is_available()
        self.images = []
        for q in range(pool_size):
            H, W = SyntheticDecodedImages.SIZES[q % len(SyntheticDecodedImages.SIZES)]
            im = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, generator=g)
            self.images.append(im.pin_memory() if pin else im)
        self.labels = torch.randint(0, num_classes, (pool_size * 16,), generator=g).tolist()
        self.size, self.transform, self.num_classes = int(size), transform, num_classes

    def __len__(self):
        return self.size

    def load(self, index):
        if index >= self.size:
            raise IndexError("{} index out of range".format(self.__class__.__name__))
        return self.images[index % len(self.images)], self.labels[index % len(self.labels)]

    def __getitem__(self, index):
        im, target = self.load(index)
        box, flip = self.transform(im)
        return im, box, flip, target


IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')   # torchvision.datasets.folder


class ImageFolderDecoded(object):
    """torchvision.datasets.ImageFolder's role (utils/dataflow.py:176-184, `dataset: imagenet1k`) for the GPU input pipeline: samples are
    `root/<class>/<file>` with the classes sorted by name (class index = position) and the files of a class sorted by path; `load(i)`
    decodes with PIL (`Image.open(path).convert('RGB')`, what torchvision's default loader does) into a uint8 HWC tensor (pinned when a
    GPU is present) -- the decode releases the GIL, so DecodedLoader runs it on `data_loader_workers` threads --, `__getitem__` adds the
    split's DeviceTransform decisions: (image, box, flip, target).  The pixel work of the transform stays on the GPU."""

    def __init__(self, root, transform):
        import os
        self.root, self.transform = root, transform
        classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not classes:
            raise FileNotFoundError("Couldn't find any class folder in {}.".format(root))
        self.classes = classes
        self.class_to_idx = {c: i for i, c in enumerate(classes)}
        self.samples = []
        for c in classes:
            for base, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(base, f), self.class_to_idx[c]))
        if not self.samples:
            raise FileNotFoundError("Found no valid file for the classes in {}.".format(root))
        self.pin = torch.cuda.is_available()

    def __len__(self):
        return len(self.samples)

    def load(self, index):
        from PIL import Image
        path, target = self.samples[index]
        with open(path, 'rb') as f:
            arr = np.asarray(Image.open(f).convert('RGB'))
        im = torch.from_numpy(np.ascontiguousarray(arr))
        return (im.pin_memory() if self.pin else im), target

    def __getitem__(self, index):
        im, target = self.load(index)
        box, flip = self.transform(im)
        return im, box, flip, target


def dataset(train_transforms, val_transforms, test_transforms, FLAGS):
    """Get dataset for classification (utils/dataflow.py:173-211) -> (train_set, val_set, test_set)."""
    name = FLAGS.dataset
    if name == 'imagenet1k_fake':
        shape = (3, FLAGS.image_size, FLAGS.image_size)
        return FakeData(size=1281167, image_size=shape, num_classes=1000), FakeData(size=50000, image_size=shape, num_classes=1000), None
    if name == 'imagenet1k_decoded_fake':
        ntrain, nval = int(FLAGS.get('fake_train_size', 1281167)), int(FLAGS.get('fake_val_size', 50000))
        seed = int(FLAGS.get('random_seed', 0))
        train_set = DecodedFakeData(ntrain, train_transforms, seed=seed) if (not FLAGS.get('test_only', False) or FLAGS.get('bn_calibration', False)) else None
        return train_set, DecodedFakeData(nval, val_transforms, seed=seed + 1), None
    if name == 'imagenet1k':
        import os
        train_set = (ImageFolderDecoded(os.path.join(FLAGS.dataset_dir, 'train'), train_transforms)
                     if (not FLAGS.get('test_only', False) or FLAGS.get('bn_calibration', False)) else None)
        return train_set, ImageFolderDecoded(os.path.join(FLAGS.dataset_dir, 'val'), val_transforms), None
    if name == 'imagenet1k_lmdb':
        raise NotImplementedError("dataset 'imagenet1k_lmdb': the lmdb module is not in this image (utils/lmdb_dataset.py:24-71); the same "
                                  "records decode through ImageFolderDecoded's PIL path once they can be read -- use 'imagenet1k' (image folders)")
    try:
        dataset_lib = importlib.import_module(name)
        return dataset_lib.dataset(train_transforms, val_transforms, test_transforms)
    except ImportError:
        raise NotImplementedError('Dataset {} is not yet implemented.'.format(name))


class DecodedLoader(object):
    """torch.utils.data.DataLoader's role for decoded samples (utils/dataflow.py:217-225 `_build_loader`): batches of `batch_size`
    samples (image, box, flip, target) -> (images, boxes, flips, targets int64 pinned), the form DevicePrefetcher takes.  shuffle:
    a fresh seeded permutation per pass; rank / world: the DistributedSampler split (every rank the same number of samples, the
    index list padded by wrapping around); drop_last as torch's.  workers > 0: the samples of a batch are LOADED (decoded) on that many
    threads (PIL's decoder releases the GIL); the transform's random decisions are then drawn in sample order by the iterating thread
    (one stream of Python's `random`, whatever the thread timing) -- with DevicePrefetcher that is its worker thread, off the training
    thread."""

    def __init__(self, dset, batch_size, shuffle, rank=0, world=1, drop_last=False, seed=0, workers=0):
        self.dset, self.batch_size, self.shuffle = dset, int(batch_size), bool(shuffle)
        self.workers = max(0, int(workers))
        self._pool = None
        self.rank, self.world, self.drop_last, self.seed = int(rank), int(world), bool(drop_last), int(seed)
        self.epoch = 0
        n = len(dset)
        self.per_rank = (n + self.world - 1) // self.world

    def __len__(self):
        return self.per_rank // self.batch_size if self.drop_last else (self.per_rank + self.batch_size - 1) // self.batch_size

    def _indices(self):
        n = len(self.dset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        total = self.per_rank * self.world
        idx += idx[:total - n]
        return idx[self.rank:total:self.world]

    def __iter__(self):
        idx = self._indices()
        self.epoch += 1
        pin = torch.cuda.is_available()
        for b in range(len(self)):
            chunk = idx[b * self.batch_size:(b + 1) * self.batch_size]
            if self.workers > 0 and hasattr(self.dset, "load") and hasattr(self.dset, "transform"):
                if self._pool is None:
                    import concurrent.futures
                    self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="atomnas-decode")
                loaded = list(self._pool.map(self.dset.load, chunk))
                samples = []
                for im, tgt in loaded:
                    box, flip = self.dset.transform(im)
                    samples.append((im, box, flip, tgt))
            else:
                samples = [self.dset[i] for i in chunk]
            target = torch.tensor([s[3] for s in samples], dtype=torch.int64)
            yield [s[0] for s in samples], [s[1] for s in samples], [s[2] for s in samples], (target.pin_memory() if pin else target)


def data_loader(train_set, val_set, test_set, FLAGS):
    """Get data loader (utils/dataflow.py:214-267) -> (train_loader, calib_loader, val_loader, test_loader).  `data_loader_workers`:
    decode threads of a loader (at most 16; the reference's worker processes also crop / resize / normalize, which is GPU work here)."""
    if FLAGS.data_loader != 'imagenet1k_basic':
        try:
            data_loader_lib = importlib.import_module(FLAGS.data_loader)
            return data_loader_lib.data_loader(train_set, val_set, test_set)
        except ImportError:
            raise NotImplementedError('Data loader {} is not yet implemented.'.format(FLAGS.data_loader))
    rank = world = None
    if FLAGS.use_distributed:
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
    training = not FLAGS.get('test_only', False)
    calibrating = bool(FLAGS.get('bn_calibration', False))
    seed = int(FLAGS.get('random_seed', 0))
    workers = min(16, int(FLAGS.get('data_loader_workers', 0) or 0))

    def _build_loader(dset, batch_size, shuffle):
        # distributed: the sampler shuffles (DistributedSampler's default) and the loader does not; single process: the loader does
        return DecodedLoader(dset, batch_size, shuffle if world is None else True, rank=rank or 0, world=world or 1,
                             drop_last=FLAGS.get('drop_last', False), seed=seed, workers=workers)

    train_loader = _build_loader(train_set, FLAGS._loader_batch_size, True) if training else None
    calib_loader = _build_loader(train_set, FLAGS.get('_loader_batch_size_calib', FLAGS._loader_batch_size), True) if calibrating else None
    val_loader = _build_loader(val_set, FLAGS._loader_batch_size, False) if world is None else \
        DecodedLoader(val_set, FLAGS._loader_batch_size, True, rank=rank, world=world, drop_last=FLAGS.get('drop_last', False), seed=seed, workers=workers)
    return train_loader, calib_loader, val_loader, val_loader
