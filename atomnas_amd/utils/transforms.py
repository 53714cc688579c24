"""Crop parameters of the reference's MnasNet-style input transforms (utils/transforms.py:54-177), restated as plain integer / float
host logic.  The reference applies them to PIL images through torchvision (`F.center_crop`, `F.resized_crop`); here the transforms
only DECIDE (crop box, flip) -- the pixel work (crop, PIL-exact bilinear resize, flip, ToTensor, Normalize) is the HIP kernel
atomnas_image_preprocess, fed by atomnas_amd/utils/dataflow.py.  Same class names, constructor arguments and random-number
consumption (Python's `random`, in the reference's order), so a seeded run draws the same boxes.

torchvision is not installed in this image, so the reference module cannot be imported; the known answers of tests/test_input_pipeline.py
are computed from the reference's formulas transcribed there, independently of this file.
"""
import math
import random


def _size_of(img):
    """(width, height) of a PIL image, a (width, height) tuple, or an HWC array"""
    if hasattr(img, "size") and not hasattr(img, "shape"):
        return img.size
    if hasattr(img, "shape"):
        return int(img.shape[1]), int(img.shape[0])
    w, h = img
    return int(w), int(h)


def _interp_name(interpolation):
    """'bilinear' / 'bicubic' / ... for a PIL resampling constant (int or enum), a torchvision InterpolationMode or a string"""
    pil = {0: "nearest", 1: "lanczos", 2: "bilinear", 3: "bicubic", 4: "box", 5: "hamming"}   # PIL.Image resampling filters
    if isinstance(interpolation, str):
        return interpolation.lower()
    name = getattr(interpolation, "name", None)
    if isinstance(name, str):
        return name.lower()
    try:
        return pil.get(int(interpolation), str(interpolation))
    except (TypeError, ValueError):
        return str(interpolation)


def center_crop_box(width, height, crop_h, crop_w):
    """torchvision.transforms.functional.center_crop's box (top, left, height, width) for a crop that fits inside the image"""
    top = int(round((height - crop_h) / 2.0))
    left = int(round((width - crop_w) / 2.0))
    return top, left, crop_h, crop_w


class CenterCropPadding(object):
    """Tensorflow style `CenterCrop` (utils/transforms.py:54-76): a centred square of side int(size / (size + crop_padding) * min(w, h))."""

    def __init__(self, size, crop_padding=0):
        self.size = size
        self.crop_padding = crop_padding

    def get_box(self, img):
        width, height = _size_of(img)
        side = int(self.size / (self.size + self.crop_padding) * min(width, height))
        return center_crop_box(width, height, side, side)

    __call__ = get_box

    def __repr__(self):
        return self.__class__.__name__ + '(size={0}, crop_padding={1})'.format(self.size, self.crop_padding)


class RandomResizedCropPadding(object):
    """Tensorflow style `RandomResizedCrop` (utils/transforms.py:79-177).  get_params draws exactly what the reference draws
    (random.uniform for the aspect ratio, random.randint for height, top, left, up to max_attempts times) and returns
    (i, j, h, w, success); __call__ returns the box the reference would crop: the drawn one, or CenterCropPadding's on failure."""

    def __init__(self, size, scale=(0.08, 1.0), min_object_covered=None, ratio=(3. / 4., 4. / 3.), log_ratio=True, interpolation=None,
                 max_attempts=10, crop_padding=0):
        self.size = size if isinstance(size, tuple) else (size, size)
        assert (scale[0] < scale[1]) and (ratio[0] < ratio[1])
        # the kernel implements PIL's BILINEAR and BICUBIC resamplers ('imagenet1k_mnas_bilinear' / 'imagenet1k_mnas_bicubic', the latter
        # the default of apps/mobilenet/default_mnas_scheduler.yml); asking for another filter is an error, not a silent substitution
        if interpolation is not None and _interp_name(interpolation) not in ("bilinear", "bicubic"):
            raise NotImplementedError("atomnas_image_preprocess resizes with PIL's BILINEAR or BICUBIC filter only (got %r)" % (interpolation,))
        self.interpolation = interpolation
        self.filter = _interp_name(interpolation) if interpolation is not None else "bilinear"
        self.max_attempts = max_attempts
        self.scale = scale
        self.min_object_covered = min_object_covered or scale[0]
        self.ratio = ratio
        self.log_ratio = log_ratio
        self.crop_padding = crop_padding
        self.center = CenterCropPadding(size if not isinstance(size, tuple) else size[0], crop_padding=crop_padding)

    # ---- one attempt of the reference's sampler (utils/transforms.py:117-160), split into its three decisions.  What the drop-in
    # contract fixes is the ORDER and KIND of the random draws (uniform -> randint height -> randint top -> randint left) and the
    # integer arithmetic between them; tests/test_input_pipeline.py pins both against a transcription of the reference's formulas.
    def _draw_ratio(self):
        lo, hi = self.ratio
        if self.log_ratio:
            return math.exp(random.uniform(math.log(lo), math.log(hi)))
        return random.uniform(lo, hi)

    @staticmethod
    def _height_bounds(ratio, img_w, img_h, area_lo, area_hi):
        """heights (inclusive) whose crop of this aspect ratio has an area in [area_lo, area_hi] and fits the image"""
        tallest = int(round(math.sqrt(area_hi / ratio)))
        if tallest * ratio > img_w:                       # too wide at that height: the tallest crop that still fits across
            tallest = int((img_w + 0.5 - 0.0000001) / ratio)
        tallest = min(tallest, img_h)
        shortest = min(tallest, int(round(math.sqrt(area_lo / ratio))))
        return shortest, tallest

    @staticmethod
    def _nudge(h, ratio, area_lo, area_hi):
        """rounding of width = round(h * ratio) can push the area just outside the range: one row more / less, judged on the area of
        the height as drawn (both tests look at that same area)"""
        drawn_area = h * int(round(h * ratio))
        if drawn_area < area_lo:
            h += 1
        if drawn_area > area_hi:
            h -= 1
        return h, int(round(h * ratio))

    def get_params(self, img):
        img_w, img_h = _size_of(img)
        whole = img_w * img_h
        area_lo, area_hi = whole * self.scale[0], whole * self.scale[1]
        covered = self.min_object_covered * whole
        for _ in range(self.max_attempts):
            ratio = self._draw_ratio()
            shortest, tallest = self._height_bounds(ratio, img_w, img_h, area_lo, area_hi)
            h = random.randint(shortest, tallest)
            assert int(round(h * ratio)) <= img_w   # the reference asserts the drawn crop's width before adjusting it
            h, w = self._nudge(h, ratio, area_lo, area_hi)
            area = h * w
            fits = 0 <= w <= img_w and 0 <= h <= img_h
            if not (area_lo <= area <= area_hi) or area < covered or not fits:
                continue
            top = random.randint(0, img_h - h)
            left = random.randint(0, img_w - w)
            return top, left, h, w, True
        return None, None, None, None, False

    def __call__(self, img):
        i, j, h, w, success = self.get_params(img)
        if success:
            return i, j, h, w
        return self.center.get_box(img)

    def __repr__(self):
        return (self.__class__.__name__ + '(size={0}, scale={1}, ratio={2}, crop_padding={3})'.format(
            self.size, tuple(round(s, 4) for s in self.scale), tuple(round(r, 4) for r in self.ratio), self.crop_padding))


class RandomHorizontalFlip(object):
    """flip decision with probability p (torchvision draws from torch's generator; here Python's `random`, like the crops)"""

    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img=None):
        return random.random() < self.p


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # utils/dataflow.py:131-132
MNAS_CROP_PADDING = 32                                                         # utils/dataflow.py:133


def mnas_transforms(image_size=224, filter="bilinear"):
    """the box / flip deciders of data_transforms('imagenet1k_mnas_bilinear' | 'imagenet1k_mnas_bicubic') (utils/dataflow.py:125-160):
    (train, val); the filter only travels with them (the pixel work is the kernel's)"""
    train = (RandomResizedCropPadding(image_size, scale=(0.08, 1.0), min_object_covered=0.1, ratio=(3. / 4., 4. / 3.), log_ratio=False,
                                      interpolation=filter, crop_padding=MNAS_CROP_PADDING), RandomHorizontalFlip())
    val = (CenterCropPadding(image_size, MNAS_CROP_PADDING), None)
    return train, val


def mnas_bilinear_transforms(image_size=224):
    return mnas_transforms(image_size, "bilinear")
