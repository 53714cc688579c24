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
        self.interpolation = interpolation   # the kernel resizes bilinearly (PIL's BILINEAR): the 'imagenet1k_mnas_bilinear' transform
        self.max_attempts = max_attempts
        self.scale = scale
        self.min_object_covered = min_object_covered or scale[0]
        self.ratio = ratio
        self.log_ratio = log_ratio
        self.crop_padding = crop_padding
        self.center = CenterCropPadding(size if not isinstance(size, tuple) else size[0], crop_padding=crop_padding)

    def get_params(self, img):
        original_width, original_height = _size_of(img)
        original_area = original_width * original_height
        min_area, max_area = [original_area * scale for scale in self.scale]
        for attempt in range(self.max_attempts):
            if self.log_ratio:
                log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
                aspect_ratio = math.exp(random.uniform(*log_ratio))
            else:
                aspect_ratio = random.uniform(*self.ratio)
            min_height = int(round(math.sqrt(min_area / aspect_ratio)))
            max_height = int(round(math.sqrt(max_area / aspect_ratio)))
            if max_height * aspect_ratio > original_width:
                max_height = int((original_width + 0.5 - 0.0000001) / aspect_ratio)
            max_height = min(max_height, original_height)
            min_height = min(max_height, min_height)
            height = random.randint(min_height, max_height)
            width = int(round(height * aspect_ratio))
            assert width <= original_width
            # try to fix rounding errors
            area = height * width
            if area < min_area:
                height += 1
            if area > max_area:
                height -= 1
            width = int(round(height * aspect_ratio))
            area = height * width
            if area < min_area or area > max_area:
                continue
            if area < self.min_object_covered * original_area:
                continue
            if width > original_width or height > original_height or width < 0 or height < 0:
                continue
            if width <= original_width and height <= original_height:
                i = random.randint(0, original_height - height)
                j = random.randint(0, original_width - width)
                return i, j, height, width, True
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


def mnas_bilinear_transforms(image_size=224):
    """the box / flip deciders of data_transforms('imagenet1k_mnas_bilinear') (utils/dataflow.py:125-160): (train, val)"""
    train = (RandomResizedCropPadding(image_size, scale=(0.08, 1.0), min_object_covered=0.1, ratio=(3. / 4., 4. / 3.), log_ratio=False,
                                      crop_padding=MNAS_CROP_PADDING), RandomHorizontalFlip())
    val = (CenterCropPadding(image_size, MNAS_CROP_PADDING), None)
    return train, val
