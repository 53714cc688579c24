#!/usr/bin/env python
"""Generates tests/golden/input_pipeline.pt with PIL in this container (the reference's own pipeline is PIL + torchvision,
utils/dataflow.py:125-160; torchvision is not installed, so the fixture restates F.resized_crop as PIL's crop + resize, which is what
torchvision calls): uint8 images, crop boxes, flips, and PIL's resized uint8 results -- data only.

    python tools/make_golden_input.py
"""
import os
import random

import numpy as np
import torch
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "input_pipeline.pt")


def image(rng, H, W, kind):
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == 0:      # noise: every tap matters
        return rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    if kind == 1:      # smooth gradients with saturation at both ends
        return np.stack([np.clip(yy * 255 // max(H - 1, 1) * 2 - 100, 0, 255), (xx * 3) % 256, np.clip(255 - (yy + xx), 0, 255)], 2).astype(np.uint8)
    return ((np.sin(yy / 3.0)[..., None] * np.cos(xx / 5.0)[..., None] * 127 + 128) + rng.randint(-20, 20, (H, W, 3))).clip(0, 255).astype(np.uint8)


def main():
    rng = np.random.RandomState(12)
    random.seed(12)
    cases = []
    # (H, W, box or None = whole image, output size, flip)
    spec = [(375, 500, (30, 41, 300, 410), 224, False), (250, 188, (0, 0, 250, 188), 112, True), (240, 320, (17, 3, 199, 251), 224, True),
            (224, 224, (0, 0, 224, 224), 224, False), (150, 130, (20, 10, 100, 111), 224, True),      # up-scaling
            (320, 240, (32, 0, 256, 240), 32, False), (97, 61, (5, 7, 80, 33), 32, True), (288, 288, (0, 0, 288, 288), 32, True),  # 9x down
            (167, 250, (1, 2, 151, 201), 112, False)]
    for q, (H, W, box, S, flip) in enumerate(spec):
        img = image(rng, H, W, q % 3)
        i, j, h, w = box
        res = {}
        for name, filt in (("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC)):   # data_transforms 'imagenet1k_mnas_bilinear' / '_bicubic'
            r = Image.fromarray(img).crop((j, i, j + w, i + h)).resize((S, S), filt)
            if flip:
                r = r.transpose(Image.FLIP_LEFT_RIGHT)
            res[name] = torch.from_numpy(np.asarray(r).copy())
        cases.append(dict(image=torch.from_numpy(img), box=box, size=S, flip=flip, resized=res["bilinear"], resized_bicubic=res["bicubic"]))
    import PIL
    torch.save(dict(cases=cases, pil_version=PIL.__version__, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)), OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes, PIL", PIL.__version__)


if __name__ == "__main__":
    main()
