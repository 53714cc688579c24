"""Input pipeline, GPU side: atomnas_image_preprocess against PIL-generated fixtures (tests/golden/input_pipeline.pt,
tools/make_golden_input.py) -- the resized uint8 image must equal PIL's byte for byte, the normalized fp32 tensor must equal
torchvision's ToTensor + Normalize arithmetic on it exactly -- and the DevicePrefetcher (the reference's DataPrefetcher,
utils/dataflow.py:13-58) end to end.  JPEG decoding / LMDB are out of scope."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pil_resize as pr  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(images, boxes, flips, size, mean, std, out_mode, filter="bilinear"):
    from atomnas_amd.utils import dataflow as DF
    n = len(images)
    sizes = [int(im.numel()) for im in images]
    offs = np.concatenate([[0], np.cumsum([(b + 15) // 16 * 16 for b in sizes])])
    pool = torch.zeros(int(offs[-1]), dtype=torch.uint8, device="cuda")
    d = np.zeros(n, dtype=DF.DESC_DTYPE)
    for q, (im, box, fl) in enumerate(zip(images, boxes, flips)):
        pool[int(offs[q]):int(offs[q]) + sizes[q]] = im.reshape(-1).cuda()
        DF.check_box(im.shape[0], im.shape[1], box, size)
        d[q] = (int(offs[q]), im.shape[0], im.shape[1], box[0], box[1], box[2], box[3], 1 if fl else 0, 0)
    desc = torch.from_numpy(d.view(np.uint8).copy()).cuda()
    if out_mode == 2:
        out = torch.full((n, size, size, 3), 77, dtype=torch.uint8, device="cuda")
    elif out_mode == 1:
        out = torch.full((n, size, size, 8), 7.0, dtype=torch.bfloat16, device="cuda")
    else:
        out = torch.full((n, 3, size, size), float("nan"), dtype=torch.float32, device="cuda")
    DF.preprocess(pool, desc, n, size, mean, std, out, out_mode, filter=filter)
    torch.cuda.synchronize()
    return out


def test_preprocess_kernel_matches_pil_fixture_exactly(gpu_lib):
    g = torch.load(os.path.join(ROOT, "tests", "golden", "input_pipeline.pt"), weights_only=False)
    by_size = {}
    for c in g["cases"]:
        by_size.setdefault(c["size"], []).append(c)
    for S, cases in by_size.items():
        imgs, boxes, flips = [c["image"] for c in cases], [c["box"] for c in cases], [c["flip"] for c in cases]
        for filt, key in (("bilinear", "resized"), ("bicubic", "resized_bicubic")):
            u8 = _run(imgs, boxes, flips, S, g["mean"], g["std"], 2, filt).cpu()
            f32 = _run(imgs, boxes, flips, S, g["mean"], g["std"], 0, filt).cpu()
            b16 = _run(imgs, boxes, flips, S, g["mean"], g["std"], 1, filt).cpu()
            for q, c in enumerate(cases):
                assert torch.equal(u8[q], c[key]), (filt, S, q, int((u8[q] != c[key]).sum()))          # PIL's bytes
                want = torch.from_numpy(pr.to_tensor_normalize(c[key].numpy(), g["mean"], g["std"]))
                assert torch.equal(f32[q], want), (filt, S, q, float((f32[q] - want).abs().max()))         # ToTensor + Normalize, fp32
                assert torch.equal(b16[q, :, :, :3].permute(2, 0, 1), want.bfloat16()) and float(b16[q, :, :, 3:].abs().max()) == 0.0


def test_preprocess_kernel_random_boxes_against_the_pil_restatement(gpu_lib):
    """boxes drawn by the product's RandomResizedCropPadding on images of ImageNet-like sizes, output 224: byte-identical to the
    oracle's restatement of PIL (itself pinned against PIL in tests/test_input_pipeline.py)"""
    from atomnas_amd.utils import transforms as T
    (crop, flip), _ = T.mnas_bilinear_transforms(224)
    rng = np.random.RandomState(5)
    random.seed(5)
    imgs, boxes, flips = [], [], []
    for (H, W) in [(375, 500), (500, 333), (224, 224), (300, 1200), (90, 70), (640, 480)]:
        im = torch.from_numpy(rng.randint(0, 256, (H, W, 3)).astype(np.uint8))
        imgs.append(im)
        boxes.append(crop(im))
        flips.append(flip())
    for filt in ("bilinear", "bicubic"):
        u8 = _run(imgs, boxes, flips, 224, T.IMAGENET_MEAN, T.IMAGENET_STD, 2, filt).cpu().numpy()
        for q in range(len(imgs)):
            want = pr.crop_resize_flip(imgs[q].numpy(), boxes[q], 224, flips[q], filt)
            assert np.array_equal(u8[q], want), (filt, q, boxes[q], flips[q], int((u8[q] != want).sum()))


def test_device_prefetcher_yields_the_batches_of_its_loader(gpu_lib):
    """DataPrefetcher protocol (utils/dataflow.py:13-58): batches in order, one ahead, StopIteration at the end, len(); every batch
    equals the per-sample PIL restatement; the trainer's static input accepts it (engine.TrainStep.set_batch shape / dtype)."""
    from atomnas_amd.utils import dataflow as DF, transforms as T
    loader = DF.SyntheticDecodedImages(batch=5, steps=3, num_classes=10, image_size=64, pool_size=6, seed=1)
    want = []
    for imgs, boxes, flips, target in loader:
        want.append((torch.stack([torch.from_numpy(pr.to_tensor_normalize(pr.crop_resize_flip(im.numpy(), b, 64, f), T.IMAGENET_MEAN, T.IMAGENET_STD))
                                  for im, b, f in zip(imgs, boxes, flips)]), target.clone()))
    pf = DF.DevicePrefetcher(DF.SyntheticDecodedImages(batch=5, steps=3, num_classes=10, image_size=64, pool_size=6, seed=1), image_size=64)
    assert len(pf) == 3
    got = [(x.clone(), y.clone()) for x, y in pf]
    assert len(got) == 3
    for (x, y), (wx, wy) in zip(got, want):
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (5, 3, 64, 64)
        assert torch.equal(x.cpu(), wx) and torch.equal(y.cpu(), wy)
    with pytest.raises(ValueError):
        DF.check_box(100, 100, (0, 0, 101, 50), 64)
