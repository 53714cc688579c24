"""Input pipeline, CPU side: the crop-parameter logic of the reference's transforms (utils/transforms.py:54-177)
restated in atomnas_amd/utils/transforms.py, and the oracle's restatement of PIL's bilinear resize (oracle/pil_resize.py) pinned
against PIL itself and against the committed fixture.

torchvision is not installed in this image, so the reference's module cannot be imported: the known answers below come from the
reference's FORMULAS, transcribed here independently of the product file (every line cites the reference line it restates)."""
import math
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pil_resize as pr  # noqa: E402

from atomnas_amd.utils import transforms as T  # noqa: E402


def ref_center_crop_padding(width, height, size, crop_padding):
    side = int(size / (size + crop_padding) * min(width, height))          # utils/transforms.py:69-71
    top = int(round((height - side) / 2.0))                                  # torchvision F.center_crop
    left = int(round((width - side) / 2.0))
    return top, left, side, side


def ref_get_params(rng, ow, oh, scale, min_cov, ratio, log_ratio, max_attempts):
    """utils/transforms.py:117-160, with `rng` in place of the module-level `random`"""
    area0 = ow * oh
    min_area, max_area = area0 * scale[0], area0 * scale[1]
    for _ in range(max_attempts):
        ar = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1]))) if log_ratio else rng.uniform(ratio[0], ratio[1])   # :122-126
        min_h = int(round(math.sqrt(min_area / ar)))                                                                          # :128
        max_h = int(round(math.sqrt(max_area / ar)))                                                                          # :129
        if max_h * ar > ow:                                                                                                     # :130-132
            max_h = int((ow + 0.5 - 0.0000001) / ar)
        max_h = min(max_h, oh)                                                                                                  # :133
        min_h = min(max_h, min_h)                                                                                               # :134
        h = rng.randint(min_h, max_h)                                                                                           # :135
        w = int(round(h * ar))                                                                                                  # :136
        a = h * w                                                                                                               # :140
        if a < min_area:
            h += 1
        if a > max_area:
            h -= 1
        w = int(round(h * ar))                                                                                                  # :145
        a = h * w
        if a < min_area or a > max_area or a < min_cov * area0:                                                                 # :148-151
            continue
        if w > ow or h > oh or w < 0 or h < 0:                                                                                  # :152-154
            continue
        return rng.randint(0, oh - h), rng.randint(0, ow - w), h, w, True                                                     # :157-159
    return None, None, None, None, False


def test_center_crop_padding_known_answers():
    # by hand: int(224 / 256 * 375) = 328; (500 - 328) / 2 = 86; (375 - 328) / 2 = 23.5 -> round-half-even 24
    assert T.CenterCropPadding(224, 32)((500, 375)) == (24, 86, 328, 328)
    assert T.CenterCropPadding(224, 0)((224, 224)) == (0, 0, 224, 224)
    for (w, h) in [(500, 375), (375, 500), (333, 500), (640, 480), (100, 37), (1, 1), (4032, 3024)]:
        for size, pad in [(224, 32), (224, 0), (192, 32), (299, 40)]:
            assert T.CenterCropPadding(size, pad)((w, h)) == ref_center_crop_padding(w, h, size, pad)


@pytest.mark.parametrize("log_ratio", [False, True])
def test_random_resized_crop_padding_draws_the_reference_boxes(log_ratio):
    """same seed -> the same boxes as the reference's formulas, for the 'imagenet1k_mnas_bilinear' settings and a few others; the
    module-level random stream is consumed exactly as the reference consumes it (the next draw after a call agrees as well)"""
    sizes = [(500, 375), (375, 500), (640, 480), (333, 500), (224, 224), (100, 37), (37, 100), (2000, 30)]
    for scale, cov, ratio in [((0.08, 1.0), 0.1, (3. / 4., 4. / 3.)), ((0.25, 1.0), None, (0.5, 2.0)), ((0.9, 1.0), 0.95, (0.99, 1.01))]:
        t = T.RandomResizedCropPadding(224, scale=scale, min_object_covered=cov, ratio=ratio, log_ratio=log_ratio, crop_padding=32)
        n_fail = 0
        for seed in range(40):
            for (w, h) in sizes:
                random.seed(seed * 7 + w)
                got = t.get_params((w, h))
                nxt = random.random()
                rng = random.Random(seed * 7 + w)
                want = ref_get_params(rng, w, h, scale, cov or scale[0], ratio, log_ratio, 10)
                assert got == want, (seed, w, h, got, want)
                assert nxt == rng.random()
                if want[4]:
                    i, j, hh, ww = want[:4]
                    assert 0 <= i and 0 <= j and i + hh <= h and j + ww <= w
                    assert t.__class__(224, scale=scale, min_object_covered=cov, ratio=ratio, log_ratio=log_ratio, crop_padding=32) is not None
                else:
                    n_fail += 1
                    random.seed(seed * 7 + w)
                    assert t((w, h)) == ref_center_crop_padding(w, h, 224, 32)    # utils/transforms.py:162-168: the fall-back crop
        assert n_fail > 0 or scale[0] < 0.5   # the extreme aspect ratios do exercise the fall-back


def test_mnas_transform_settings():
    (crop, flip), (vcrop, vflip) = T.mnas_bilinear_transforms(224)
    assert crop.scale == (0.08, 1.0) and crop.min_object_covered == 0.1 and crop.ratio == (3. / 4., 4. / 3.) and crop.log_ratio is False
    assert crop.crop_padding == 32 and vcrop.crop_padding == 32 and vflip is None and flip.p == 0.5     # utils/dataflow.py:125-160
    assert T.IMAGENET_MEAN == (0.485, 0.456, 0.406) and T.IMAGENET_STD == (0.229, 0.224, 0.225)


def test_pil_resize_restatement_is_bit_identical_to_pil():
    """oracle/pil_resize.py against PIL itself on random images, boxes and output sizes (up- and down-scaling up to 9x)"""
    from PIL import Image
    rng = np.random.RandomState(3)
    n = 0
    for t in range(30):
        H, W = int(rng.randint(8, 260)), int(rng.randint(8, 260))
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        h, w = int(rng.randint(4, H + 1)), int(rng.randint(4, W + 1))
        i, j = int(rng.randint(0, H - h + 1)), int(rng.randint(0, W - w + 1))
        S = int(rng.choice([17, 32, 64, 224]))
        if h > 9 * S or w > 9 * S:
            continue
        for filt, pf in ((pr.BILINEAR, Image.BILINEAR), (pr.BICUBIC, Image.BICUBIC)):
            ref = np.asarray(Image.fromarray(img).crop((j, i, j + w, i + h)).resize((S, S), pf))
            assert np.array_equal(pr.resize_u8(img[i:i + h, j:j + w], S, S, filt), ref), (t, filt, H, W, (i, j, h, w), S)
        n += 1
    assert n >= 25


def test_pil_resize_restatement_matches_the_committed_fixture():
    g = torch.load(os.path.join(ROOT, "tests", "golden", "input_pipeline.pt"), weights_only=False)
    assert len(g["cases"]) >= 8
    for c in g["cases"]:
        got = pr.crop_resize_flip(c["image"].numpy(), c["box"], c["size"], c["flip"])
        assert np.array_equal(got, c["resized"].numpy()), (c["box"], c["size"], c["flip"])
        gotc = pr.crop_resize_flip(c["image"].numpy(), c["box"], c["size"], c["flip"], pr.BICUBIC)
        assert np.array_equal(gotc, c["resized_bicubic"].numpy()), ("bicubic", c["box"], c["size"], c["flip"])
        t = pr.to_tensor_normalize(got, g["mean"], g["std"])
        assert t.dtype == np.float32 and t.shape == (3, c["size"], c["size"])


# ---- the reference's factories (utils/dataflow.py:92-267) over decoded sources: host logic, no GPU
class _Flags(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _flags(**kw):
    f = _Flags(data_transforms="imagenet1k_mnas_bilinear", dataset="imagenet1k_decoded_fake", data_loader="imagenet1k_basic", image_size=224,
               use_distributed=False, test_only=False, bn_calibration=True, fake_train_size=50, fake_val_size=11, random_seed=3,
               _loader_batch_size=8, _loader_batch_size_calib=4, data_loader_workers=62)
    f.update(kw)
    return f


def test_data_factories_have_the_reference_protocol():
    from atomnas_amd.utils import dataflow as DF
    F = _flags()
    tr, va, te = DF.data_transforms(F)
    assert va is te and tr.size == 224 and tr.mean == T.IMAGENET_MEAN and va.flip is None
    train_set, val_set, test_set = DF.dataset(tr, va, te, F)
    assert (len(train_set), len(val_set), test_set) == (50, 11, None)
    train_loader, calib_loader, val_loader, test_loader = DF.data_loader(train_set, val_set, test_set, F)
    assert (len(train_loader), len(calib_loader), len(val_loader)) == (7, 13, 2) and test_loader is val_loader
    random.seed(5)
    batches = list(train_loader)
    assert [len(b[0]) for b in batches] == [8] * 6 + [2]
    images, boxes, flips, target = batches[0]
    assert images[0].dtype == torch.uint8 and images[0].dim() == 3 and target.dtype == torch.int64 and len(boxes) == len(flips) == 8
    for im, (i, j, h, w) in zip(images, boxes):
        assert 0 <= i and 0 <= j and i + h <= im.shape[0] and j + w <= im.shape[1]
    # validation: centre crops, no flips, dataset order
    vb = list(val_loader)
    assert not any(vb[0][2]) and [len(b[0]) for b in vb] == [8, 3]
    assert vb[0][1][0] == T.CenterCropPadding(224, 32).get_box(vb[0][0][0])
    # drop_last, and the DistributedSampler split: every rank the same number of samples, together they cover the set
    F2 = _flags(drop_last=True)
    assert len(DF.data_loader(train_set, val_set, None, F2)[0]) == 6
    seen = []
    for r in range(3):
        ld = DF.DecodedLoader(val_set, 2, False, rank=r, world=3)
        got = [t for b in ld for t in b[3].tolist()]
        assert len(got) == 4
        seen.append(got)
    labels = [val_set[i][3] for i in range(11)]
    assert sorted(sum(seen, [])) == sorted(labels + labels[:1])


def test_data_factories_refuse_what_this_image_cannot_do():
    from atomnas_amd.utils import dataflow as DF
    tb = DF.data_transforms(_flags(data_transforms="imagenet1k_mnas_bicubic"))   # the reference's default transform (default_mnas_scheduler.yml)
    assert tb[0].filter == "bicubic" and tb[1].filter == "bicubic" and DF.data_transforms(_flags())[0].filter == "bilinear"
    with pytest.raises(NotImplementedError):
        DF.data_transforms(_flags(data_transforms="imagenet1k_basic"))
    with pytest.raises(NotImplementedError, match="not yet implemented"):
        DF.data_transforms(_flags(data_transforms="no_such_module_xyz"))
    with pytest.raises(NotImplementedError, match="lmdb"):
        DF.dataset(None, None, None, _flags(dataset="imagenet1k_lmdb"))
    with pytest.raises(NotImplementedError, match="not yet implemented"):
        DF.data_loader(None, None, None, _flags(data_loader="no_such_loader_xyz"))
    with pytest.raises(NotImplementedError, match="BILINEAR"):
        T.RandomResizedCropPadding(224, interpolation=1)   # PIL.Image.LANCZOS: no kernel
    assert T.RandomResizedCropPadding(224, interpolation=3).filter == "bicubic" and T.RandomResizedCropPadding(224, interpolation=2).filter == "bilinear"
    fake = DF.dataset(None, None, None, _flags(dataset="imagenet1k_fake"))   # the reference's zero-image smoke source keeps its form
    assert len(fake[0]) == 1281167 and fake[0][0][0].shape == (3, 224, 224) and fake[0][0][1] == 0


def _make_image_folder(root, classes, per_class, seed=0):
    """root/{train,val}/<class>/<k>.jpg|png written with PIL: what torchvision's ImageFolder reads"""
    from PIL import Image
    rng = np.random.RandomState(seed)
    for split in ("train", "val"):
        for c in classes:
            d = os.path.join(root, split, c)
            os.makedirs(d)
            for k in range(per_class):
                H, W = int(rng.randint(40, 90)), int(rng.randint(40, 90))
                img = Image.fromarray(rng.randint(0, 256, (H, W, 3)).astype(np.uint8))
                img.save(os.path.join(d, "%02d.%s" % (k, "png" if k % 3 == 0 else "jpg")), quality=90)
        with open(os.path.join(root, split, "notes.txt"), "w") as f:   # stray files at the class level are ignored
            f.write("x")


def test_image_folder_dataset_decodes_like_the_reference_loader(tmp_path):
    """`dataset: imagenet1k` (utils/dataflow.py:176-184: torchvision ImageFolder over dataset_dir/train and /val): classes sorted by name,
    files sorted inside a class, decoded with PIL to RGB; samples carry the split's crop / flip decisions; decode on loader threads
    gives the same batches as the sequential form (the random decisions are drawn in sample order by one thread)."""
    from PIL import Image
    from atomnas_amd.utils import dataflow as DF
    root = str(tmp_path)
    _make_image_folder(root, ["n02", "n01", "n10"], 4)
    F = _flags(dataset="imagenet1k", dataset_dir=root, _loader_batch_size=5, bn_calibration=False, data_loader_workers=3)
    tr, va, te = DF.data_transforms(F)
    train_set, val_set, _ = DF.dataset(tr, va, te, F)
    assert train_set.classes == ["n01", "n02", "n10"] and len(train_set) == 12 and len(val_set) == 12
    assert [t for _, t in train_set.samples] == [0] * 4 + [1] * 4 + [2] * 4
    assert [os.path.basename(p_) for p_, _ in train_set.samples[:4]] == ["00.png", "01.jpg", "02.jpg", "03.png"]
    im, box, flip, target = val_set[5]
    want = np.asarray(Image.open(val_set.samples[5][0]).convert("RGB"))
    assert im.dtype == torch.uint8 and np.array_equal(im.numpy(), want) and target == 1
    assert box == T.CenterCropPadding(224, 32).get_box(im) and flip is False
    loaders = DF.data_loader(train_set, val_set, None, F)
    assert loaders[0].workers == 3 and len(loaders[0]) == 3
    random.seed(9)
    threaded = list(loaders[0])
    seq_loader = DF.DecodedLoader(train_set, 5, True, seed=F.random_seed, workers=0)
    random.seed(9)
    sequential = list(seq_loader)
    assert [len(b[0]) for b in threaded] == [5, 5, 2]
    for a, b in zip(threaded, sequential):
        assert all(torch.equal(x, y) for x, y in zip(a[0], b[0])) and a[1] == b[1] and a[2] == b[2] and torch.equal(a[3], b[3])
    with pytest.raises(FileNotFoundError):
        DF.ImageFolderDecoded(os.path.join(root, "train", "n01"), tr)   # no class folders below
