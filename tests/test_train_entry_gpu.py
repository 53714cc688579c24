"""`python train.py app:<yml>` end to end on the GPU (reference entry point train.py:310-330): two shortened epochs of the
AtomNAS-C search -- graph-captured training steps, per-epoch validation of the EMA model with BN calibration, mask /
pruned-MACs accounting, the final shrink, checkpoint + exported architecture."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_entry_runs_two_epochs(gpu_lib, tmp_path):
    env = dict(os.environ, ATOMNAS_E2E_DIR=str(tmp_path), ARNOLD_OUTPUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "app:" + os.path.join(ROOT, "tests", "data", "tiny_search.yml")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count(" val: ") >= 2, out[-4000:]
    assert "Prune threshold" in out
    assert os.path.exists(os.path.join(str(tmp_path), "latest_checkpoint.pt"))
    assert os.path.exists(os.path.join(str(tmp_path), "latest_checkpoint.yml"))
    # resume from the checkpoint for one more epoch (train.py:181-196 of the reference: model, optimizer, EMA, epoch counters)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "app:" + os.path.join(ROOT, "tests", "data", "tiny_search.yml"),
                         "--resume", str(tmp_path), "--num_epochs", "3"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out2 = r2.stdout + r2.stderr
    assert r2.returncode == 0, out2[-4000:]
    assert "Epoch 2/3" in out2 and "Epoch 0/3" not in out2, out2[-3000:]


def test_train_entry_on_the_gpu_input_pipeline(gpu_lib, tmp_path):
    """the same entry fed by `dataset: imagenet1k_decoded_fake`: data_transforms('imagenet1k_mnas_bicubic') / dataset / data_loader
    (the reference's factories, utils/dataflow.py:92-267) -> DevicePrefetcher -> TrainStep.set_batch; calibration and validation
    batches through the same pipeline"""
    env = dict(os.environ, ATOMNAS_E2E_DIR=str(tmp_path), ARNOLD_OUTPUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "app:" + os.path.join(ROOT, "tests", "data", "tiny_search_decoded.yml")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count(" val: ") >= 2 and "Prune threshold" in out, out[-4000:]
    assert out.count(" step ") >= 6, out[-4000:]   # 2 epochs x 3 steps were trained on pipeline batches
    assert os.path.exists(os.path.join(str(tmp_path), "latest_checkpoint.pt"))


def test_train_entry_on_an_image_folder(gpu_lib, tmp_path):
    """`dataset: imagenet1k` from the yaml: JPEG / PNG files under dataset_dir/{train,val}/<class>/ decoded with PIL on loader threads
    (as the reference's DataLoader workers do), everything after the decode on the GPU"""
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(1)
    data = tmp_path / "data"
    for split, per_class in (("train", 12), ("val", 6)):
        for c in ("n01440764", "n01443537", "n01484850"):
            d = data / split / c
            d.mkdir(parents=True)
            for k in range(per_class):
                H, W = int(rng.randint(60, 200)), int(rng.randint(60, 200))
                Image.fromarray(rng.randint(0, 256, (H, W, 3)).astype(np.uint8)).save(str(d / ("%03d.JPEG" % k)), quality=92)
    out = tmp_path / "out"
    env = dict(os.environ, ATOMNAS_E2E_DIR=str(out), ARNOLD_OUTPUT=str(out))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "app:" + os.path.join(ROOT, "tests", "data", "tiny_search_decoded.yml"),
                        "--dataset", "imagenet1k", "--dataset_dir", str(data), "--data_loader_workers", "4"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    o = r.stdout + r.stderr
    assert r.returncode == 0, o[-4000:]
    assert o.count(" val: ") >= 2 and o.count(" step ") >= 6, o[-4000:]
