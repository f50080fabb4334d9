"""File-backed datasets (reference: src/pytorch/{CNN,LSTM,MLP}/dataset.py) on tiny generated fixtures.

Parity unpinned against the reference's own data: /data/PCB_DATASET etc. do not exist here and torchvision /
libxml2 are not installed, so the checks are against the documented sample semantics."""
import os

import numpy as np
import pytest
import torch

from distributed_deep_learning_amd.data import files


def _make_pcb(root, classes=("missing_hole", "short"), per_class=2):
    from PIL import Image
    rng = np.random.default_rng(0)
    for ci, cls in enumerate(classes):
        os.makedirs(os.path.join(root, "images", cls))
        os.makedirs(os.path.join(root, "Annotations", cls))
        for i in range(per_class):
            arr = rng.integers(0, 255, size=(120, 160, 3), dtype=np.uint8)
            arr[40:80, 50:110] = 200 + ci * 20                     # the "defect": a flat patch
            Image.fromarray(arr).save(os.path.join(root, "images", cls, f"b{i}.jpg"), quality=95)
            boxes = "".join(
                f"<object><name>{cls}</name><bndbox><xmin>{x0}</xmin><ymin>{y0}</ymin><xmax>{x1}</xmax><ymax>{y1}</ymax>"
                "</bndbox></object>" for (x0, y0, x1, y1) in [(45, 35, 95, 65), (100, 90, 158, 118)][: i + 1])
            with open(os.path.join(root, "Annotations", cls, f"b{i}.xml"), "w") as f:
                f.write(f"<annotation><filename>b{i}.jpg</filename>{boxes}</annotation>")


def test_read_boxes_and_instances(tmp_path):
    _make_pcb(str(tmp_path))
    classes, inst = files.list_pcb_instances(str(tmp_path))
    assert classes == ["missing_hole", "short"]
    assert len(inst) == 2 * (1 + 2)                                  # image i carries i+1 boxes
    assert inst[0][1] == (45, 35, 95, 65) and inst[0][2] == 0
    assert inst[-1][2] == 1


def test_pcb_dataset_shapes_and_semantics(tmp_path):
    _make_pcb(str(tmp_path))
    ds = files.pcb_defects(str(tmp_path), seed=3)
    assert len(ds) == 12                                             # every box twice (reference dataset.py:92)
    x, y = ds[0]
    assert x.shape == (3, 64, 64) and x.dtype == torch.float32
    assert 0.0 <= float(x.min()) and float(x.max()) <= 255.0 and float(x.max()) > 1.0     # 0..255, not 0..1
    assert y.shape == (2,) and float(y.sum()) == 1.0
    assert ds.y[:6].argmax(1).tolist() == [0] * 6 and ds.y[6:].argmax(1).tolist() == [1] * 6
    # the crop of the first box sits (shifted by 5..10) on the flat patch: mostly the patch value
    assert abs(float(x.median()) - 200.0) < 12.0
    xb, yb = ds.batch(torch.tensor([0, 5, 11]))
    assert xb.shape == (3, 3, 64, 64) and yb.shape == (3, 2)


def test_crop_outside_image_is_zero_padded():
    img = torch.full((3, 20, 20), 100, dtype=torch.uint8)
    out = files.crop_resize(img, top=10, left=10, height=20, width=20, size=8)
    assert out.shape == (3, 8, 8)
    assert float(out[:, :3, :3].mean()) > 90.0 and float(out[:, 5:, 5:].abs().max()) < 1e-3


def test_missing_class_images_raise(tmp_path):
    _make_pcb(str(tmp_path))
    os.makedirs(os.path.join(str(tmp_path), "Annotations", "spur"))
    with pytest.raises(FileNotFoundError):
        files.list_pcb_instances(str(tmp_path))


def test_window_csv(tmp_path):
    machines, rows, feat = 3, 20, 4
    data = np.arange(machines * rows * (feat + 5), dtype=np.float32).reshape(machines * rows, feat + 5)
    path = os.path.join(str(tmp_path), "pm.csv")
    np.savetxt(path, data, delimiter=",", header=",".join(f"c{i}" for i in range(feat + 5)), comments="", fmt="%.1f")
    ds = files.window_csv(path, history=10, rows_per_machine=rows)
    assert len(ds) == machines * (rows - 9)                          # reference LSTM/dataset.py:31-33
    x, y = ds[0]
    assert x.shape == (10, feat) and y.shape == (5,)
    assert torch.equal(x, torch.from_numpy(data[0:10, :feat]))
    assert torch.equal(y, torch.from_numpy(data[0, feat:]))          # targets of the window's FIRST row (:45)
    # windows never span machines: the first window of machine 1 starts at its row 0
    x1, _ = ds[rows - 9]
    assert torch.equal(x1, torch.from_numpy(data[rows:rows + 10, :feat]))
    xb, yb = ds.batch(torch.tensor([0, rows - 9]))
    assert torch.equal(xb[1], x1) and yb.shape == (2, 5)
    with pytest.raises(ValueError):
        files.window_csv(path, rows_per_machine=7)


def test_tabular_csv(tmp_path):
    data = np.random.default_rng(1).random((17, 1 + 48 + 5)).astype(np.float32)
    path = os.path.join(str(tmp_path), "mqtt.csv")
    np.savetxt(path, data, delimiter=",", header=",".join(f"c{i}" for i in range(data.shape[1])), comments="")
    ds = files.tabular_csv(path)
    assert len(ds) == 17
    x, y = ds[16]
    assert x.shape == (48,) and y.shape == (5,)                      # first column dropped (MLP/dataset.py:28)
    np.testing.assert_allclose(x.numpy(), data[16, 1:49], rtol=1e-6)
    np.testing.assert_allclose(y.numpy(), data[16, 49:], rtol=1e-6)


def test_trainer_runs_on_csv(tmp_path):
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    rng = np.random.default_rng(2)
    x = rng.random((64, 48)).astype(np.float32)
    y = np.eye(5, dtype=np.float32)[rng.integers(0, 5, 64)]
    data = np.concatenate([np.zeros((64, 1), np.float32), x, y], axis=1)
    path = os.path.join(str(tmp_path), "mqtt.csv")
    np.savetxt(path, data, delimiter=",", header=",".join(f"c{i}" for i in range(data.shape[1])), comments="")
    cfg = Config(model="mlp", epochs=1, batch=8, data=path)
    t = Trainer(cfg).setup()
    assert len(t.dataset) == 64
    out = t.fit()
    assert "test_acc" in out[-1]
