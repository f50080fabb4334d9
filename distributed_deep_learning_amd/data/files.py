"""File-backed datasets for the reference's three workloads (used when ``--data PATH`` is given).

Same sample semantics as the reference readers, different mechanics: each dataset is decoded ONCE into whole
tensors (optionally device-resident, see :class:`.synthetic.TensorDataset`), so a training step slices a batch
instead of running a per-sample ``__getitem__`` + ``.to(device)``.

* :func:`pcb_defects` — ``/root/reference/src/pytorch/CNN/dataset.py:32-111``: ``<root>/images/<class>/*.jpg`` with
  Pascal-VOC ``<root>/Annotations/<class>/*.xml``; one sample per bounding box, doubled with a random shift of
  5..10 px; bilinear anti-aliased resize of the crop to 64x64 (outside the image = 0); float 0..255; one-hot target.
  XML is read with the standard library (the reference needs libxml2), images with PIL, resizing with
  ``torch.nn.functional.interpolate`` (the reference needs torchvision).  The reference reads a box as
  ``(xmin, xmax, ymin, ymax)`` and unpacks it as ``(xmin, ymin, xmax, ymax)`` (dataset.py:37 vs :94); the INTENT
  (a crop of the annotated box) is implemented here.
* :func:`window_csv` — ``/root/reference/src/pytorch/LSTM/dataset.py:24-45``: float CSV of ``machines`` blocks of
  ``rows_per_machine`` rows; sample = ``history`` consecutive rows of one machine, x = all but the last 5 columns,
  y = last 5 columns of the window's first row.
* :func:`tabular_csv` — ``/root/reference/src/pytorch/MLP/dataset.py:24-37``: float CSV, first column dropped,
  x = columns[:-5], y = columns[-5:].

Parity is unpinned against the reference's own data (``/data/...`` does not exist here); the tests build tiny
folders / CSVs of the same structure.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .synthetic import TensorDataset, WindowDataset


# ------------------------------------------------------------------------------------------------ PCB defects
def read_boxes(xml_path: str) -> List[Tuple[int, int, int, int]]:
    """``/annotation/object/bndbox`` entries as ``(xmin, ymin, xmax, ymax)``."""
    out = []
    for bb in ET.parse(xml_path).getroot().iterfind("./object/bndbox"):
        vals = {k: int(float(bb.findtext(k))) for k in ("xmin", "ymin", "xmax", "ymax")}
        out.append((vals["xmin"], vals["ymin"], vals["xmax"], vals["ymax"]))
    return out


def list_pcb_instances(root: str) -> Tuple[List[str], List[Tuple[str, Tuple[int, int, int, int], int]]]:
    """Classes (sorted sub-directories of ``Annotations``) and one ``(image path, box, class index)`` per box."""
    ann_root, img_root = os.path.join(root, "Annotations"), os.path.join(root, "images")
    classes = sorted(d for d in os.listdir(ann_root) if os.path.isdir(os.path.join(ann_root, d)))
    if not classes:
        raise FileNotFoundError(f"no class directories under {ann_root}")
    instances, seen = [], set()
    for ci, cls in enumerate(classes):
        cdir = os.path.join(img_root, cls)
        if not os.path.isdir(cdir):
            continue
        for dirpath, _, fnames in sorted(os.walk(cdir, followlinks=True)):
            for fname in sorted(fnames):
                if not fname.endswith(".jpg"):
                    continue
                xml_path = os.path.join(ann_root, cls, os.path.splitext(fname)[0] + ".xml")
                for box in read_boxes(xml_path):
                    instances.append((os.path.join(dirpath, fname), box, ci))
                    seen.add(cls)
    missing = set(classes) - seen
    if missing:
        raise FileNotFoundError("no .jpg with boxes for classes: " + ", ".join(sorted(missing)))
    return classes, instances


def crop_resize(img: torch.Tensor, top: int, left: int, height: int, width: int, size: int = 64) -> torch.Tensor:
    """``img`` uint8/float [C,H,W] -> float [C,size,size]; the part of the crop window outside the image is 0."""
    C, H, W = img.shape
    height, width = max(int(height), 1), max(int(width), 1)
    crop = torch.zeros((C, height, width), dtype=torch.float32)
    y0, x0 = max(top, 0), max(left, 0)
    y1, x1 = min(top + height, H), min(left + width, W)
    if y1 > y0 and x1 > x0:
        crop[:, y0 - top:y1 - top, x0 - left:x1 - left] = img[:, y0:y1, x0:x1].float()
    out = F.interpolate(crop[None], size=(size, size), mode="bilinear", antialias=True, align_corners=False)[0]
    return out.clamp_(0.0, 255.0)


class LazyPCBDataset(torch.utils.data.Dataset):
    """The PCB-defect dataset decoded PER SAMPLE on access (what the loader's ``-w`` worker threads parallelise):
    same samples, shifts and targets as :func:`pcb_defects` — which decodes everything once into RAM — for datasets
    that should not be held decoded in memory.  A small per-thread cache keeps the most recent image, since the two
    shifted crops of a box (and neighbouring boxes of one photo) are usually fetched together."""

    def __init__(self, root: str = "/data/PCB_DATASET", size: int = 64, seed: int = 0, one_hot: bool = True):
        import threading
        self.classes, self.instances = list_pcb_instances(root)
        g = torch.Generator().manual_seed(seed)
        self.shifts = torch.randint(5, 11, (2 * len(self.instances),), generator=g).tolist()
        self.size, self.one_hot = size, one_hot
        self._local = threading.local()

    def __len__(self) -> int:
        return 2 * len(self.instances)

    def _image(self, path: str) -> torch.Tensor:
        from PIL import Image
        if getattr(self._local, "path", None) != path:
            with Image.open(path) as im:
                self._local.img = torch.from_numpy(np.asarray(im.convert("RGB")).copy()).permute(2, 0, 1)
            self._local.path = path
        return self._local.img

    def __getitem__(self, i):
        i = int(i)
        path, (xmin, ymin, xmax, ymax), ci = self.instances[i >> 1]        # reference: index>>1 picks the box
        s = self.shifts[i]
        x = crop_resize(self._image(path), ymin + s, xmin + s, ymax - ymin, xmax - xmin, self.size)
        y = F.one_hot(torch.tensor(ci), len(self.classes)).float() if self.one_hot else torch.tensor(ci)
        return x, y


def pcb_defects(root: str = "/data/PCB_DATASET", size: int = 64, seed: int = 0, one_hot: bool = True) -> TensorDataset:
    from PIL import Image

    classes, instances = list_pcb_instances(root)
    g = torch.Generator().manual_seed(seed)
    shifts = torch.randint(5, 11, (2 * len(instances),), generator=g).tolist()
    xs = torch.empty((2 * len(instances), 3, size, size), dtype=torch.float32)
    ys = torch.empty((2 * len(instances),), dtype=torch.long)
    cache_path, cache_img = None, None
    for i, (path, (xmin, ymin, xmax, ymax), ci) in enumerate(instances):
        if path != cache_path:
            with Image.open(path) as im:
                cache_img = torch.from_numpy(np.asarray(im.convert("RGB")).copy()).permute(2, 0, 1)
            cache_path = path
        for k in (0, 1):                                    # every box twice, each with its own shift
            s = shifts[2 * i + k]
            xs[2 * i + k] = crop_resize(cache_img, ymin + s, xmin + s, ymax - ymin, xmax - xmin, size)
            ys[2 * i + k] = ci
    y = F.one_hot(ys, len(classes)).float() if one_hot else ys
    ds = TensorDataset(xs, y)
    ds.classes = classes
    return ds


# ------------------------------------------------------------------------------------------------ CSV workloads
def _read_float_csv(path: str, drop_first_column: bool = False) -> Tuple[List[str], torch.Tensor]:
    with open(path, "r", encoding="utf-8") as f:
        header = f.readline().strip().split(",")
    try:
        import pandas
        frame = pandas.read_csv(path, low_memory=False, dtype="float32")
        data = torch.from_numpy(frame.to_numpy(dtype=np.float32, copy=True))
    except ImportError:                                     # pandas is optional: numpy reads the same files
        data = torch.from_numpy(np.loadtxt(path, delimiter=",", skiprows=1, dtype=np.float32, ndmin=2))
    if drop_first_column:
        header, data = header[1:], data[:, 1:]
    return header, data.contiguous()


def window_csv(path: str = "/data/PredictiveMaintenance/dataset.csv", history: int = 10,
               rows_per_machine: int = 8759, targets: int = 5) -> WindowDataset:
    _, data = _read_float_csv(path)
    if len(data) % rows_per_machine != 0:
        raise ValueError(f"{path}: {len(data)} rows is not a multiple of rows_per_machine={rows_per_machine}")
    machines = len(data) // rows_per_machine
    ds = WindowDataset.__new__(WindowDataset)
    ds.machines, ds.rows, ds.history = machines, rows_per_machine, history
    block = data.view(machines, rows_per_machine, -1)
    ds.data = block[:, :, :-targets].contiguous()
    ds.targets = block[:, :, -targets:].contiguous()
    ds.per_machine = rows_per_machine - (history - 1)
    return ds


def tabular_csv(path: str = "/data/MQTT/dataset.csv", targets: int = 5, drop_first_column: bool = True) -> TensorDataset:
    _, data = _read_float_csv(path, drop_first_column=drop_first_column)
    return TensorDataset(data[:, :-targets].contiguous(), data[:, -targets:].contiguous())


def open_dataset(kind: str, path: str, seed: int = 0, lazy: bool = False):
    """``kind``: the workload family (``cnn`` | ``lstm`` | ``mlp``).  ``lazy`` (set when ``-w N > 0``): the image
    dataset decodes per sample on access, in the loader's worker threads, instead of once into RAM."""
    if kind in ("cnn", "densenet_ref"):
        return LazyPCBDataset(path, seed=seed) if lazy else pcb_defects(path, seed=seed)
    if kind in ("lstm", "lstm_ref"):
        return window_csv(path)
    if kind in ("mlp", "mlp_ref"):
        return tabular_csv(path)
    raise KeyError(f"no file-backed dataset for workload {kind!r}")
