import os

import pytest
import torch

from distributed_deep_learning_amd.config import Config
from distributed_deep_learning_amd.trainer import Trainer


def _cfg(**kw):
    base = dict(model="mlp", epochs=3, batch=32, samples=1024, lr=0.01, seed=42)
    base.update(kw)
    return Config(**base)


def test_fit_prints_reference_lines_and_learns():
    lines = []
    hist = Trainer(_cfg(epochs=4), printer=lines.append).setup().fit()
    assert len(lines) == 4 * 3 + 1
    assert lines[0].startswith('"train epoch 1 begins at ') and "with accuracy" in lines[1]
    ep = hist[:-1]
    assert ep[-1]["train_loss"] < ep[0]["train_loss"]
    assert "test_acc" in hist[-1]


@pytest.mark.parametrize("model,mode", [("cnn", "sequential"), ("cnn", "model"), ("mlp", "pipeline"), ("lstm", "sequential"),
                                        ("lstm", "model")])
def test_all_workloads_and_modes_run(model, mode):
    cfg = _cfg(model=model, mode=mode, epochs=1, samples=96, batch=16, stages=2, pipeline=4, max_steps=2)
    hist = Trainer(cfg, printer=lambda s: None).setup().fit()
    assert len(hist) == 2 and hist[0]["train_loss"] == hist[0]["train_loss"]


def test_pipeline_mode_trains_same_as_sequential_without_bn():
    a = Trainer(_cfg(epochs=1, mode="sequential"), printer=lambda s: None).setup()
    b = Trainer(_cfg(epochs=1, mode="pipeline", pipeline=8, stages=3), printer=lambda s: None).setup()
    ha, hb = a.fit(), b.fit()
    assert abs(ha[0]["train_loss"] - hb[0]["train_loss"]) < 1e-4


def test_checkpoint_resume_reproduces_the_loss_curve(tmp_path):
    path = os.path.join(tmp_path, "ck.pt")
    full = Trainer(_cfg(epochs=4), printer=lambda s: None).setup().fit()
    Trainer(_cfg(epochs=2, checkpoint=path), printer=lambda s: None).setup().fit()
    resumed = Trainer(_cfg(epochs=4, resume=path), printer=lambda s: None).setup().fit()
    assert [r["epoch"] for r in resumed[:-1]] == [3, 4]
    for want, got in zip(full[2:4], resumed[:2]):
        assert abs(want["train_loss"] - got["train_loss"]) < 1e-6
        assert abs(want["val_loss"] - got["val_loss"]) < 1e-6
    assert abs(full[-1]["test_loss"] - resumed[-1]["test_loss"]) < 1e-6
