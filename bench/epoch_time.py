"""Wall time of whole epochs (train + validation + test phases) of a reference workload through the Trainer:
python bench/epoch_time.py [model] [graph: auto|off]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_deep_learning_amd.config import Config
from distributed_deep_learning_amd.trainer import Trainer

model = sys.argv[1] if len(sys.argv) > 1 else "cnn"
graph = {"auto": None, "off": False, "on": True}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
cfg = Config(model=model, device="gpu", epochs=4, batch=32, samples=8192, graph=graph)
t = Trainer(cfg, printer=lambda *_: None).setup()
torch.cuda.synchronize()
t0 = time.perf_counter()
hist = t.fit()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tl = [h["train_loss"] for h in hist if "train_loss" in h]
print(f"{model} graph={sys.argv[2] if len(sys.argv) > 2 else 'auto'}: 4 epochs of 8192 samples (70/10/20 split, batch 32) in {dt:.2f} s "
      f"({4 * 8192 / dt:.0f} samples/s end to end); train loss {tl[0]:.4f} -> {tl[-1]:.4f}", flush=True)
