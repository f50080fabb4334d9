"""cProfile of Trainer.fit for a reference workload: python bench/trainer_profile.py [model]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_deep_learning_amd.config import Config
from distributed_deep_learning_amd.trainer import Trainer

model = sys.argv[1] if len(sys.argv) > 1 else "lstm"
t = Trainer(Config(model=model, device="gpu", epochs=3, batch=32, samples=8192), printer=lambda *_: None).setup()
pr = cProfile.Profile()
pr.enable()
t.fit()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
