from .files import open_dataset, pcb_defects, tabular_csv, window_csv
from .loader import BatchLoader
from .sampler import ShardedSampler, split_indices
from .synthetic import (TensorDataset, WindowDataset, cifar_like, image_dataset, imagenet_like, mnist_like, pcb_like,
                        tabular_dataset)

__all__ = ["BatchLoader", "ShardedSampler", "split_indices", "TensorDataset", "WindowDataset", "cifar_like",
           "image_dataset", "imagenet_like", "mnist_like", "pcb_like", "tabular_dataset", "open_dataset",
           "pcb_defects", "tabular_csv", "window_csv"]
