import re

import pytest

from distributed_deep_learning_amd import dist as ddist
from distributed_deep_learning_amd.config import Config, parse_args
from distributed_deep_learning_amd import metrics


def test_defaults_match_reference_cli():
    c = parse_args([])
    assert (c.epochs, c.batch, c.device, c.nworkers, c.mode, c.pipeline, c.run) == (10, 32, "cpu", 0, "sequential", 2, 1)
    assert (c.lr, c.momentum, c.lr_step, c.lr_gamma, c.seed) == (0.01, 0.9, 7, 0.1, 42)
    assert tuple(c.split) == (0.7, 0.1, 0.2)


def test_short_and_long_flags():
    c = parse_args(["-l", "3", "-s", "8", "-e", "2", "-b", "16", "-d", "cpu", "-w", "1", "-m", "data", "-p", "4", "-r", "2"])
    assert (c.nlayers, c.size, c.epochs, c.batch, c.nworkers, c.mode, c.pipeline, c.run) == (3, 8, 2, 16, 1, "data", 4, 2)
    c = parse_args(["--nlayers", "1", "--mode", "pipeline", "--sync", "ps", "--optimizer", "lars"])
    assert c.mode == "pipeline" and c.sync == "ps" and c.optimizer == "lars"


def test_invalid_values_rejected():
    with pytest.raises(SystemExit):
        parse_args(["-m", "bogus"])
    with pytest.raises(ValueError):
        Config(batch=0).validate()
    with pytest.raises(ValueError):
        Config(sync="nope").validate()


def test_rank_discovery_sources():
    assert ddist.discover({}) == ddist.RankInfo()
    r = ddist.discover({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3", "LOCAL_WORLD_SIZE": "8"})
    assert (r.rank, r.world, r.local_rank, r.local_world, r.launcher) == (3, 8, 3, 8, "torchrun")
    # the reference's MPI variables (CNN/main.py:63-67)
    r = ddist.discover({"OMPI_COMM_WORLD_RANK": "5", "OMPI_COMM_WORLD_SIZE": "16", "OMPI_COMM_WORLD_LOCAL_RANK": "1",
                        "OMPI_COMM_WORLD_LOCAL_SIZE": "4"})
    assert (r.rank, r.world, r.local_rank, r.local_world, r.launcher) == (5, 16, 1, 4, "mpi")
    r = ddist.discover({"SLURM_PROCID": "2", "SLURM_NTASKS": "4", "SLURM_LOCALID": "0"})
    assert (r.rank, r.world, r.launcher) == (2, 4, "slurm")
    assert ddist.backend_for(__import__("torch").device("cpu")) == "gloo"


def test_log_line_formats_are_the_references():
    assert metrics.fmt_train_begin(3, 12.5) == '"train epoch 3 begins at 12.500000"'
    assert metrics.fmt_train_end(3, 12.5, 91.23456, 0.1234567891) == \
        '"train epoch 3 ends at 12.500000 with accuracy 91.235 and loss 0.123456789"'
    assert metrics.fmt_val_end(1, 1.0, 50.0, 2.0) == \
        '"validation epoch 1 ends at 1.000000 with accuracy 50.000 and loss 2.000000000"'
    assert metrics.fmt_test_end(2.0, 10.0, 3.0) == '"test ends at 2.000000 with accuracy 10.000 and loss 3.000000000"'
    assert re.fullmatch(r'"train epoch \d+ begins at \d+\.\d{6}"', metrics.fmt_train_begin(1, metrics.now_ts()))


def test_phase_meter_true_mean():
    import torch
    m = metrics.PhaseMeter(torch.device("cpu"))
    m.update(torch.tensor(2.0), torch.tensor(3.0), 4)
    m.update(torch.tensor(1.0), torch.tensor(1.0), 2)
    acc, loss, n, _ = m.result()
    assert n == 6 and abs(loss - (2.0 * 4 + 1.0 * 2) / 6) < 1e-9 and abs(acc - 100 * 4 / 6) < 1e-9
