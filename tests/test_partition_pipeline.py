import pytest
import torch

from distributed_deep_learning_amd import models
from distributed_deep_learning_amd.parallel import (PipelinedModel, StagedModel, partition_layers,
                                                    reference_partition_cnn, reference_partition_lstm,
                                                    reference_partition_mlp, schedule)

CPU = torch.device("cpu")


def _as_list(d, n):
    return [d[i] for i in range(n)]


def test_reference_tables_golden():
    # SURVEY 5.7.3 (verified by re-trace of the reference)
    assert _as_list(reference_partition_cnn(8, 2), 8) == [0, 0, 0, 0, 1, 1, 1, 1]
    assert _as_list(reference_partition_cnn(8, 1), 8) == [0] * 8
    mlp = {(3, 2): [0, 1, 1], (4, 2): [0, 0, 1, 1], (5, 2): [0, 0, 1, 1, 1], (6, 2): [0, 0, 0, 1, 1, 1],
           (5, 3): [0, 0, 1, 2, 2], (10, 4): [0, 0, 0, 1, 1, 2, 2, 3, 3, 3]}
    for (n, d), want in mlp.items():
        assert _as_list(reference_partition_mlp(n, d), n) == want
    lstm = {(4, 2): [0, 0, 1, 1], (5, 2): [0, 0, 0, 1, 1], (6, 2): [0, 0, 0, 1, 1, 1], (6, 4): [0, 0, 1, 2, 3, 3],
            (8, 4): [0, 0, 0, 1, 1, 2, 3, 3], (5, 5): [0, 1, 2, 3, 4]}
    for (n, d), want in lstm.items():
        assert _as_list(reference_partition_lstm(n, d), n) == want


@pytest.mark.parametrize("n,s", [(1, 1), (5, 2), (8, 3), (8, 8), (3, 5), (12, 4)])
def test_balanced_partition_properties(n, s):
    p = partition_layers(n, s)
    assert len(p) == n and p == sorted(p) and p[0] == 0
    assert set(p) == set(range(min(n, s)))            # contiguous, every stage used
    sizes = [p.count(k) for k in set(p)]
    assert max(sizes) - min(sizes) <= 1


def test_weighted_partition_minimises_heaviest_stage():
    assert partition_layers(6, 2, [1, 5, 5, 5, 5, 1]) == [0, 0, 0, 1, 1, 1]
    assert partition_layers(4, 2, [10, 1, 1, 1]) == [0, 1, 1, 1]


@pytest.mark.parametrize("stages", [1, 2, 3, 4])
@pytest.mark.parametrize("chunks", list(range(1, 17)))
def test_schedule_is_a_correct_pipeline(stages, chunks):
    ticks = schedule(chunks, stages)
    assert len(ticks) == chunks + stages - 1
    done = {c: -1 for c in range(chunks)}             # last stage finished per chunk
    finished = []
    for tick in ticks:
        busy = set()
        for s, c in tick:
            assert s not in busy                       # a stage does one chunk per tick
            busy.add(s)
            assert done[c] == s - 1                    # stages visited in order, exactly once
            done[c] = s
            if s == stages - 1:
                finished.append(c)
        # within a tick later stages come first
        assert [s for s, _ in tick] == sorted((s for s, _ in tick), reverse=True)
    assert finished == list(range(chunks))             # reference defect A4: [0,0,1,...] with a chunk lost
    assert all(v == stages - 1 for v in done.values())


@pytest.mark.parametrize("name,shape", [("mlp_ref", (9, 48)), ("lstm_ref", (8, 10, 32)), ("lenet5", (6, 1, 28, 28)),
                                        ("densenet_ref", (4, 3, 64, 64))])
def test_staged_and_pipelined_equal_sequential(name, shape):
    torch.manual_seed(0)
    m = models.build(name).eval()                      # eval: BN uses running stats, so micro-batching is exact
    x = torch.randn(*shape)
    want = m(x)
    for ndev in (2, 3):
        devs = [CPU] * ndev                            # the reference's repeated-CPU-device trick
        assert torch.allclose(StagedModel(m, devs)(x), want, atol=1e-6)
        for mb in (1, 2, 4):
            got = PipelinedModel(m, devs, micro_batch=mb)(x)
            assert got.shape == want.shape and torch.allclose(got, want, atol=1e-5)


def test_pipeline_backward_matches_sequential():
    torch.manual_seed(1)
    m = models.build("mlp_ref", hidden_layers=2)
    x = torch.randn(8, 48)
    m(x).sum().backward()
    ref = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    PipelinedModel(m, [CPU, CPU], micro_batch=3)(x).sum().backward()
    for g, p in zip(ref, m.parameters()):
        assert torch.allclose(g, p.grad, atol=1e-5)


@pytest.mark.parametrize("stages", [1, 2, 3, 4])
@pytest.mark.parametrize("chunks", [1, 2, 3, 5, 8, 13])
def test_1f1b_issue_order(stages, chunks):
    """Every chunk's backward directly follows the tick of its last forward; never more than ``stages`` chunks between
    their first forward and their backward (GPipe keeps all of them)."""
    from distributed_deep_learning_amd.parallel.pipeline import schedule_1f1b
    ops = schedule_1f1b(chunks, stages)
    assert [c for k, _, c in ops if k == "B"] == list(range(chunks))
    assert sorted((s, c) for k, s, c in ops if k == "F") == [(s, c) for s in range(stages) for c in range(chunks)]
    live, peak, seen_last = set(), 0, set()
    for k, s, c in ops:
        if k == "F":
            live.add(c)
            if s == stages - 1:
                seen_last.add(c)
        else:
            assert c in seen_last
            live.discard(c)
        peak = max(peak, len(live))
    assert peak <= min(stages, chunks)


@pytest.mark.parametrize("mb", [1, 3, 4, 8])
def test_1f1b_train_step_matches_whole_batch_gradients(mb):
    """``PipelinedModel.train_step``: sample-weighted mean of the micro-batch losses, gradients accumulated over the
    micro-batches == one backward of the whole batch (no BatchNorm in the MLP); hits are summed; at most ``stages``
    micro-batches alive."""
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    torch.manual_seed(1)
    m = models.build("mlp_ref", hidden_layers=2)
    x = torch.randn(10, 48)
    y = torch.randint(0, 3, (10,))
    crit = SoftmaxCrossEntropyLoss()
    ref_loss = crit(m(x), y)
    ref_hits = float(crit.last_correct)
    ref_loss.backward()
    ref = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    pm = PipelinedModel(m, [CPU, CPU, CPU], micro_batch=mb)
    loss, hits = pm.train_step(x, y, crit)
    assert abs(float(loss) - float(ref_loss)) < 1e-5 and float(hits) == ref_hits
    assert pm.max_live_chunks <= 3
    for g, p in zip(ref, m.parameters()):
        assert torch.allclose(g, p.grad, atol=1e-5)


def test_trainer_pipeline_schedules_agree():
    """-m pipeline: the 1F1B-style default and the reference's GPipe order train the same model to the same weights."""
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    w = {}
    for sched in ("1f1b", "gpipe"):
        cfg = Config(model="mlp", mode="pipeline", stages=2, epochs=1, batch=16, samples=160, optimizer="sgd", lr=0.05,
                     graph=False, pipeline=4, pipeline_schedule=sched)
        t = Trainer(cfg, printer=lambda *_: None).setup()
        t.fit()
        w[sched] = torch.cat([v.detach().flatten().float() for _, v in sorted(t.model.state_dict().items())])
    assert torch.allclose(w["1f1b"], w["gpipe"], atol=1e-5)


def test_bad_arguments_raise():
    m = models.build("mlp_ref")
    with pytest.raises(TypeError):
        StagedModel(m, torch.device("cpu"))
    with pytest.raises(ValueError):
        PipelinedModel(m, [CPU, CPU], micro_batch=0)
    with pytest.raises(ValueError):
        StagedModel(m, [CPU], partition=[0, 1, 1])
