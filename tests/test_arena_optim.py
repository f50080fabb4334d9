import math

import numpy as np
import pytest
import torch
import torch.nn as tnn

from distributed_deep_learning_amd import models
from distributed_deep_learning_amd.optim import LARS, FusedAdam, FusedSGD, StepLR, WarmupPolyLR
from distributed_deep_learning_amd.parallel import CHUNK, ParamArena, plan_buckets


def _mlp(seed=0):
    torch.manual_seed(seed)
    return tnn.Sequential(tnn.Linear(20, 33), tnn.ReLU(), tnn.Linear(33, 7))


def test_plan_buckets_orders_and_caps():
    sizes = [10, 10, 10, 10, 10, 10]
    b = plan_buckets(sizes, bucket_elems=20, first_bucket_elems=10)
    assert b == [[0], [1, 2], [3, 4], [5]]            # last bucket (first ready) is the small one
    assert [i for g in b for i in g] == list(range(6))
    assert plan_buckets([5], 100) == [[0]]


def test_arena_aliases_parameters_and_preserves_layout():
    m = models.build("resnet18_cifar")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    a = ParamArena(m.named_parameters(), bucket_bytes=4 << 20, shard_multiple=2)
    a.check_layout()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    for p, s in zip(a.params, a.segments):
        assert s.offset % CHUNK == 0
        assert p.grad is not None and p.grad.data_ptr() == a.grad.data_ptr() + 4 * s.offset
    assert m.conv1.weight.is_contiguous(memory_format=torch.channels_last)
    for b in a.buckets:
        assert (b.hi - b.lo) % (CHUNK * 2) == 0
    assert a.buckets[0].lo == 0 and a.buckets[-1].hi == a.total
    # writing the flat buffer is visible through the parameter, and backward accumulates into the arena
    a.master.zero_()
    assert float(m.fc.weight.abs().sum()) == 0.0


def test_backward_accumulates_in_place_into_arena():
    m = _mlp()
    a = ParamArena(m.named_parameters())
    m(torch.randn(5, 20)).sum().backward()
    a.check_layout()
    for p, s in zip(a.params, a.segments):
        assert p.grad.data_ptr() == a.grad.data_ptr() + 4 * s.offset
    assert float(a.grad.abs().sum()) > 0


@pytest.mark.parametrize("wd,nesterov", [(0.0, False), (1e-2, False), (1e-2, True)])
def test_fused_sgd_matches_torch_sgd(wd, nesterov):
    m1, m2 = _mlp(1), _mlp(1)
    o1 = torch.optim.SGD(m1.parameters(), lr=0.05, momentum=0.9, weight_decay=wd, nesterov=nesterov)
    o2 = FusedSGD(m2.named_parameters(), lr=0.05, momentum=0.9, weight_decay=wd, nesterov=nesterov, no_decay=None)
    x = torch.randn(16, 20)
    for _ in range(5):
        o1.zero_grad()
        m1(x).pow(2).mean().backward()
        o1.step()
        m2(x).pow(2).mean().backward()
        o2.step()                                    # fused step also clears the gradient
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
    assert float(o2.arena.grad.abs().sum()) == 0.0


def test_fused_adam_matches_torch_adam():
    m1, m2 = _mlp(2), _mlp(2)
    o1 = torch.optim.Adam(m1.parameters(), lr=1e-2)
    o2 = FusedAdam(m2.named_parameters(), lr=1e-2)
    x = torch.randn(16, 20)
    for _ in range(6):
        o1.zero_grad()
        m1(x).pow(2).mean().backward()
        o1.step()
        m2(x).pow(2).mean().backward()
        o2.step()
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(a, b, atol=2e-6), (a - b).abs().max()


def test_lars_matches_numpy_reference():
    torch.manual_seed(3)
    m = _mlp(3)
    lr, mu, wd, eta, eps = 0.1, 0.9, 1e-3, 0.001, 1e-9
    opt = LARS(m.named_parameters(), lr=lr, momentum=mu, weight_decay=wd, eta=eta, eps=eps)
    ref_w = [p.detach().numpy().astype(np.float64).copy() for p in m.parameters()]
    ref_v = [np.zeros_like(w) for w in ref_w]
    x = torch.randn(8, 20)
    for _ in range(3):
        m(x).pow(2).mean().backward()
        grads = [p.grad.detach().numpy().astype(np.float64).copy() for p in m.parameters()]
        for i, (w, g) in enumerate(zip(ref_w, grads)):
            if w.ndim <= 1:                           # biases: no decay, no adaptation
                gp = g
            else:
                wn, gn = np.linalg.norm(w), np.linalg.norm(g)
                trust = eta * wn / (gn + wd * wn + eps) if wn > 0 and gn > 0 else 1.0
                gp = trust * (g + wd * w)
            ref_v[i] = mu * ref_v[i] + gp
            ref_w[i] = w - lr * ref_v[i]
        opt.step()
    for w, p in zip(ref_w, m.parameters()):
        assert np.allclose(w, p.detach().numpy(), rtol=1e-4, atol=1e-6)


def test_sharded_ranges_equal_full_step():
    m1, m2 = _mlp(4), _mlp(4)
    a1 = ParamArena(m1.named_parameters(), shard_multiple=2)
    a2 = ParamArena(m2.named_parameters(), shard_multiple=2)
    o1, o2 = FusedSGD(a1, lr=0.1, momentum=0.9), FusedSGD(a2, lr=0.1, momentum=0.9)
    x = torch.randn(4, 20)
    m1(x).sum().backward(); m2(x).sum().backward()
    o1.step()
    half = a2.total // 2
    o2.step(ranges=[(0, half), (half, a2.total)])
    assert torch.equal(a1.master, a2.master)


def test_optimizer_state_roundtrip_across_bucket_plans():
    m1, m2 = _mlp(5), _mlp(5)
    o1 = FusedSGD(ParamArena(m1.named_parameters(), bucket_bytes=1 << 12), lr=0.1, momentum=0.9)
    o2 = FusedSGD(ParamArena(m2.named_parameters(), bucket_bytes=1 << 30, shard_multiple=4), lr=0.5, momentum=0.9)
    x = torch.randn(4, 20)
    m1(x).sum().backward(); o1.step()
    o2.load_state_dict(o1.state_dict())
    assert o2.lr == 0.1 and o2.step_count == 1
    for s1, s2 in zip(o1.arena.segments, o2.arena.segments):
        assert torch.equal(o1.flat_state["momentum"][s1.offset:s1.offset + s1.numel],
                           o2.flat_state["momentum"][s2.offset:s2.offset + s2.numel])


def test_step_lr_schedule_is_the_references():
    m = _mlp()
    o = FusedSGD(m.named_parameters(), lr=0.01)
    s = StepLR(o, step_size=7, gamma=0.1)
    lrs = []
    for _ in range(15):
        lrs.append(o.lr)
        s.step()
    assert lrs[:7] == [0.01] * 7 and all(abs(v - 0.001) < 1e-12 for v in lrs[7:14]) and abs(lrs[14] - 1e-4) < 1e-12


def test_warmup_poly_schedule():
    o = FusedSGD(_mlp().named_parameters(), lr=1.0)
    s = WarmupPolyLR(o, warmup_steps=4, total_steps=12, power=2.0)
    lrs = []
    for _ in range(12):
        lrs.append(o.lr)
        s.step()
    assert lrs[:4] == [0.25, 0.5, 0.75, 1.0] and lrs[4] == 1.0 and lrs[-1] < 0.1 and lrs == sorted(lrs[:4]) + sorted(lrs[4:], reverse=True)
