"""GPU integration: fused units vs stock torch layers, fused optimizer vs its CPU path, end-to-end training."""
import copy

import pytest
import torch
import torch.nn as tnn
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def test_kernel_library_is_loaded_in_tree():
    from distributed_deep_learning_amd.ops import _lib
    _lib.require()
    maps = open("/proc/self/maps").read()
    assert "libddl_hip.so" in maps, "the HIP kernel library must be the code that runs on the GPU box"


def _ref_bottleneck(block, x):
    """fp32 reference of a Bottleneck using stock torch ops on the same (master) weights."""
    def cbr(conv, bn, inp, relu=True, res=None):
        y = TF.conv2d(inp, conv.weight.float(), None, conv.stride, conv.padding)
        y = TF.batch_norm(y, None, None, bn.weight, bn.bias, True, 0.1, bn.eps)
        if res is not None:
            y = y + res
        return TF.relu(y) if relu else y
    idt = x
    o = cbr(block.conv1, block.bn1, x)
    o = cbr(block.conv2, block.bn2, o)
    if block.downsample is not None:
        idt = cbr(block.downsample[0], block.downsample[1], x, relu=False)
    return cbr(block.conv3, block.bn3, o, res=idt)


@pytest.mark.parametrize("stride,inp,planes", [(1, 256, 64), (2, 256, 128)])
def test_bottleneck_matches_fp32_reference(stride, inp, planes):
    from distributed_deep_learning_amd.models.resnet import Bottleneck, _conv1x1
    from distributed_deep_learning_amd import nn as dnn
    torch.manual_seed(0)
    down = None
    if stride != 1 or inp != planes * 4:
        down = tnn.Sequential(_conv1x1(inp, planes * 4, stride), dnn.BatchNorm2d(planes * 4))
    blk = Bottleneck(inp, planes, stride, down).to(_dev())
    for m in blk.modules():
        if isinstance(m, tnn.BatchNorm2d):
            tnn.init.uniform_(m.weight, 0.5, 1.5)
            tnn.init.normal_(m.bias, 0, 0.1)
    ref = copy.deepcopy(blk)
    x = torch.randn(8, inp, 14, 14, device=_dev())
    xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = xb.detach().float().requires_grad_(True)
    y = blk(xb)
    yr = _ref_bottleneck(ref, xr)
    assert (y.float() - yr).abs().max() < 0.15 and (y.float() - yr).abs().mean() < 0.01
    g = torch.randn_like(yr)
    y.backward(g.to(torch.bfloat16))
    yr.backward(g)
    rel = lambda a, b: ((a.float() - b).norm() / (b.norm() + 1e-12)).item()  # noqa: E731
    assert rel(xb.grad, xr.grad) < 0.12      # bf16 intermediates + ReLU-boundary flips over 3 layers
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad) < 0.12, (n, rel(p.grad, q.grad))
    # running statistics follow PyTorch semantics
    assert torch.allclose(blk.bn1.running_var, 0.9 + 0.1 * TF.conv2d(xr, ref.conv1.weight.float()).var((0, 2, 3)), rtol=2e-2, atol=1e-3)
    sd = blk.state_dict()
    assert int(sd["bn1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("kind", ["sgd", "lars", "adam"])
def test_fused_optimizer_kernel_matches_cpu_path(kind):
    from distributed_deep_learning_amd.optim import LARS, FusedAdam, FusedSGD
    from distributed_deep_learning_amd.parallel import ParamArena

    def make(device):
        torch.manual_seed(0)
        from distributed_deep_learning_amd import nn as dnn
        m = tnn.Sequential(dnn.Conv2d(8, 16, 3), tnn.BatchNorm2d(16), tnn.Flatten(), tnn.Linear(16 * 36, 10)).to(device)
        a = ParamArena(m.named_parameters())
        if kind == "sgd":
            o = FusedSGD(a, lr=0.1, momentum=0.9, weight_decay=1e-2, nesterov=True)
        elif kind == "lars":
            o = LARS(a, lr=0.5, momentum=0.9, weight_decay=1e-3, eta=0.01)
        else:
            o = FusedAdam(a, lr=1e-2, weight_decay=1e-3)
        return m, a, o

    mc, ac, oc = make(torch.device("cpu"))
    mg, ag, og = make(_dev())
    g = torch.Generator().manual_seed(1)
    for _ in range(4):
        grad = torch.randn(ac.total, generator=g)
        # only real elements carry gradient (padding stays zero, as after a real backward)
        maskv = torch.zeros(ac.total)
        for s in ac.segments:
            maskv[s.offset:s.offset + s.numel] = 1
        grad *= maskv
        ac.grad.copy_(grad)
        ag.grad.copy_(grad.to(_dev()))
        oc.step()
        og.step()
    assert torch.allclose(ag.master.cpu(), ac.master, rtol=2e-5, atol=2e-6)
    assert float(ag.grad.abs().sum()) == 0.0                                  # fused zero-grad
    assert torch.equal(ag.shadow.float().cpu(), ag.master.to(torch.bfloat16).float().cpu())   # bf16 shadow rewritten
    conv = ag.params[0]
    assert torch.equal(conv._ddl_shadow_t.permute(3, 0, 1, 2).float(), conv.detach().to(torch.bfloat16).float())


def test_resnet18_cifar_loss_decreases_and_matches_eval():
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import ParamArena
    torch.manual_seed(0)
    model = models.resnet18_cifar().to(_dev())
    arena = ParamArena(model.named_parameters())
    opt = FusedSGD(arena, lr=0.05, momentum=0.9)
    crit = SoftmaxCrossEntropyLoss()
    x = torch.randn(64, 3, 32, 32, device=_dev())
    y = torch.randint(0, 10, (64,), device=_dev())
    losses = []
    for _ in range(50):
        loss = crit(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.5 * losses[0], losses[::10]
    model.eval()
    with torch.no_grad():
        out = model(x)
    assert out.shape == (64, 10) and torch.isfinite(out.float()).all()
    arena.check_layout()


def test_rccl_world1_ddp_smoke_and_ps_mode(tmp_path):
    """RCCL communicator init + in-place collectives on arena buckets with world_size = 1 (identity)."""
    import os
    import torch.distributed as td
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    from distributed_deep_learning_amd.utils import GradLedger
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    td.init_process_group("nccl", rank=0, world_size=1, device_id=_dev())
    try:
        t = torch.arange(8, dtype=torch.float32, device=_dev())
        td.all_reduce(t)
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        torch.manual_seed(0)
        model = models.resnet18_cifar().to(_dev())
        ddp = DistributedDataParallel(model, bucket_mb=8)
        assert not ddp.dist                      # world 1: reducer is a pass-through
        opt = FusedSGD(ddp.arena, lr=0.05, momentum=0.9)
        crit = SoftmaxCrossEntropyLoss()
        x = torch.randn(32, 3, 32, 32, device=_dev())
        y = torch.randint(0, 10, (32,), device=_dev())
        l0 = None
        for _ in range(5):
            loss = crit(ddp(x), y)
            loss.backward()
            ddp.step(opt)
            l0 = l0 or float(loss.detach())
        assert float(loss.detach()) < l0

        # The same loop with the collectives really issued over RCCL (identities at world 1): bucketed async
        # all-reduce launched from the gradient-ready notifications behind the side-stream wgrads, and the PS mode's
        # reduce-scatter -> sharded step -> all-gather.  Both must train exactly like the pass-through reducer.
        def run(mode, force, wire=None, comm="torch"):
            torch.manual_seed(0)
            m = models.resnet18_cifar().to(_dev())
            d = DistributedDataParallel(m, bucket_mb=2, mode=mode, force_collectives=force, grad_dtype=wire, comm=comm)
            assert (d.ncomm is not None) == (comm == "native" and force)
            assert d.dist == force and len(d.arena.buckets) > 3
            o = FusedSGD(d.arena, lr=0.05, momentum=0.9)
            d.attach(o)
            # The native kernels write gradients straight into the arena and report readiness themselves; at world 1
            # a bucket reduced too early would go unnoticed (identity collective), so the ledger asserts the
            # contract: every segment fires once, no bucket is launched with gradients outstanding.
            ledger = GradLedger(d, snapshot=True) if force else None
            ls = []
            for _ in range(6):
                loss = crit(d(x), y)
                loss.backward()
                if force:
                    # all buckets were launched from inside backward (overlap), last bucket first
                    assert d._next_bucket == -1, d._next_bucket                   # noqa: SLF001
                    assert ledger.launch_order() == list(range(len(d.arena.buckets) - 1, -1, -1))
                    assert len(ledger.fired) == len(d.arena.segments) and set(d._pending) == {0}   # noqa: SLF001
                d.step(o)                            # -> on_synced -> ledger.verify_complete (bucket snapshots)
                ls.append(float(loss.detach()))
            if force and wire is None:
                assert ledger.verified == 6 * len(d.arena.buckets)
            return ls, d.arena.master.clone()

        base, wb = run("allreduce", False)
        for mode, wire, comm in (("allreduce", None, "torch"), ("ps", None, "torch"),
                                 ("allreduce", torch.bfloat16, "torch"), ("ps", torch.bfloat16, "torch"),
                                 ("allreduce", None, "native"), ("ps", None, "native"),
                                 ("ps", torch.bfloat16, "native")):
            ls, w = run(mode, True, wire, comm)
            tol = 0.05 if wire is None else 0.25     # bf16-rounded gradients: same descent, looser trajectory
            assert all(abs(a - b) < tol * max(1.0, abs(b)) for a, b in zip(ls, base)), (mode, wire, comm, ls, base)
            assert ls[-1] < 0.2 * ls[0]
            assert (w - wb).abs().max().item() < tol * wb.abs().max().item()
    finally:
        td.destroy_process_group()


def test_trainer_gpu_cnn_reference_workload():
    """The reference's CNN workload (DenseNet variant, 64x64 crops, one-hot targets) end to end on the GPU."""
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    lines = []
    cfg = Config(model="cnn", device="gpu", epochs=2, batch=32, samples=512, lr=0.01)
    hist = Trainer(cfg, printer=lines.append).setup().fit()
    assert len(lines) == 7 and hist[1]["train_loss"] < hist[0]["train_loss"] * 1.05


def test_hip_graph_step_matches_eager_and_counts_batches():
    """utils.GraphedTrainStep: a captured step replays to the same training trajectory as eager steps."""
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.nn.modules import BatchNorm2d
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import ParamArena
    from distributed_deep_learning_amd.utils import GraphedTrainStep
    x = torch.randn(32, 3, 32, 32, device=_dev())
    y = torch.randint(0, 10, (32,), device=_dev())

    def make():
        torch.manual_seed(0)
        model = models.resnet18_cifar().to(_dev())
        arena = ParamArena(model.named_parameters())
        return model, arena, FusedSGD(arena, lr=0.02, momentum=0.9), SoftmaxCrossEntropyLoss()

    warm, steps = 2, 12
    m1, a1, o1, c1 = make()
    eager = []
    for _ in range(warm + steps):                        # the warm-up steps train too; the capture itself does not
        loss = c1(m1(x), y)
        loss.backward()
        o1.step()
        eager.append(float(loss.detach()))

    m2, a2, o2, c2 = make()
    step = GraphedTrainStep(m2, c2, o2, x, y, warmup=warm)
    graphed = [float(step(x, y)) for _ in range(steps)]
    assert o2.step_count == o1.step_count
    ref = eager[warm:]
    assert graphed[-1] < graphed[0]
    for a, b in zip(graphed, ref):                       # bf16 + atomics order: trajectories agree loosely
        assert abs(a - b) < 0.15 * max(1.0, abs(b)), (graphed, ref)
    bn = next(m for m in m2.modules() if isinstance(m, BatchNorm2d))
    bn.flush_counters()
    assert int(bn.num_batches_tracked) == warm + steps
    # a new learning rate reaches the replayed kernels through the device-resident hyper array
    before = a2.master.clone()
    o2.param_groups[0]["lr"] = 0.0
    o2.flat_state["momentum"].zero_()
    step(x, y)
    assert torch.equal(a2.master, before)


def test_trainer_with_hip_graph_flag():
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    cfg = Config(model="lenet5", device="gpu", epochs=2, batch=64, samples=640, graph=True, lr=0.05)
    t = Trainer(cfg).setup()
    hist = t.fit()
    assert getattr(t, "_graphed", None) is not None and t._graphed.replays > 0
    assert hist[1]["train_loss"] < hist[0]["train_loss"]



@pytest.mark.parametrize("model,mode", [("mlp", "sequential"), ("lstm", "sequential"), ("cnn", "pipeline"), ("mlp", "model")])
def test_trainer_gpu_reference_workloads(model, mode):
    """The reference's three workloads and its model / pipeline modes through the Trainer on the GPU (stages may share
    the one device, like the reference's all-CPU 4-"device" LSTM self-test, LSTM/model.py:180-192)."""
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    cfg = Config(model=model, device="gpu", mode=mode, epochs=2, batch=32, samples=256, stages=2, pipeline=8)
    t = Trainer(cfg).setup()
    hist = t.fit()
    assert len(hist) == 3 and all(torch.isfinite(torch.tensor(h.get("train_loss", 0.0))) for h in hist[:2])
    assert "test_acc" in hist[-1]


def test_lenet_runs_on_the_native_convs_with_padded_channels():
    """LeNet-5's 1 -> 6 -> 16 channel convolutions run on the MFMA kernels (input and conv1 output padded to 8
    channels, ``Conv2d(pad_out=True)``) and match the fp32 CPU model: logits and every parameter gradient."""
    import warnings
    from distributed_deep_learning_amd import models
    torch.manual_seed(11)
    ref = models.lenet5()
    gpu = models.lenet5().to(_dev())
    gpu.load_state_dict(ref.state_dict())
    x = torch.randn(16, 1, 28, 28)
    y = torch.randint(0, 10, (16,))
    lr = torch.nn.functional.cross_entropy(ref(x), y)
    lr.backward()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # the ATen fallback warns: it must not be taken
        out = gpu(x.to(_dev()))
        assert gpu.conv1(torch.randn(2, 1, 28, 28, device=_dev())).shape[1] == 8
    lg = torch.nn.functional.cross_entropy(out.float(), y.to(_dev()))
    lg.backward()
    assert abs(float(lg.detach()) - float(lr.detach())) < 3e-2
    for (n, p), (_, q) in zip(gpu.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, n
        g = p.grad.float().cpu()
        err = float((g - q.grad).norm() / (q.grad.norm() + 1e-12))
        cos = float(torch.nn.functional.cosine_similarity(g.flatten(), q.grad.flatten(), dim=0))
        # bf16 activations / gradients against an fp32 model: the error grows towards the input (0.5 % at fc3,
        # ~5 % at fc1/fc2, ~10 % at conv1, measured); a channel-mapping bug would destroy the direction instead
        assert err < 0.2 and cos > 0.985, (n, err, cos)


def test_deterministic_mode_makes_resnet50_gradients_reproducible():
    """Two identical forward + backward passes of ResNet-50.  With the float atomics of the default filter-gradient
    path the two runs differ by summation order, which 53 stacked BatchNorms amplify towards the input (a few percent
    at the stem); in deterministic mode (ordered reduce of the split-M slabs, csrc/conv_wgrad.hip) the only unordered
    sums left are the fp64 BatchNorm statistics, whose order noise (1e-16) almost never survives the rounding to
    fp32: the gradients then repeat to the bit or very nearly."""
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.ops import state, streams
    x = torch.randn(16, 3, 96, 96, device=_dev())
    y = torch.randint(0, 1000, (16,), device=_dev())

    def run():
        torch.manual_seed(3)
        m = models.build("resnet50").to(_dev())
        torch.nn.functional.cross_entropy(m(x).float(), y).backward()
        streams.join()
        torch.cuda.synchronize()
        return {n: p.grad.float().clone() for n, p in m.named_parameters()}

    def worst(a, b):
        return max(float((a[n] - b[n]).norm() / (a[n].norm() + 1e-12)) for n in a)

    state.set_deterministic(True)
    try:
        det = worst(run(), run())
    finally:
        state.set_deterministic(False)
    fast = worst(run(), run())
    print("worst relative run-to-run difference: deterministic", det, "default (atomics)", fast)
    assert det < 1e-6, det                  # measured: exactly 0.0
    assert fast < 0.1, fast                 # measured: 3e-7 at this size; grows with the number of split-M partials


@pytest.mark.parametrize("model", ["cnn", "lstm", "mlp"])
def test_trainer_graph_auto_engages_for_reference_workloads(model):
    """Default ``graph=None`` (auto): the reference's three launch-bound workloads replay from a hipGraph on a
    single-process GPU run and still learn; ``graph=False`` keeps the eager step."""
    from distributed_deep_learning_amd.config import Config
    from distributed_deep_learning_amd.trainer import Trainer
    cfg = Config(model=model, device="gpu", epochs=3, batch=32, samples=640)
    t = Trainer(cfg, printer=lambda *_: None).setup()
    hist = t.fit()
    assert getattr(t, "_graphed", None) is not None and t._graphed.replays > 0
    tl = [h["train_loss"] for h in hist if "train_loss" in h]
    assert len(tl) == 3 and tl[-1] < tl[0] * 1.02
    t2 = Trainer(Config(model=model, device="gpu", epochs=1, batch=32, samples=320, graph=False), printer=lambda *_: None).setup()
    t2.fit()
    assert getattr(t2, "_graphed", None) is None


def test_conv_relu_chain_gated_dgrad_matches_fp32_reference():
    """ConvReluChain (VGG feature run as one node; ReLU gate + bias gradient taken in the next unit's dgrad epilogue,
    also through a max-pool) against plain fp32 torch, and against the same chain with the fusion switched off."""
    from distributed_deep_learning_amd import nn as dnn
    from distributed_deep_learning_amd.nn import functional as Fn
    from distributed_deep_learning_amd.nn.modules import conv_relu_chain
    torch.manual_seed(2)
    chans = [(3, 16, True), (16, 32, False), (32, 32, True), (32, 64, False)]
    ref = [torch.nn.Conv2d(a, b, 3, 1, 1) for a, b, _ in chans]
    x = torch.randn(4, 3, 16, 16)
    h = x
    for m, (_, _, pool) in zip(ref, chans):
        h = torch.relu(m(h))
        if pool:
            h = torch.nn.functional.max_pool2d(h, 2, 2)
    w = torch.randn_like(h)
    (h * w).sum().backward()
    results = {}
    for flag in (True, False):
        convs = [dnn.Conv2d(a, b, 3, 1, 1).to(_dev()) for a, b, _ in chans]
        for c, r in zip(convs, ref):
            c.load_state_dict(r.state_dict())
        units = [(c, dnn.MaxPool2d(2, 2, 0) if pool else None) for c, (_, _, pool) in zip(convs, chans)]
        Fn._RELU_GATE = flag
        try:
            out = conv_relu_chain(units, x.to(_dev()))
            assert out.grad_fn is not None and "ConvReluChain" in type(out.grad_fn).__name__
            (out.float() * w.to(_dev())).sum().backward()
        finally:
            Fn._RELU_GATE = True
        results[flag] = [(c.weight.grad.float().cpu(), c.bias.grad.float().cpu()) for c in convs]
        assert float((out.float().cpu() - h).norm() / h.norm()) < 2e-2
    for (gw, gb), (uw, ub), r in zip(results[True], results[False], ref):
        for got, unf, want in ((gw, uw, r.weight.grad), (gb, ub, r.bias.grad)):
            # vs fp32: bf16 activations flip ~0.3 % of the ReLU gates (values within rounding of 0), each flip is a full
            # +-dy*x term, so the norm error is ~sqrt(fraction): 5 % at the last unit, 17 % at the first (measured,
            # identical for the fused and the unfused backward); the direction must agree
            cos = float(torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0))
            assert float((got - want).norm() / (want.norm() + 1e-9)) < 0.3 and cos > 0.97
            # the fusion itself only moves the gate and the sums into another kernel: same values
            assert float((got - unf).norm() / (unf.norm() + 1e-9)) < 1e-3


def test_native_rccl_communicator_world1_collectives():
    """csrc/comm.cpp: RCCL resolved with dlopen from torch's copy, communicator init from a unique id, the four
    collectives as identities at world 1 (fp32 and bf16), ordering against the compute stream through events."""
    from distributed_deep_learning_amd.parallel.native_comm import NativeCommunicator
    c = NativeCommunicator(0, 1, _dev())
    assert c.version > 20000
    for dt in (torch.float32, torch.bfloat16):
        a = torch.randn(1 << 20, device=_dev()).to(dt)
        ref = a.clone()
        a.mul_(2.0)                                   # producer on the current stream: the collective must see it
        c.all_reduce(a, average=True).wait()
        assert torch.equal(a, ref * 2.0)
        out = torch.empty_like(a)
        c.reduce_scatter(out, a, average=False).wait()
        assert torch.equal(out, a)
        g = torch.zeros_like(a)
        c.all_gather(g, a).wait()
        assert torch.equal(g, a)
        c.broadcast(a, 0).wait()
        assert torch.equal(a, ref * 2.0)
    torch.cuda.synchronize()
    c.close()


@pytest.mark.parametrize("mode", ["allreduce", "ps"])
def test_hip_graph_captures_native_collectives(mode):
    """A data-parallel step (bucketed reducer, native RCCL communicator, world 1 with the collectives forced) replays
    from ONE hipGraph — kernels and collectives — along the eager reducer's trajectory."""
    import os
    import torch.distributed as td
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    from distributed_deep_learning_amd.utils import GraphedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", HSA_ENABLE_IPC_MODE_LEGACY="0")
    td.init_process_group("nccl", rank=0, world_size=1, device_id=_dev())
    try:
        x = torch.randn(32, 3, 32, 32, device=_dev())
        y = torch.randint(0, 10, (32,), device=_dev())

        def make():
            torch.manual_seed(0)
            m = models.resnet18_cifar().to(_dev())
            d = DistributedDataParallel(m, bucket_mb=2, mode=mode, force_collectives=True, comm="native")
            o = FusedSGD(d.arena, lr=0.02, momentum=0.9)
            d.attach(o)
            return m, d, o, SoftmaxCrossEntropyLoss()

        m1, d1, o1, c1 = make()
        eager = []
        for _ in range(10):
            loss = c1(d1(x), y)
            loss.backward()
            d1.step(o1)
            eager.append(float(loss.detach()))
        m2, d2, o2, c2 = make()
        g = GraphedTrainStep(m2, c2, o2, x, y, warmup=2, ddp=d2)
        graphed = [float(g(x, y)) for _ in range(8)]
        assert graphed[-1] < graphed[0]
        for a, b in zip(graphed, eager[2:]):
            assert abs(a - b) < 0.15 * max(1.0, abs(b)), (graphed, eager)
    finally:
        td.destroy_process_group()


def test_failed_native_communicator_falls_back_to_an_nccl_group_not_to_host_staging(monkeypatch):
    """Control plane gloo, data plane native RCCL: when the native communicator cannot be opened (here: its constructor
    raises) every rank learns the verdict through the control plane and the reducer opens an RCCL group inside
    ``torch.distributed`` for the gradients - not the host-staged gloo rehearsal transport."""
    import os
    import torch.distributed as td
    from distributed_deep_learning_amd import dist as ddist
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel, native_comm

    def broken(self, *a, **k):
        raise RuntimeError("simulated: librccl refused the communicator")

    monkeypatch.setattr(native_comm.NativeCommunicator, "__init__", broken)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29647", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ddist.init(ddist.RankInfo(rank=0, world=1), _dev(), force=True, comm="native")
    d = None
    try:
        assert td.get_backend() == "gloo"
        torch.manual_seed(0)
        m = models.resnet18_cifar().to(_dev())
        d = DistributedDataParallel(m, bucket_mb=2, force_collectives=True, comm="native")
        assert d.ncomm is None and d.backend == "nccl" and not d.staged and "nccl group" in d.comm_note
        o = FusedSGD(d.arena, lr=0.02, momentum=0.9)
        d.attach(o)
        crit = SoftmaxCrossEntropyLoss()
        x = torch.randn(32, 3, 32, 32, device=_dev())
        y = torch.randint(0, 10, (32,), device=_dev())
        losses = []
        for _ in range(6):
            loss = crit(d(x), y)
            loss.backward()
            d.step(o)
            losses.append(float(loss.detach()))
        assert losses[-1] < losses[0], losses
        cs = d.param_checksum()
        assert torch.equal(cs, cs[0:1].expand_as(cs))
    finally:
        if d is not None:
            d.close()
        td.destroy_process_group()


def test_ledger_snapshot_on_vgg_chain_and_lenet():
    """The conv-ReLU chain writes bias gradients from the NEXT unit's dgrad epilogue (slab fold into the arena) and
    filter gradients on the side stream: the ledger's bucket snapshots prove that every one of them has landed before
    its bucket's collective is enqueued (world 1, RCCL identity collectives)."""
    import os
    import torch.distributed as td
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import DistributedDataParallel
    from distributed_deep_learning_amd.utils import GradLedger
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29651", HSA_ENABLE_IPC_MODE_LEGACY="0")
    td.init_process_group("nccl", rank=0, world_size=1, device_id=_dev())
    try:
        cases = [(models.vgg11(num_classes=10, input_size=32, hidden=256, dropout=0.0), (16, 3, 32, 32), 0.25),
                 (models.lenet5(), (16, 1, 28, 28), 0.02)]
        for model, shape, bucket_mb in cases:
            torch.manual_seed(1)
            m = model.to(_dev())
            d = DistributedDataParallel(m, bucket_mb=bucket_mb, force_collectives=True)
            assert len(d.arena.buckets) > 2
            ledger = GradLedger(d, snapshot=True)
            o = FusedSGD(d.arena, lr=0.01, momentum=0.9)
            crit = SoftmaxCrossEntropyLoss()
            x = torch.randn(*shape, device=_dev())
            y = torch.randint(0, 10, (shape[0],), device=_dev())
            first = None
            for _ in range(4):
                loss = crit(d(x), y)
                loss.backward()
                assert d._next_bucket == -1                         # noqa: SLF001 - all launched inside backward
                d.step(o)
                first = first if first is not None else float(loss.detach())
            assert ledger.verified == 4 * len(d.arena.buckets) and not ledger.violations
            assert float(loss.detach()) < first
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("name,shape,classes", [("cnn", (32, 3, 64, 64), 6), ("lenet5", (32, 1, 28, 28), 10)])
def test_graphed_eval_step_tracks_training(name, shape, classes):
    """GraphedEvalStep: the captured eval forward gives the eager eval loss — also AFTER more training steps changed
    the weights and the BN running statistics (nothing stale is baked into the graph)."""
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import ParamArena
    from distributed_deep_learning_amd.utils import GraphedEvalStep
    torch.manual_seed(0)
    m = models.build(name).to(_dev())
    arena = ParamArena(m.named_parameters())
    opt = FusedSGD(arena, lr=0.05, momentum=0.9)
    crit = SoftmaxCrossEntropyLoss()
    x = torch.randn(*shape, device=_dev())
    y = torch.randint(0, classes, (shape[0],), device=_dev())

    def train(k):
        m.train()
        for _ in range(k):
            crit(m(x), y).backward()
            opt.step()
        m.eval()

    train(2)
    ev = GraphedEvalStep(m, crit, x, y)
    seen = []
    for _ in range(3):
        with torch.no_grad():
            want = float(crit(m(x), y))
        got, correct = ev(x, y)
        assert abs(float(got) - want) < 2e-3 * max(1.0, abs(want)), (float(got), want)
        assert correct is not None
        seen.append(want)
        train(3)
    assert seen[-1] < seen[0]                      # the evaluated model did change between the replays


def test_train_graph_survives_empty_cache_and_a_second_capture():
    """Regression: a captured hipMemsetAsync (memset node on a small graph-pool buffer; LeNet-5's padded-bias gradient
    was the only captured user) made replays produce garbage after torch.cuda.empty_cache() — which every later
    torch.cuda.graph capture calls.  All zero fills of the kernel library are fill kernels now."""
    from distributed_deep_learning_amd import models
    from distributed_deep_learning_amd.nn import SoftmaxCrossEntropyLoss
    from distributed_deep_learning_amd.optim import FusedSGD
    from distributed_deep_learning_amd.parallel import ParamArena
    from distributed_deep_learning_amd.utils import GraphedEvalStep, GraphedTrainStep
    torch.manual_seed(0)
    m = models.lenet5().to(_dev())
    arena = ParamArena(m.named_parameters())
    opt = FusedSGD(arena, lr=0.05, momentum=0.9)
    crit = SoftmaxCrossEntropyLoss()
    x = torch.randn(64, 1, 28, 28, device=_dev())
    y = torch.randint(0, 10, (64,), device=_dev())
    g = GraphedTrainStep(m, crit, opt, x, y, warmup=1)
    a = [float(g(x, y)) for _ in range(3)]
    torch.cuda.empty_cache()
    b = [float(g(x, y)) for _ in range(3)]
    m.eval()
    ev = GraphedEvalStep(m, SoftmaxCrossEntropyLoss(), x, y)      # a second capture (gc + empty_cache inside)
    e = float(ev(x, y)[0])
    m.train()
    c = [float(g(x, y)) for _ in range(3)]
    seq = a + b + c
    assert all(v == v and v < 3.0 for v in seq + [e]), (seq, e)
    assert seq[-1] < seq[0]


# --------------------------------------------------------------------------------------------- whole-network parity
def _whole_network_parity(name, batch, image, classes=1000, calibrate=True, branch_gain=None):
    """One state_dict in the GPU model (bf16 HIP kernels) and in the SAME model on the CPU (fp32 stock torch ops,
    ``nn/modules.py``): one training-mode forward + backward on the same batch.  ``calibrate`` runs the CPU model a
    second time under stock bf16 autocast: how far bf16 arithmetic ALONE moves each gradient of this network (the
    gradients of batch-norm parameters that sit in front of another batch-norm are sums that cancel almost completely,
    so their relative error is large in ANY reduced precision) - the yardstick our kernels are held to."""
    from distributed_deep_learning_amd import models
    torch.manual_seed(21)
    ref = models.build(name)
    if branch_gain is not None:                    # scale every residual branch: the last batch-norm of each block
        with torch.no_grad():
            for n, p in ref.named_parameters():
                if n.endswith("bn3.weight"):
                    p.fill_(branch_gain)
    gpu = models.build(name).to(_dev())
    gpu.load_state_dict(ref.state_dict())
    ref.train(); gpu.train()
    if name.startswith("vgg"):
        for m in (ref, gpu):                       # dropout draws different masks on the two paths: compare without it
            for mod in m.modules():
                if hasattr(mod, "dropout"):
                    mod.dropout = 0.0
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
    x = torch.randn(batch, 3, image, image)
    y = torch.randint(0, classes, (batch,))

    def cpu_step(autocast):
        for p in ref.parameters():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = ref(x)
        loss = torch.nn.functional.cross_entropy(out.float(), y)
        loss.backward()
        return out.detach().float(), float(loss.detach()), [p.grad.clone() for p in ref.parameters()]

    stock = cpu_step(True) if calibrate else None
    out_r, loss_r, grads_r = cpu_step(False)
    out_g = gpu(x.to(_dev()))
    loss_g = torch.nn.functional.cross_entropy(out_g.float(), y.to(_dev()))
    loss_g.backward()
    torch.cuda.synchronize()
    cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))  # noqa: E731
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-12))          # noqa: E731
    rows, grads_g = [], []
    for i, (n, p) in enumerate(gpu.named_parameters()):
        assert p.grad is not None, n
        g = p.grad.float().cpu()
        grads_g.append(g)
        es, cs = (rel(stock[2][i], grads_r[i]), cos(stock[2][i], grads_r[i])) if calibrate else (0.0, 1.0)
        rows.append((n, g.numel(), rel(g, grads_r[i]), cos(g, grads_r[i]), es, cs))
    flat = lambda gs: torch.cat([g.flatten() for g in gs])                                            # noqa: E731
    res = {"logit_err": rel(out_g.detach().float().cpu(), out_r), "loss_err": abs(float(loss_g.detach()) - loss_r),
           "cos_all": cos(flat(grads_g), flat(grads_r)), "rows": rows}
    if calibrate:
        res.update(logit_err_stock=rel(stock[0], out_r), loss_err_stock=abs(stock[1] - loss_r),
                   cos_all_stock=cos(flat(stock[2]), flat(grads_r)))
    return res


def _held_to_stock_bf16(res, slack_cos, slack_err):
    """Every parameter gradient is as close to fp32 as stock PyTorch's bf16 autocast gets on the same network."""
    bad = [(n, e, c, es, cs) for n, _, e, c, es, cs in res["rows"] if c < cs - slack_cos or e > es * 1.25 + slack_err]
    assert not bad, bad[:5]
    assert res["logit_err"] <= 1.25 * res["logit_err_stock"] + 2e-3, (res["logit_err"], res["logit_err_stock"])
    assert res["loss_err"] <= 1.5 * res["loss_err_stock"] + 2e-3, (res["loss_err"], res["loss_err_stock"])
    assert res["cos_all"] >= res["cos_all_stock"] - slack_cos


def test_resnet50_whole_network_matches_fp32_cpu_model():
    """The benchmarked network as a whole against fp32 PyTorch with SHARED weights (the reference's only model check is a
    whole-model forward, /root/reference/src/pytorch/CNN/model.py:257-292): logits, loss and every one of the 161
    parameter gradients of one training step at batch 32, 224x224.

    Two weight states.  (a) residual branches scaled to 0.1 (every block's last batch-norm weight): a well-conditioned
    network, where bf16 keeps every gradient's direction - absolute thresholds on every tensor.  (b) the default
    initialisation: at random weights the deep features of all images nearly coincide and batch-norm divides by that
    small spread, so bf16 rounding ALONE (stock PyTorch autocast, measured in the same test) turns the early-layer
    gradients by ~80 degrees; there our model is held to what stock bf16 achieves, tensor by tensor.
    ``profiles/r3_whole_network_parity.txt`` holds both tables."""
    res = _whole_network_parity("resnet50", 32, 224, branch_gain=0.1)
    rows = res["rows"]
    worst = sorted(rows, key=lambda r: r[3])[:3]
    print("resnet50 (branch gain 0.1) logits rel err", res["logit_err"], "loss err", res["loss_err"], "all-gradient cosine",
          res["cos_all"], "worst", worst)
    assert res["logit_err"] < 2e-2 and res["loss_err"] < 1e-2
    cos = sorted(r[3] for r in rows)
    assert cos[0] > 0.85, worst                                   # every tensor
    assert cos[len(cos) // 2] > 0.95 and res["cos_all"] > 0.95    # the typical tensor, the gradient as a whole
    _held_to_stock_bf16(res, 0.05, 0.05)
    res = _whole_network_parity("resnet50", 32, 224)
    print("resnet50 (default init) logits rel err", res["logit_err"], "stock bf16", res["logit_err_stock"],
          "all-gradient cosine", res["cos_all"], "stock bf16", res["cos_all_stock"])
    late = {r[0]: r for r in res["rows"]}
    assert late["fc.weight"][3] > 0.98 and late["fc.bias"][3] > 0.999 and late["layer4.2.bn3.bias"][3] > 0.98
    _held_to_stock_bf16(res, 0.25, 0.25)


def test_vgg16_whole_network_matches_fp32_cpu_model():
    """VGG-16 (conv + bias + ReLU chains, max-pools, the 25088 x 4096 classifier) at batch 8, 224x224, dropout off:
    every gradient keeps its direction (bf16 error grows from 1 % at the classifier to ~40 % at the first conv, the
    same for stock bf16 autocast), and is as close to fp32 as stock bf16 gets."""
    res = _whole_network_parity("vgg16", 8, 224)
    rows = res["rows"]
    worst = sorted(rows, key=lambda r: r[3])[:3]
    print("vgg16 logits rel err", res["logit_err"], "loss err", res["loss_err"], "all-gradient cosine", res["cos_all"],
          "worst", worst)
    assert res["logit_err"] < 2e-2 and res["loss_err"] < 5e-3
    cos = sorted(r[3] for r in rows)
    assert cos[0] > 0.88, worst
    assert cos[len(cos) // 2] > 0.97 and res["cos_all"] > 0.99
    _held_to_stock_bf16(res, 0.03, 0.05)
