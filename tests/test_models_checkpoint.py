import os

import pytest
import torch

from distributed_deep_learning_amd import checkpoint as ckpt
from distributed_deep_learning_amd import models
from distributed_deep_learning_amd.optim import FusedSGD
from distributed_deep_learning_amd.parallel import ParamArena


def _count(m):
    return sum(p.numel() for p in m.parameters()), len(list(m.parameters()))


def test_parameter_census():
    # SURVEY Appendix B / §2.5
    assert _count(models.resnet50()) == (25557032, 161)
    assert _count(models.resnet18_cifar()) == (11173962, 62)
    assert _count(models.resnet18()) == (11689512, 62)
    assert _count(models.densenet_ref()) == (208838, 32)
    assert _count(models.mlp_ref()) == (3539, 6)
    assert _count(models.lstm_ref()) == (84293, 8)
    assert _count(models.lenet5()) == (61706, 10)


@pytest.mark.slow
def test_vgg16_census():
    assert _count(models.vgg16()) == (138357544, 32)


def _torchvision_resnet_keys(block_counts, bottleneck):
    keys = ["conv1.weight"] + [f"bn1.{s}" for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    bn = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")
    nconv = 3 if bottleneck else 2
    inplanes, exp = 64, 4 if bottleneck else 1
    for li, (n, planes) in enumerate(zip(block_counts, (64, 128, 256, 512)), start=1):
        for b in range(n):
            for c in range(1, nconv + 1):
                keys.append(f"layer{li}.{b}.conv{c}.weight")
                keys += [f"layer{li}.{b}.bn{c}.{s}" for s in bn]
            stride = 1 if li == 1 else 2
            if b == 0 and (stride != 1 or inplanes != planes * exp):
                keys.append(f"layer{li}.{b}.downsample.0.weight")
                keys += [f"layer{li}.{b}.downsample.1.{s}" for s in bn]
            inplanes = planes * exp
    return keys + ["fc.weight", "fc.bias"]


def test_resnet_state_dict_keys_and_shapes_match_torchvision():
    sd = models.resnet50().state_dict()
    assert list(sd.keys()) == _torchvision_resnet_keys([3, 4, 6, 3], True)
    assert tuple(sd["conv1.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["layer1.0.conv1.weight"].shape) == (64, 64, 1, 1)
    assert tuple(sd["layer1.0.conv2.weight"].shape) == (64, 64, 3, 3)
    assert tuple(sd["layer1.0.downsample.0.weight"].shape) == (256, 64, 1, 1)
    assert tuple(sd["layer2.0.conv2.weight"].shape) == (128, 128, 3, 3)
    assert tuple(sd["layer4.2.conv3.weight"].shape) == (2048, 512, 1, 1)
    assert tuple(sd["fc.weight"].shape) == (1000, 2048) and sd["fc.weight"].dtype == torch.float32
    sd18 = models.resnet18().state_dict()
    assert list(sd18.keys()) == _torchvision_resnet_keys([2, 2, 2, 2], False)


def test_vgg_state_dict_keys_match_torchvision():
    m = models.vgg11(input_size=32, hidden=64)
    keys = list(m.state_dict().keys())
    conv_idx = [0, 3, 6, 8, 11, 13, 16, 18]                 # torchvision vgg11 feature indices
    want = [f"features.{i}.{s}" for i in conv_idx for s in ("weight", "bias")]
    want += [f"classifier.{i}.{s}" for i in (0, 3, 6) for s in ("weight", "bias")]
    assert keys == want
    # VGG-16 (config D) conv indices
    from distributed_deep_learning_amd.models.vgg import CFG_D
    idx, i = [], 0
    for v in CFG_D:
        if v == "M":
            i += 1
        else:
            idx.append(i)
            i += 2
    assert idx == [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]


def test_state_dict_loads_into_stock_torch_modules_and_back():
    """A checkpoint is interchangeable with stock PyTorch layers of the same architecture."""
    import torch.nn as tnn
    ours = models.densenet_ref()
    sd = ours.state_dict()
    # rebuild the stem with stock modules and load our tensors
    conv = tnn.Conv2d(3, 64, 7, 2, 3, bias=False)
    conv.load_state_dict({"weight": sd["conv0.weight"]})
    bn = tnn.BatchNorm2d(64, eps=1e-3)
    bn.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("norm0.")})
    x = torch.randn(2, 3, 64, 64)
    ours.eval(); bn.eval()
    assert torch.allclose(ours.norm0(ours.conv0(x)), bn(conv(x)), atol=1e-5)


def test_checkpoint_roundtrip_with_arena(tmp_path):
    torch.manual_seed(0)
    m = models.build("densenet_ref")
    arena = ParamArena(m.named_parameters())
    opt = FusedSGD(arena, lr=0.1, momentum=0.9)
    x = torch.randn(4, 3, 64, 64)
    m(x).sum().backward(); opt.step()
    path = os.path.join(tmp_path, "ck", "model.pt")
    ckpt.save(path, m, optimizer=opt, epoch=3, step=17, extra={"note": "x"})
    assert os.path.exists(path) and not [f for f in os.listdir(os.path.dirname(path)) if ".tmp." in f]

    torch.manual_seed(1)
    m2 = models.build("densenet_ref")
    arena2 = ParamArena(m2.named_parameters())
    opt2 = FusedSGD(arena2, lr=0.5, momentum=0.9)
    meta = ckpt.load(path, m2, optimizer=opt2)
    assert meta["epoch"] == 3 and meta["step"] == 17 and meta["extra"]["note"] == "x"
    arena2.check_layout()                                   # load_state_dict copied INTO the arena views
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert torch.equal(opt.flat_state["momentum"], opt2.flat_state["momentum"]) and opt2.lr == 0.1
    # tensors-only loader that executes nothing from the file
    sd = ckpt.load_model_state(path)
    assert set(sd.keys()) == set(m.state_dict().keys())


@pytest.mark.parametrize("name,shape,out", [("lenet5", (2, 1, 28, 28), (2, 10)), ("resnet18_cifar", (2, 3, 32, 32), (2, 10)),
                                            ("densenet_ref", (2, 3, 64, 64), (2, 6)), ("mlp_ref", (3, 48), (3, 5)),
                                            ("lstm_ref", (3, 10, 32), (3, 5))])
def test_forward_shapes(name, shape, out):
    m = models.build(name)
    assert tuple(m(torch.randn(*shape)).shape) == out


def test_reference_defects_fixed_in_models():
    m = models.densenet_ref()
    y = m(torch.randn(2, 3, 64, 64))
    assert not torch.allclose(y.sum(1), torch.ones(2))     # logits, not softmax probabilities (A5)
    with pytest.raises(ValueError):
        models.densenet_ref(dense_blocks=0)                 # A13: the reference builds the ValueError but never raises
    with pytest.raises(ValueError):
        models.mlp_ref(hidden_layers=0)
