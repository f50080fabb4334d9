"""Model zoo: north-star image classifiers + the reference's three workloads."""
from .densenet_ref import DenseNetRef, densenet_ref
from .lenet import LeNet5, lenet5
from .lstm_ref import LSTMRef, lstm_ref
from .mlp_ref import MLPRef, mlp_ref
from .resnet import ResNet, resnet18, resnet18_cifar, resnet34, resnet50, resnet101
from .vgg import VGG, vgg11, vgg16, vgg16_bn

REGISTRY = {
    "lenet5": lenet5, "resnet18": resnet18, "resnet18_cifar": resnet18_cifar, "resnet34": resnet34,
    "resnet50": resnet50, "resnet101": resnet101, "vgg11": vgg11, "vgg16": vgg16, "vgg16_bn": vgg16_bn,
    "densenet_ref": densenet_ref, "cnn": densenet_ref, "mlp_ref": mlp_ref, "mlp": mlp_ref, "lstm_ref": lstm_ref,
    "lstm": lstm_ref,
}


def build(name: str, **kw):
    if name not in REGISTRY:
        raise KeyError(f"unknown model {name!r}; choose from {sorted(REGISTRY)}")
    return REGISTRY[name](**kw)
