"""Same-box comparators in stock PyTorch-ROCm (MIOpen convs, ATen BN, bf16 autocast, channels_last, torch SGD):
ResNet-50 / ResNet-18 (CIFAR stem) / VGG-16.  Usage: python bench/stock_models.py MODEL BATCH [STEPS]"""
import sys
import threading
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, "bench")
from stock_resnet import ResNet50  # noqa: E402


class Basic(nn.Module):
    def __init__(s, inp, pl, stride=1):
        super().__init__()
        s.c1 = nn.Conv2d(inp, pl, 3, stride, 1, bias=False); s.b1 = nn.BatchNorm2d(pl)
        s.c2 = nn.Conv2d(pl, pl, 3, 1, 1, bias=False); s.b2 = nn.BatchNorm2d(pl)
        s.down = None
        if stride != 1 or inp != pl:
            s.down = nn.Sequential(nn.Conv2d(inp, pl, 1, stride, bias=False), nn.BatchNorm2d(pl))

    def forward(s, x):
        o = F.relu(s.b1(s.c1(x)), inplace=True)
        o = s.b2(s.c2(o))
        return F.relu(o + (x if s.down is None else s.down(x)), inplace=True)


class ResNet18Cifar(nn.Module):
    def __init__(s, nc=10):
        super().__init__()
        s.c = nn.Conv2d(3, 64, 3, 1, 1, bias=False); s.b = nn.BatchNorm2d(64)
        L, inp = [], 64
        for pl, st in ((64, 1), (128, 2), (256, 2), (512, 2)):
            L += [Basic(inp, pl, st), Basic(pl, pl)]
            inp = pl
        s.layers = nn.Sequential(*L)
        s.fc = nn.Linear(512, nc)

    def forward(s, x):
        x = s.layers(F.relu(s.b(s.c(x)), inplace=True))
        return s.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


class VGG16(nn.Module):
    def __init__(s, nc=1000):
        super().__init__()
        L, inp = [], 3
        for v in (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"):
            if v == "M":
                L.append(nn.MaxPool2d(2, 2))
            else:
                L += [nn.Conv2d(inp, v, 3, 1, 1), nn.ReLU(inplace=True)]
                inp = v
        s.features = nn.Sequential(*L)
        s.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Linear(4096, 4096), nn.ReLU(True),
                                     nn.Linear(4096, nc))

    def forward(s, x):
        return s.classifier(torch.flatten(s.features(x), 1))


if __name__ == "__main__":
    name = sys.argv[1]
    B = int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True
    model, img, nc = {"resnet50": (ResNet50, 224, 1000), "resnet18_cifar": (ResNet18Cifar, 32, 10),
                      "vgg16": (VGG16, 224, 1000)}[name]
    model = model().to(dev).to(memory_format=torch.channels_last)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(B, 3, img, img, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, nc, (B,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        return loss

    done = threading.Event()

    def _hb():
        n = 0
        while not done.wait(30):
            n += 30
            print(f"[stock] warming up (MIOpen find) {n}s", flush=True)

    threading.Thread(target=_hb, daemon=True).start()
    for i in range(5):
        step()
        torch.cuda.synchronize()
        print("[stock] warm step", i, flush=True)
    done.set()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.freeze()             # as bench.py does for ours (utils/host.py): no full-GC pause inside the timed loop
    t0 = time.time()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    print(f"STOCK {name} B={B} {dt * 1e3:.2f} ms/step  {B / dt:.1f} img/s  loss {float(loss.detach()):.4f}", flush=True)
