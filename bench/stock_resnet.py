"""Same-box comparator: stock PyTorch-ROCm ResNet-50 (MIOpen convs, ATen BN, bf16 autocast, channels_last)."""
import sys, time
import torch, torch.nn as nn, torch.nn.functional as F

class Bottleneck(nn.Module):
    def __init__(s, inp, pl, stride=1, down=None):
        super().__init__()
        s.conv1 = nn.Conv2d(inp, pl, 1, bias=False); s.bn1 = nn.BatchNorm2d(pl)
        s.conv2 = nn.Conv2d(pl, pl, 3, stride, 1, bias=False); s.bn2 = nn.BatchNorm2d(pl)
        s.conv3 = nn.Conv2d(pl, pl * 4, 1, bias=False); s.bn3 = nn.BatchNorm2d(pl * 4)
        s.downsample = down
    def forward(s, x):
        idt = x
        o = F.relu(s.bn1(s.conv1(x)), inplace=True)
        o = F.relu(s.bn2(s.conv2(o)), inplace=True)
        o = s.bn3(s.conv3(o))
        if s.downsample is not None: idt = s.downsample(x)
        return F.relu(o + idt, inplace=True)

class ResNet50(nn.Module):
    def __init__(s, nc=1000):
        super().__init__()
        s.inp = 64
        s.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False); s.bn1 = nn.BatchNorm2d(64)
        s.maxpool = nn.MaxPool2d(3, 2, 1)
        s.layer1 = s._ml(64, 3, 1); s.layer2 = s._ml(128, 4, 2); s.layer3 = s._ml(256, 6, 2); s.layer4 = s._ml(512, 3, 2)
        s.fc = nn.Linear(2048, nc)
    def _ml(s, pl, n, stride):
        down = None
        if stride != 1 or s.inp != pl * 4:
            down = nn.Sequential(nn.Conv2d(s.inp, pl * 4, 1, stride, bias=False), nn.BatchNorm2d(pl * 4))
        L = [Bottleneck(s.inp, pl, stride, down)]; s.inp = pl * 4
        L += [Bottleneck(s.inp, pl) for _ in range(1, n)]
        return nn.Sequential(*L)
    def forward(s, x):
        x = s.maxpool(F.relu(s.bn1(s.conv1(x)), inplace=True))
        x = s.layer4(s.layer3(s.layer2(s.layer1(x))))
        return s.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True
    model = ResNet50().to(dev).to(memory_format=torch.channels_last)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (B,), device=dev)
    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(model(x), y)
        loss.backward(); opt.step()
        return loss
    import threading
    def _hb():
        n = 0
        while not done.wait(30):
            n += 30; print(f"[stock] warming up (MIOpen find) {n}s", flush=True)
    done = threading.Event(); threading.Thread(target=_hb, daemon=True).start()
    for i in range(5):
        l = step(); torch.cuda.synchronize(); print("[stock] warm step", i, flush=True)
    done.set()
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(steps): l = step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    print(f"STOCK B={B} {dt*1e3:.2f} ms/step  {B/dt:.1f} img/s  loss {float(l.detach()):.4f}", flush=True)
