"""Where the host time of the launch-bound configs goes: cProfile of one eager epoch of BASELINE configs[0] (Chickenpox
DCRNN) and configs[4] (England-Covid EvolveGCN-H), the loops of bench_configs.py.  Prints the top functions by own time."""
import cProfile
import io
import pstats
import sys
import time

import torch
import torch.nn.functional as TF

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytorch_geometric_temporal_amd.dataset import ChickenpoxDatasetLoader, EnglandCovidDatasetLoader
from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN, EvolveGCNH
from pytorch_geometric_temporal_amd.signal import temporal_signal_split

dev = torch.device("cuda:0")


def profile(name, epoch, n=3):
    for _ in range(3):
        epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        epoch()
    torch.cuda.synchronize()
    print(f"== {name}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per eager epoch")
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(n):
        epoch()
    torch.cuda.synchronize()
    print(f"   with the cyclic collector off: {1e3 * (time.perf_counter() - t0) / n:.2f} ms")
    gc.enable()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        epoch()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])


class M1(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.recurrent = DCRNN(4, 32, 1)
        self.linear = torch.nn.Linear(32, 1)

    def forward(self, x, e, w):
        return self.linear(TF.relu(self.recurrent(x, e, w)))


class M5(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.recurrent = EvolveGCNH(129, 8)
        self.linear = torch.nn.Linear(8, 1)

    def forward(self, x, e, w):
        return self.linear(TF.relu(self.recurrent(x, e, w)))


def make_epoch(model, snaps, reinit):
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)

    def epoch():
        if reinit:
            model.recurrent.reinitialize_weight()
        cost = 0
        for x, e, w, y in snaps:
            cost = cost + torch.mean((model(x, e, w).view(-1) - y.view(-1)) ** 2)
        cost = cost / len(snaps)
        for p in model.parameters():
            p.grad.zero_()
        cost.backward()
        opt.step()
    return epoch


train, _ = temporal_signal_split(ChickenpoxDatasetLoader().get_dataset(), train_ratio=0.2)
train = train.to(dev)
snaps = [(s.x, s.edge_index, s.edge_attr, s.y) for s in train]
torch.manual_seed(0)
profile("config 1 (Chickenpox DCRNN K=1)", make_epoch(M1().to(dev), snaps, False))
ds = EnglandCovidDatasetLoader().get_dataset(lags=8)
snaps = [tuple(t.to(dev) for t in (s.x, s.edge_index, s.edge_attr, s.y)) for s in ds]
profile("config 5 (Covid EvolveGCN-H)", make_epoch(M5().to(dev), snaps, True))
