"""cProfile of BASELINE configs[0]'s eager epoch on the GPU (103 Chickenpox snapshots through DCRNN(4, 32, K) + relu + Linear, one
backward, Adam): where the host time per snapshot goes.  usage: eager_epoch_profile.py [K = 1]"""
import cProfile
import pstats
import sys
import time

import torch
import torch.nn.functional as TF

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytorch_geometric_temporal_amd.dataset import ChickenpoxDatasetLoader
from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN
from pytorch_geometric_temporal_amd.signal import temporal_signal_split

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")


class RecurrentGCN(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.recurrent = DCRNN(4, 32, K)
        self.linear = torch.nn.Linear(32, 1)

    def forward(self, x, edge_index, edge_weight):
        return self.linear(TF.relu(self.recurrent(x, edge_index, edge_weight)))


train, _ = temporal_signal_split(ChickenpoxDatasetLoader().get_dataset(), train_ratio=0.2)
snaps = [(s.x, s.edge_index, s.edge_attr, s.y) for s in train.to(dev)]
torch.manual_seed(0)
model = RecurrentGCN().to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01)


def forward_only():
    cost = 0
    for x, e, w, y in snaps:
        cost = cost + torch.mean((model(x, e, w) - y) ** 2)
    return cost / len(snaps)


def epoch():
    cost = forward_only()
    opt.zero_grad()
    cost.backward()
    opt.step()


for _ in range(3):
    epoch()
torch.cuda.synchronize()
for name, fn in (("forward loop", forward_only), ("epoch", epoch)):
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms ({1e6 * (time.perf_counter() - t0) / 5 / len(snaps):.1f} us per snapshot)")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    forward_only()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
