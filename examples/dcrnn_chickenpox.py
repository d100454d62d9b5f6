"""BASELINE.json configs[0] on an MI355X: the reference's examples/recurrent/dcrnn_example.py:1-60 with the imports
swapped (model, dataset loader, signal split) and the data moved to the GPU once.  The Chickenpox dataset comes from
the packaged .pgtc cache (no network).

    python examples/dcrnn_chickenpox.py [epochs]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd.dataset import ChickenpoxDatasetLoader  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN  # noqa: E402
from pytorch_geometric_temporal_amd.signal import temporal_signal_split  # noqa: E402


class RecurrentGCN(torch.nn.Module):
    def __init__(self, node_features):
        super().__init__()
        self.recurrent = DCRNN(node_features, 32, 1)
        self.linear = torch.nn.Linear(32, 1)

    def forward(self, x, edge_index, edge_weight):
        h = self.recurrent(x, edge_index, edge_weight)
        return self.linear(F.relu(h))


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    device = torch.device("cuda:0")
    dataset = ChickenpoxDatasetLoader().get_dataset()
    train_dataset, test_dataset = temporal_signal_split(dataset, train_ratio=0.2)
    train_dataset, test_dataset = train_dataset.to(device), test_dataset.to(device)   # one upload, snapshots are views
    torch.manual_seed(0)
    model = RecurrentGCN(node_features=4).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)
    model.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for epoch in range(epochs):
        cost = 0
        for step, snapshot in enumerate(train_dataset):
            y_hat = model(snapshot.x, snapshot.edge_index, snapshot.edge_attr)
            cost = cost + torch.mean((y_hat - snapshot.y) ** 2)
        cost = cost / (step + 1)
        cost.backward()
        optimizer.step()
        optimizer.zero_grad()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.eval()
    cost = 0
    with torch.no_grad():
        for step, snapshot in enumerate(test_dataset):
            y_hat = model(snapshot.x, snapshot.edge_index, snapshot.edge_attr)
            cost = cost + torch.mean((y_hat - snapshot.y) ** 2)
    print(f"{epochs} epochs of {train_dataset.snapshot_count} snapshots: {dt / epochs * 1e3:.1f} ms / epoch; "
          f"train MSE {float(cost_train(model, train_dataset)):.4f}, test MSE {float(cost / (step + 1)):.4f}")


def cost_train(model, ds):
    with torch.no_grad():
        c = 0
        for step, snapshot in enumerate(ds):
            c = c + torch.mean((model(snapshot.x, snapshot.edge_index, snapshot.edge_attr) - snapshot.y) ** 2)
    return c / (step + 1)


if __name__ == "__main__":
    main()
