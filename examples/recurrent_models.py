#!/usr/bin/env python
"""The reference's per-model training scripts (examples/recurrent/{dcrnn,tgcn,a3tgcn,evolvegcnh,evolvegcno,gconvgru,gconvlstm,
gclstm}_example.py) as ONE script on the drop-in modules: the same RecurrentGCN wrapper (cell -> relu -> Linear), the same
cumulative-MSE epoch over the Chickenpox snapshots, Adam(lr = 0.01) — only the import lines differ from the reference
(`torch_geometric_temporal` -> `pytorch_geometric_temporal_amd`), the dataset comes from the packaged cache (no network) and is
moved to the GPU once.  `--graph` records the whole epoch (forward over all snapshots, loss, backward, update) as one hipGraph:
the documented path for graphs of tens of nodes, where an eager snapshot costs more host time than GPU time.

    python examples/recurrent_models.py --model tgcn --epochs 20
    python examples/recurrent_models.py --model gconvlstm --epochs 20 --graph
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_temporal_amd.dataset import ChickenpoxDatasetLoader  # noqa: E402
from pytorch_geometric_temporal_amd.nn import recurrent as R  # noqa: E402
from pytorch_geometric_temporal_amd.signal import temporal_signal_split  # noqa: E402

LAGS = 4

# name -> (cell constructor, "h": the cell carries its state as an argument / "hc": state pair / None: stateless call)
MODELS = {
    "dcrnn": (lambda: R.DCRNN(LAGS, 32, 1), None),                          # dcrnn_example.py:19-21
    "tgcn": (lambda: R.TGCN(LAGS, 32), None),                               # tgcn_example.py
    "a3tgcn": (lambda: R.A3TGCN(LAGS, 32, 1), None),                        # a3tgcn_example.py (periods = 1 on lagged features)
    "evolvegcnh": (lambda: R.EvolveGCNH(20, LAGS), None),                   # evolvegcnh_example.py (20 counties)
    "evolvegcno": (lambda: R.EvolveGCNO(LAGS), None),                       # evolvegcno_example.py
    "gconvgru": (lambda: R.GConvGRU(LAGS, 32, 2), None),                    # gconvgru_example.py
    "gconvlstm": (lambda: R.GConvLSTM(LAGS, 32, 2), "hc"),                  # gconvlstm_example.py: (h, c) threaded through the epoch
    "gclstm": (lambda: R.GCLSTM(LAGS, 32, 2), "hc"),                        # gclstm_example.py
}


class RecurrentGCN(torch.nn.Module):
    def __init__(self, name):
        super().__init__()
        make, self.state = MODELS[name]
        self.name = name
        self.recurrent = make()
        width = LAGS if name.startswith("evolvegcn") else 32
        self.linear = torch.nn.Linear(width, 1)

    def forward(self, x, edge_index, edge_weight, h=None, c=None):
        if self.name == "a3tgcn":
            h_out = self.recurrent(x.view(x.shape[0], LAGS, 1), edge_index, edge_weight)
        elif self.state == "hc":
            h_out, c = self.recurrent(x, edge_index, edge_weight, h, c)
        else:
            h_out = self.recurrent(x, edge_index, edge_weight)
        return self.linear(F.relu(h_out)), h_out, c


def run_epoch(model, dataset, optimizer=None):
    """One pass over the snapshots; with an optimizer: the reference's update on the epoch's mean squared error."""
    cost, h, c = 0, None, None
    for step, snapshot in enumerate(dataset):
        y_hat, h, c = model(snapshot.x, snapshot.edge_index, snapshot.edge_attr, h, c)
        cost = cost + torch.mean((y_hat.squeeze(-1) - snapshot.y) ** 2)
    cost = cost / (step + 1)
    if optimizer is not None:
        cost.backward()
        optimizer.step()
        optimizer.zero_grad()
    if model.name.startswith("evolvegcn") and model.recurrent.weight is not None:
        # cut the graph of the evolved weight between epochs (evolvegcnh_example.py:49-50)
        model.recurrent.weight = model.recurrent.weight.detach()
    return cost.detach()


def main(argv=None, device=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=sorted(MODELS), default="dcrnn")
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--snapshots", type=int, default=0, help="use only the first N training and N test snapshots (0 = all)")
    ap.add_argument("--graph", action="store_true", help="the whole training epoch as one hipGraph")
    args = ap.parse_args(argv)
    device = device or torch.device("cuda:0")
    dataset = ChickenpoxDatasetLoader().get_dataset(lags=LAGS)
    train, test = temporal_signal_split(dataset, train_ratio=0.2)
    if args.snapshots:
        train, test = train[:args.snapshots], test[:args.snapshots]
    train, test = train.to(device), test.to(device)
    torch.manual_seed(0)
    model = RecurrentGCN(args.model).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01, **({"capturable": True} if args.graph else {}))
    step = lambda: run_epoch(model, train, optimizer)          # noqa: E731
    if args.graph:
        if args.model.startswith("evolvegcn"):
            # its evolved weight is module state carried ACROSS epochs: a replayed capture would keep reading the buffer it saw
            raise SystemExit("--graph: EvolveGCN carries its weight across epochs; see tests/test_graphed.py for the captured form")
        from pytorch_geometric_temporal_amd.graphed import GraphedStep
        step = GraphedStep(step, [], warmup=2)
    if device.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.epochs):
        cost = step()
    if device.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.eval()
    with torch.no_grad():
        test_cost = run_epoch(model, test)
    print(f"{args.model}: {args.epochs} epochs of {train.snapshot_count} snapshots, {dt / max(args.epochs, 1) * 1e3:.1f} ms / epoch"
          f"{' (one hipGraph)' if args.graph else ''}; train MSE {float(cost):.4f}, test MSE {float(test_cost):.4f}")
    return float(cost), float(test_cost)


if __name__ == "__main__":
    main()
