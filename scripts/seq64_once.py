"""A few forward + backward passes of BatchedDCRNN(2, 64, 3) on the one-launch sequence kernels (for rocprofv3 counter passes):
seq64_once.py [B = 256] [reps = 3]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytorch_geometric_temporal_amd import ops
from pytorch_geometric_temporal_amd.dataset import synthetic as syn
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
ei_np, ew_np = syn.sensor_graph(207, 1515, seed=0, symmetric=False)
ei, ew = torch.from_numpy(ei_np).to(dev), torch.from_numpy(ew_np).to(dev)
torch.manual_seed(0)
model = BatchedDCRNN(2, 64, 3).to(dev)
ops.SEQ64_MIN_BATCH = 1
X = torch.randn(B, 12, 207, 2, device=dev)
w = torch.randn(B, 12, 207, 64, device=dev)
for _ in range(reps):
    model.zero_grad()
    (model(X, ei, ew) * w).sum().backward()
torch.cuda.synchronize()
