"""BASELINE configs[0] and configs[4] alone (bench_configs.chickenpox_epoch / covid_epoch): eager, graphed, CPU oracle."""
import json
import os
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench_configs

dev = torch.device("cuda:0")
cores = min(32, os.cpu_count() or 1)
for name, fn in (("config1_chickenpox", bench_configs.chickenpox_epoch), ("config5_covid_evolvegcnh", bench_configs.covid_epoch)):
    print(json.dumps({name: fn(dev, cores)}))
