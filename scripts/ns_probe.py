#!/usr/bin/env python
"""The north-star aggregation lines of bench.py on their own (same measurement: 60 launches over 6 rotating buffer pairs as one
hipGraph, mean of 3 replays).  usage: ns_probe.py [name ...]   (default: all seven numberings)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

names = tuple(sys.argv[1:]) or None
res = bench.spmm_roofline_ns(torch.device("cuda:0"), only=names)
for k, v in res.items():
    print(k, f"{v['us_per_launch']:.2f} us  frac {v['frac']:.4f}  ", v["kernel"])
    print("   replays:", [round(t, 2) for t in v["us_per_launch_replays"]])
print(json.dumps({k: {"us": v["us_per_launch"], "frac": v["frac"]} for k, v in res.items()}))
