#!/usr/bin/env python
"""Convert the in-scope datasets the reference vendors as JSON (dataset/chickenpox.json, england_covid.json: BASELINE configs[0]
and [4]) into the package's binary cache format (pytorch_geometric_temporal_amd/dataset/cache.py) so that the loaders work without a
network.  Run in the build container (reads /root/reference, writes pytorch_geometric_temporal_amd/dataset/data/).

    python scripts/make_dataset_cache.py [/path/to/reference/dataset]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd.dataset import _chickenpox_from_json, _covid_from_json  # noqa: E402
from pytorch_geometric_temporal_amd.dataset.cache import load_cache, save_cache  # noqa: E402


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/dataset"
    out = os.path.join(ROOT, "pytorch_geometric_temporal_amd", "dataset", "data")
    os.makedirs(out, exist_ok=True)
    for fname, conv, cname in (("chickenpox.json", _chickenpox_from_json, "chickenpox.pgtc"),
                               ("england_covid.json", _covid_from_json, "england_covid.pgtc")):
        with open(os.path.join(src, fname)) as f:
            c = conv(json.load(f))
        meta = dict(c.meta, source=f"benedekrozemberczki/pytorch_geometric_temporal dataset/{fname}")
        path = save_cache(os.path.join(out, cname), c.name, {k: np.asarray(v) for k, v in c.arrays.items()}, meta)
        back = load_cache(path)
        for k, v in c.arrays.items():
            assert np.array_equal(np.asarray(v), np.asarray(back.arrays[k])), k
        print(path, os.path.getsize(path), "bytes", {k: (str(v.dtype), v.shape) for k, v in back.arrays.items()})


if __name__ == "__main__":
    main()
