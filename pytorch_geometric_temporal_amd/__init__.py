"""MI355X-native (gfx950 / CDNA4) drop-in for the message-passing hot path of torch_geometric_temporal.nn.

    from pytorch_geometric_temporal_amd.nn.recurrent import DCRNN, BatchedDCRNN   # same API as the reference

The kernels live in pytorch_geometric_temporal_amd/lib/libpgt_hip.so (C ABI: include/pgt_hip.h), built by
`python -m pytorch_geometric_temporal_amd._build`.  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import nn, signal  # noqa: F401
