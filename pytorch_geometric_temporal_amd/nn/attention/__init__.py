from .stgcn import TemporalConv, STConv  # noqa: F401
from .astgcn import ChebConvAttention  # noqa: F401

__all__ = ["TemporalConv", "STConv", "ChebConvAttention"]
