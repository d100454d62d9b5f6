from .stgcn import TemporalConv, STConv  # noqa: F401
from .astgcn import ChebConvAttention, SpatialAttention, TemporalAttention, ASTGCNBlock, ASTGCN  # noqa: F401
from .mstgcn import MSTGCNBlock, MSTGCN  # noqa: F401

__all__ = ["TemporalConv", "STConv", "ChebConvAttention", "SpatialAttention", "TemporalAttention", "ASTGCNBlock",
           "ASTGCN", "MSTGCNBlock", "MSTGCN"]
