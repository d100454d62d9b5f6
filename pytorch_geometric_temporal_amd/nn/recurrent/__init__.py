from .dcrnn import DConv, DCRNN, BatchedDConv, BatchedDCRNN  # noqa: F401
from .temporalgcn import TGCN, TGCN2  # noqa: F401
from .attentiontemporalgcn import A3TGCN, A3TGCN2  # noqa: F401
from .evolvegcn import EvolveGCNH, EvolveGCNO, GCNConv_Fixed_W  # noqa: F401
from .chebcells import GConvGRU, GConvLSTM, GCLSTM  # noqa: F401

__all__ = ["DConv", "DCRNN", "BatchedDConv", "BatchedDCRNN", "TGCN", "TGCN2", "A3TGCN", "A3TGCN2", "EvolveGCNH",
           "EvolveGCNO", "GCNConv_Fixed_W", "GConvGRU", "GConvLSTM", "GCLSTM"]
