from .dcrnn import DConv, DCRNN, BatchedDConv, BatchedDCRNN  # noqa: F401

__all__ = ["DConv", "DCRNN", "BatchedDConv", "BatchedDCRNN"]
