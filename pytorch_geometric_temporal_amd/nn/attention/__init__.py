from .stgcn import TemporalConv, STConv  # noqa: F401

__all__ = ["TemporalConv", "STConv"]
