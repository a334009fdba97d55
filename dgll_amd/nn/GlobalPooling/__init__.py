"""Graph-level read-out (reference: dgll/nn/GlobalPooling/Pooling.py)."""
from .Pooling import Pooling, maxPooling, meanPooling, segments_of, sumPooling  # noqa: F401

__all__ = ["sumPooling", "meanPooling", "maxPooling", "Pooling"]
