"""Working replacement for the reference's broken package file (dgll/nn/Convolution/__init__.py:1-7 uses
absolute imports that fail); exports the same names plus the GAT/SAGE classes the modules define."""
from .gcnconv import gcnConv, GCN
from .gcn import GraphConvolution
from .sageconv import NeighborAggregator, sageConv, GraphSage
from .gatconv import gatConv, sparseGatConv, SpecialSpmm, SpecialSpmmFunction, GAT, SpGAT
from .ginconv import GinConv, GIN

__all__ = ["gcnConv", "sageConv", "gatConv", "sparseGatConv", "GinConv", "GCN", "GIN",
           "GraphConvolution", "NeighborAggregator", "GraphSage", "SpecialSpmm", "SpecialSpmmFunction", "GAT", "SpGAT"]
