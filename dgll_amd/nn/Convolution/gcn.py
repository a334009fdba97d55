"""`GraphConvolution`: the reference keeps a second copy of the GCN layer under this name
(/root/reference/dgll/nn/Convolution/gcn.py:17-43, model :51-66).  Same kernel path as gcnConv."""
from .gcnconv import gcnConv, GCN  # noqa: F401


class GraphConvolution(gcnConv):
    pass
