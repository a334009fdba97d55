from .gcn_model import GCN, GCNLayer, create_sparse_adj  # noqa: F401
from .ppi_dataloader import load_ppi_dataset, remove_self_loops  # noqa: F401
