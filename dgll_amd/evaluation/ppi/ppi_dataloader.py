"""Evaluation/PPI/ppi_dataloader.py:10-69 -- same two names, vectorised (dgll_amd/data/formats.py)."""
from ...data.formats import load_ppi_dataset  # noqa: F401


def remove_self_loops(edge_index):
    """ppi_dataloader.py:64-69."""
    return edge_index[:, edge_index[0] != edge_index[1]]
