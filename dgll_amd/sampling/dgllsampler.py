"""Multi-hop neighbour sampler (/root/reference/dgll/sampling/dgllsampler.py:5-21)."""
from .base_sampler import Base_sampler


class DGLLNeighborSampler(Base_sampler):
    def __init__(self, fanouts):
        super().__init__()
        self.fanouts = fanouts

    def sample(self, g, seed_nodes):
        """Outermost hop first: for each fan-out in reversed order sample around the current seeds; the next seeds are
        the sampled sources WITH duplicates (dgllsampler.py:14-19).  Returns (input_nodes, output_nodes, subgs).  An
        empty fan-out list raises UnboundLocalError exactly as the reference does (dgllsampler.py:21); the stray
        print of dgllsampler.py:13 is not reproduced."""
        output_nodes = seed_nodes
        subgs = []
        for fanout in reversed(self.fanouts):
            subg = self.sample_neighbours(g, seed_nodes, fanout)
            seed_nodes = subg.src_nodes()
            subgs.insert(0, subg)
            input_nodes = seed_nodes
        return input_nodes, output_nodes, subgs
