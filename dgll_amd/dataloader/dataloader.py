"""Mini-batch iterator over seed nodes (/root/reference/dgll/dataloader/dataloader.py:4-24, as graphage.py:34-44 uses
it).  The reference file references undefined names (`self.data`, `batch_size`, `from dgllsampler import *`); this is
the loop it intends: slices of `train_nodes` of length `batch_size`, each handed to `sampler.sample(Dgraph, seeds)`."""


class DataLoader:
    def __init__(self, Dgraph, train_nodes, sampler, batch_size=1, device=None):
        self.Dgraph = Dgraph
        self.sampler = sampler
        self.train_nodes = train_nodes
        self.batch_size = batch_size
        self.device = device

    def sample(self):
        for i in range(0, len(self.train_nodes), self.batch_size):
            seed_nodes = self.train_nodes[i:i + self.batch_size]
            yield self.sampler.sample(self.Dgraph, seed_nodes)

    def __iter__(self):
        return self.sample()

    def __len__(self):
        return (len(self.train_nodes) + self.batch_size - 1) // self.batch_size
