"""GIN layer (/root/reference/dgll/nn/Convolution/ginconv.py:10-66): relu(Linear(X + A@X)) on BATCHED DENSE
adjacencies with a concatenate-and-sum read-out.  Outside the sparse hot path --
kept only so `dgll.nn.Convolution.__all__` resolves; it is plain dense torch, as in the reference."""
from ... import backend as F


class GinConv(F.nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.linear = F.nn.Linear(hidden_dim, hidden_dim)

    def forward(self, Adj, Feat):
        """Adj [batch, nodes, nodes], Feat [batch, nodes, features] -> [batch, nodes, features] (ginconv.py:16-30)."""
        return F.relu(self.linear(Feat + Adj @ Feat))


class GIN(F.nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, n_layers):
        super().__init__()
        self.in_proj = F.nn.Linear(input_dim, hidden_dim)
        self.convs = F.nn.ModuleList([GinConv(hidden_dim) for _ in range(n_layers)])
        self.out_proj = F.nn.Linear(hidden_dim * (1 + n_layers), output_dim)   # ginconv.py:51

    def forward(self, A, X):
        X = self.in_proj(X)
        states = [X]
        for conv in self.convs:
            X = conv(A, X)
            states.append(X)
        return self.out_proj(F.cat(states, dim=2).sum(dim=1))   # ginconv.py:62-64
