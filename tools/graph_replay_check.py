"""Eager twin vs hipGraph-captured epoch on PPI-shaped graphs: per-epoch losses and parameter equality.
args: (none) = nn.CrossEntropyLoss [shows the large-reduction replay hazard, dgll_amd/graphs.py] | two_stage | ops"""
import sys, os, copy, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "examples", "ppi"))
from train_gcn import synthetic_split
from dgll_amd.evaluation.ppi import GCN
from dgll_amd.graphs import GraphedTrainStep
dev = torch.device("cuda:0")
train = [(e.to(dev), x.to(dev), y.to(dev)) for e, x, y in synthetic_split(20, 0)]
torch.manual_seed(0)
base = GCN(50, 64, 121, 2).to(dev)
with torch.no_grad():
    for l in base.layers: l.weight.mul_(1.0 / l.weight.shape[0] ** 0.5)
from dgll_amd import ops
crit = torch.nn.CrossEntropyLoss() if len(sys.argv) < 2 else ((lambda out, y: -(torch.log_softmax(out, 1) * y).sum(1).mean()) if sys.argv[1] == 'two_stage' else ops.cross_entropy)
E = copy.deepcopy(base); G = copy.deepcopy(base)
oe = torch.optim.Adam(E.parameters(), lr=0.01, capturable=True)
og = torch.optim.Adam(G.parameters(), lr=0.01, capturable=True)
def eager_epoch():
    tot = 0.0
    for e, x, y in train:
        oe.zero_grad(); l = crit(E(e, x), y); l.backward(); oe.step(); tot += float(l.detach())
    return tot
e0 = eager_epoch()                                    # = the warm-up epoch of the graphed twin
ep = GraphedTrainStep([lambda e=e, x=x, y=y: crit(G(e, x), y) for e, x, y in train], og, warmup=1)
for k in range(5):
    te = eager_epoch(); ep(); tg = float(ep.total)
    d = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(E.parameters(), G.parameters()))
    print("epoch", k + 1, "eager", te, "graphed", tg, "max rel param diff", d)

