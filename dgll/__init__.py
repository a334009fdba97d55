"""Drop-in `dgll` namespace: the reference's import paths (`from dgll import backend as F`,
`dgll.nn.Convolution.gcnconv`, `dgll.data.dgraph`, `dgll.sampling.dgllsampler`, `dgll.dataloader`) resolved onto
the MI355X-native implementation in dgll_amd.  /root/reference/dgll/__init__.py:1 is `import torch as backend`;
here `backend` is dgll_amd.backend (torch + the missing aliases + HIP aggregation)."""
import importlib
import sys

import dgll_amd
from dgll_amd import backend  # noqa: F401

_ALIASES = {
    "dgll.backend": "dgll_amd.backend",
    "dgll.nn": "dgll_amd.nn",
    "dgll.nn.Convolution": "dgll_amd.nn.Convolution",
    "dgll.nn.Convolution.gcnconv": "dgll_amd.nn.Convolution.gcnconv",
    "dgll.nn.Convolution.gcn": "dgll_amd.nn.Convolution.gcn",
    "dgll.nn.Convolution.sageconv": "dgll_amd.nn.Convolution.sageconv",
    "dgll.nn.Convolution.gatconv": "dgll_amd.nn.Convolution.gatconv",
    "dgll.nn.Convolution.ginconv": "dgll_amd.nn.Convolution.ginconv",
    "dgll.nn.GlobalPooling": "dgll_amd.nn.GlobalPooling",
    "dgll.nn.GlobalPooling.Pooling": "dgll_amd.nn.GlobalPooling.Pooling",
    "dgll.data": "dgll_amd.data",
    "dgll.data.dgraph": "dgll_amd.data.dgraph",
    "dgll.sampling": "dgll_amd.sampling",
    "dgll.sampling.base_sampler": "dgll_amd.sampling.base_sampler",
    "dgll.sampling.dgllsampler": "dgll_amd.sampling.dgllsampler",
    "dgll.dataloader": "dgll_amd.dataloader",
}
for _alias, _target in _ALIASES.items():
    sys.modules[_alias] = importlib.import_module(_target)
nn = sys.modules["dgll.nn"]
__version__ = dgll_amd.__version__
