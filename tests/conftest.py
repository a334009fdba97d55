import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are selected with -m gpu; when selected on a box without a GPU they fail loudly rather than skip."""


class Golden:
    def __init__(self, name):
        data = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.arrays = {k: data[k] for k in data.files if k != "meta"}
        self.meta = json.loads(str(data["meta"]))

    def __getitem__(self, k):
        return self.arrays[k]

    def __contains__(self, k):
        return k in self.arrays

    def t(self, k, device="cpu", dtype=None):
        import torch

        x = torch.from_numpy(np.ascontiguousarray(self.arrays[k])).to(device)
        return x if dtype is None else x.to(dtype)


def load_golden(name):
    return Golden(name)


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def cuda_device():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")
