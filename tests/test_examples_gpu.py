"""The example scripts run end to end on the GPU (small sizes): the full-graph trainer and the mini-batch trainer in both consumer
modes (launch by launch / one HIP graph per batch) -- the loss falls and the held-out accuracy beats chance."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", *script)] + list(args), capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    return res.stdout


def test_full_graph_example(cuda_device):
    out = _run(("graphsage", "train_full.py"), "--nodes", "20000", "--epochs", "12")
    acc = [float(v) for v in re.findall(r"held-out acc ([0-9.]+)", out)]
    assert len(acc) == 12 and acc[-1] > 0.1 and acc[-1] > acc[0]


@pytest.mark.parametrize("graph", [False, True])
def test_minibatch_example(cuda_device, graph):
    out = _run(("graphsage", "train_minibatch.py"), "--nodes", "30000", "--epochs", "2", "--batch", "256", "--fanouts", "6,5,4",
               "--sampler-threads", "2", *(["--hip-graph"] if graph else []))
    loss = [float(v) for v in re.findall(r"loss ([0-9.]+)", out)]
    acc = [float(v) for v in re.findall(r"held-out acc ([0-9.]+)", out)]
    assert len(acc) == 2 and loss[-1] < loss[0] and acc[-1] > 2.0 / 16
    if graph:
        assert "(HIP graph)" in out
