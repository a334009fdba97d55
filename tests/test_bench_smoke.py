"""bench.py end to end at small, odd shapes: one rank, and two ranks sharing the GPU (gloo transport) -- the JSON
contract, the roofline / cpu_baseline objects and the equality of the loss across rank counts."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
SHAPE = ["--nodes", "20000", "--undirected-edges", "200000", "--hidden", "64", "--classes", "10", "--in-feats", "40",
         "--steps", "2", "--warmup", "1", "--cpu-sample-rows", "5000"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


def test_bench_single_rank_contract():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SHAPE, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "edges/s" and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["traffic"] is None       # counters exist for the default shape only
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and "model" not in d["config"]
    test_bench_single_rank_contract.loss = d["loss"]


def test_bench_two_ranks_on_one_gpu_agree_with_one_rank():
    env = dict(os.environ, DGLL_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SHAPE
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "cpu_baseline" not in d
    one = getattr(test_bench_single_rank_contract, "loss", None)
    if one is not None:
        assert abs(d["loss"] - one) < 2e-2 * abs(one)
