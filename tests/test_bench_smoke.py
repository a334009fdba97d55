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
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line"] + SHAPE, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "edges/s" and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["traffic"] is None       # counters exist for the default shape only
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["frac"] == min(1.0, d["roofline"]["frac_algorithmic"])      # no counter entry for this shape
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    # the reference's COO op AND the CSR variant, warm-up + median of 3, torch version and cores stated
    assert set(("torch_sparse_mm_csr_edges_per_s", "oracle_c_openmp_csr_edges_per_s", "torch_version", "threads")) <= set(d["cpu_baseline"])
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["nnz"] == 2 * 200000                      # exactly the requested number of distinct undirected edges
    assert d["config"]["reorder"] == "lpa" and d["config"]["permuted_ids"] is True
    # every SpMM-type launch of the step is listed; the headline is the longest hidden-width one
    wide = [v for v in d["spmm_launch_table"].values() if v["feat"] == 64]
    assert len(wide) >= 2 and abs(max(v["avg_ms"] for v in wide) - d["roofline"]["avg_launch_ms"]) < 1e-9
    assert "roofline_no_locality" in d and "roofline_raw_order" in d and "roofline_no_locality_raw_order" in d
    test_bench_single_rank_contract.loss = d["loss"]


def test_bench_default_line_is_compact_and_parses(tmp_path):
    """What the driver reads: the LAST stdout line, without --full-line -- at most 6000 characters (round 5's 25 KB line was not parsed),
    valid JSON, the contract fields, numeric `roofline` / `cpu_baseline`, `other_workloads` as <= 10 numbers per config; the full record
    goes to bench_full.json (here: DGLL_BENCH_FULL_RECORD) and to stderr."""
    other = {"rmat27": ["--scale", "16", "--steps", "2", "--warmup", "1", "--cpu-sample-rows", "2000"]}
    full_path = str(tmp_path / "full.json")
    env = dict(os.environ, DGLL_BENCH_OTHER_ARGS=json.dumps(other), DGLL_BENCH_FULL_RECORD=full_path)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--other-workloads", "on"] + SHAPE, capture_output=True, text=True,
                         timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    line = lines[-1]
    assert len(line) < 6000 and sum(ln.startswith("{") for ln in lines) == 1
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["nnz"] == 400000
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0.0 < r["frac"] <= 1.0 and r["achieved"] > 0
    assert r["frac_kind"] in ("counter", "algorithmic_capped") and (r["traffic"] is None) == (r["frac_kind"] == "algorithmic_capped")
    assert all(not isinstance(v, str) or len(v) <= 100 for v in r.values()) and "frac_definition" not in r
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "edges/s" and len(cb["sample"]) <= 110
    # value and ms_per_step agree with each other: 5 SpMM-type launches of nnz edges per step
    assert abs(d["value"] - 5 * d["config"]["nnz"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    ow = d["other_workloads"]["rmat27"]
    assert len(ow) <= 10 and all(isinstance(v, (int, float)) for v in ow.values()) and ow["ms_per_step"] > 0 and 0 < ow["roofline_frac"] <= 1
    full = json.load(open(full_path))
    assert d["full_record"] == "full.json" and "spmm_launch_table" in full and "frac_definition" in full["roofline"]
    assert full["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-8) and "command" in full["other_workloads"]["rmat27"]
    assert "bench.py full record" in res.stderr


def _check_two_ranks(res):
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "cpu_baseline" not in d
    assert d["config"]["ranks"] == 2 and d["config"]["backend"] == "gloo"
    one = getattr(test_bench_single_rank_contract, "loss", None)
    if one is not None:
        assert abs(d["loss"] - one) < 2e-2 * abs(one)
    return d


def test_bench_two_ranks_on_one_gpu_agree_with_one_rank():
    env = dict(os.environ, DGLL_BENCH_BACKEND="gloo", DGLL_HALO_MODE="recompute")   # (auto adds warm-up steps: the loss is compared across rank counts)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--full-line", "--gpus", "2"] + SHAPE
    _check_two_ranks(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env))


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun around it (the shape of the driver's N = 1 command): the parent spawns the
    two ranks, relays rank 0's line and exits with their status (MQGCN.py:161-163 `mp.spawn(run, nprocs=num_gpus)`)."""
    env = dict(os.environ, DGLL_BENCH_BACKEND="gloo", DGLL_HALO_MODE="recompute")   # (auto adds warm-up steps: the loss is compared across rank counts)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--gpus", "2"] + SHAPE, capture_output=True, text=True,
                         timeout=900, env=env)
    _check_two_ranks(res)
    # asynchronous RaCoM (gradients applied one step late, drained every sync period): same contract, finite loss
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--gpus", "2", "--racom-async"] + SHAPE,
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == 2 and "async" in d["config"]["gradient_sharing"] and d["loss"] == d["loss"]


def test_bench_lets_the_live_ranks_choose_the_halo_mode():
    """bench.py at N > 1 defaults to DGLL_HALO_MODE=auto: the first two layers' two forms (recompute on halo rows / exchange of
    hidden-width rows) are both timed during warm-up, max over ranks, and the faster one runs the timed steps; the line says which."""
    env = dict(os.environ, DGLL_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DGLL_HALO_MODE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--gpus", "2"] + SHAPE, capture_output=True, text=True,
                         timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    hm = _last_json(res.stdout)["config"]["halo_mode"]
    assert hm["requested"] == "auto" and hm["mode"] in ("recompute", "exchange") and set(hm["timings_ms"]) == {"recompute", "exchange"}
    assert (hm["timings_ms"]["recompute"] <= hm["timings_ms"]["exchange"]) == (hm["mode"] == "recompute")


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--gpus", "2"] + SHAPE, capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode == 2 and "WORLD_SIZE" in res.stderr


def test_bench_eight_ranks_on_one_gpu():
    """The world-size-8 shape of BASELINE config 3 end to end (8 ranks share the one GPU over gloo: a functional run of the
    partitioner, the halo exchange with seven peers, RaCoM and the JSON contract -- never a reported number)."""
    env = dict(os.environ, DGLL_BENCH_BACKEND="gloo", DGLL_HALO_MODE="recompute")   # (auto adds warm-up steps: the loss is compared across rank counts)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--gpus", "8"] + SHAPE, capture_output=True, text=True,
                         timeout=1500, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == 8 and d["config"]["ranks"] == 8 and d["scaling"] == "strong" and "cpu_baseline" not in d
    one = getattr(test_bench_single_rank_contract, "loss", None)
    if one is not None:
        assert abs(d["loss"] - one) < 3e-2 * abs(one)


def test_bench_gat_workload_one_and_two_ranks():
    """bench.py --workload gat (BASELINE config 4): the JSON contract with the GAT roofline formula, per-pass launch rows and
    the GAT cpu_baseline; two ranks (gloo, one GPU) reach the same loss through the partitioned edge-softmax."""
    shape = ["--workload", "gat", "--nodes", "20000", "--undirected-edges", "200000", "--hidden", "64", "--heads", "8", "--classes", "10",
             "--in-feats", "40", "--steps", "2", "--warmup", "1", "--cpu-sample-rows", "2000"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line"] + shape, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["config"]["workload_id"] == "gat" and d["config"]["gather_passes_per_step"] == 6 and d["roofline"]["bound"] == "hbm"
    passes = {v["pass"] for v in d["spmm_launch_table"].values() if "pass" in v}
    assert passes == {"fwd", "bwd_rows", "bwd_cols"}
    assert "gat2_kernel" in d["roofline"]["kernel_fragment"] and d["roofline"]["frac_definition"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and "gat_pass_over_spmm" in d
    env = dict(os.environ, DGLL_BENCH_BACKEND="gloo", DGLL_HALO_MODE="recompute")   # (auto adds warm-up steps: the loss is compared across rank counts)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--gpus", "2"] + shape, capture_output=True, text=True,
                          timeout=900, env=env)
    assert res2.returncode == 0, res2.stderr[-3000:]
    d2 = _last_json(res2.stdout)
    assert d2["n_gpus"] == 2 and abs(d2["loss"] - d["loss"]) < 3e-2 * abs(d["loss"])


@pytest.mark.parametrize("fused", [True, False])
def test_bench_minibatch_and_rmat_workloads(fused):
    """bench.py --workload minibatch (BASELINE config 2: sampler thread -> loader thread with the feature cache -> training) and
    --workload rmat27 (config 5's shape at a small scale): the JSON contract, the steady-state window (at least 16 + 192 batches
    whatever --steps says), the fields the judge reads, and that reducing the outermost hop straight out of the cache changes nothing
    but the speed."""
    shape = ["--workload", "minibatch", "--mb-nodes", "20000", "--mb-undirected-edges", "400000", "--mb-feats", "50", "--mb-classes", "7",
             "--mb-batch", "64", "--mb-fanouts", "5,3,3", "--hidden", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if not fused:
        shape.append("--mb-no-fused-last-hop")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line"] + shape, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["config"]["workload_id"] == "minibatch" and d["steps"] == 192 and d["warmup"] == 16
    assert d["config"]["outermost_hop"].startswith("reduced" if fused else "fetched")
    for key in ("gpu_side_ms_per_batch", "gpu_side_ms_per_batch_p50", "gpu_side_ms_per_batch_p95", "host_sampler_ms_per_batch",
                "host_sampler_threads", "sampler_mode", "consumer_host_ms_per_batch", "loader_host_ms_per_batch", "cache_miss_rate",
                "batches_per_s", "roofline"):
        assert key in d, key
    assert d["host_sampler_threads"] >= 1 and "per-batch seeds" in d["sampler_mode"]
    assert 0.0 < d["cache_miss_rate"] < 1.0 and d["loss"] == d["loss"]
    prev = getattr(test_bench_minibatch_and_rmat_workloads, "loss", None)
    if prev is not None:
        assert abs(prev - d["loss"]) < 2e-2 * abs(prev)                 # same batches (seeded sampler), same model
    test_bench_minibatch_and_rmat_workloads.loss = d["loss"]
    if fused:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--workload", "rmat27", "--scale", "16", "--steps", "2",
                              "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        r = _last_json(res.stdout)
        assert r["config"]["workload_id"] == "rmat27" and r["roofline"]["bound"] == "hbm" and r["value"] > 0


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rmat27_row_blocks_on_several_ranks(world):
    """BASELINE config 5 at N > 1 (`bench.py --workload rmat27 --gpus N`; here scale 16, the ranks share the one GPU over gloo: a
    functional run, never a reported number): cost-balanced contiguous row blocks, X replicated, value = sum of the ranks'
    nonzeros / slowest rank, per-rank roofline, the separately timed output all-gather."""
    env = dict(os.environ, DGLL_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    one = _last_json(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--workload", "rmat27", "--scale", "16", "--steps", "2",
                                     "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600).stdout)
    assert one["config"]["layer_order"] == "transform-first" and one["aggregate_first_ms_per_step"] > 0 and len(one["per_rank"]) == 1
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--workload", "rmat27", "--scale", "16", "--steps", "2",
                          "--warmup", "1", "--gpus", str(world)], capture_output=True, text=True, timeout=1200, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["layer_order"] == "aggregate-first"
    assert d["config"]["nnz"] == one["config"]["nnz"] and d["config"]["nodes"] == one["config"]["nodes"]
    ranks = d["per_rank"]
    assert [r["rank"] for r in ranks] == list(range(world))
    assert sum(r["nnz"] for r in ranks) == d["config"]["nnz"] and sum(r["rows"] for r in ranks) == d["config"]["nodes"]
    assert [r["row_range"][0] for r in ranks] == d["config"]["row_bounds"][:-1]
    cost = [d["config"]["edge_cost"] * r["nnz"] + d["config"]["row_cost"] * r["rows"] for r in ranks]
    assert max(cost) < 1.25 * (sum(cost) / world)                                   # (one hub row is ~1 % of a share at this scale)
    assert len(d["config"]["rebalance"]["measured_ms"]) == world
    assert all(r["spmm_avg_ms"] > 0 and r["spmm_frac_algorithmic"] > 0 for r in ranks)
    assert abs(d["value"] - d["config"]["nnz"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert d["output_allgather"]["ms"] > 0 and "cpu_baseline" not in d
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])


def test_bench_minibatch_single_stream_mode_still_runs():
    """--mb-sampler-threads 0: ONE sequential sampler stream on the interpreter's generator (the default-compatible mode of rounds
    1-3), host-side translation of the outermost hop."""
    shape = ["--workload", "minibatch", "--mb-nodes", "20000", "--mb-undirected-edges", "400000", "--mb-feats", "50", "--mb-classes", "7",
             "--mb-batch", "64", "--mb-fanouts", "5,3,3", "--hidden", "64", "--no-cpu-baseline", "--mb-sampler-threads", "0", "--mb-host-translate"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line"] + shape, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["host_sampler_threads"] == 0 and "sequential" in d["sampler_mode"] and d["outermost_hop_translation"] == "host"
    assert d["loss"] == d["loss"] and d["steps"] == 192


def test_default_line_carries_the_other_baseline_configs():
    """`bench.py` (what the driver runs) appends `other_workloads` = compact records of configs 4, 2 and 5, each measured by a child
    run; here at small shapes (DGLL_BENCH_OTHER_ARGS) with --other-workloads on."""
    other = {
        "gat": ["--nodes", "20000", "--undirected-edges", "200000", "--hidden", "64", "--heads", "8", "--classes", "10", "--in-feats", "40",
                "--steps", "2", "--warmup", "1", "--cpu-sample-rows", "2000", "--no-extra-graphs"],
        "minibatch": ["--mb-nodes", "20000", "--mb-undirected-edges", "400000", "--mb-feats", "50", "--mb-classes", "7", "--mb-batch", "64",
                      "--mb-fanouts", "5,3,3", "--hidden", "64"],
        "rmat27": ["--scale", "16", "--steps", "2", "--warmup", "1", "--cpu-sample-rows", "2000"],
    }
    env = dict(os.environ, DGLL_BENCH_OTHER_ARGS=json.dumps(other))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--other-workloads", "on", "--no-extra-graphs"] + SHAPE,
                         capture_output=True, text=True, timeout=1500, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    ow = d["other_workloads"]
    assert set(ow) == {"gat", "minibatch", "rmat27"}
    for name, rec in ow.items():
        assert "error" not in rec, (name, rec)
        assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["wall_seconds"] > 0 and rec["command"].startswith("python bench.py --workload " + name)
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rec["roofline"]) and 0.0 < rec["roofline"]["frac"] <= 1.0 and rec["roofline"]["frac_algorithmic"] == rec["roofline"]["achieved"] / rec["roofline"]["peak"]
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"]) and rec["cpu_baseline"]["kind"] == "port"
    assert "gat_pass_over_spmm" not in ow["gat"] and ow["minibatch"]["batches_per_s"] > 0       # --no-extra-graphs: no SpMM comparison leg
    # the small-shape headline alone does not start them
    assert "other_workloads" not in _last_json(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--no-extra-graphs"] + SHAPE,
                                                              capture_output=True, text=True, timeout=600).stdout)


def test_bench_runs_a_dataset_directory_in_the_ogb_raw_layout(tmp_path):
    """`bench.py --dataset DIR` on files in the OGB raw layout (edge.csv / node-feat.csv / node-label.csv: what ogbn-products is
    before processing; the real dataset is not available offline, so the files hold a small synthetic graph): node count, feature
    width and class count come from the files, `data` says so, the step runs through the same engine path."""
    import numpy as np

    rng = np.random.default_rng(0)
    n, f, c, m = 3000, 20, 6, 24000
    raw = tmp_path / "products" / "raw"
    raw.mkdir(parents=True)
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = src != dst
    np.savetxt(raw / "edge.csv", np.stack([src[keep], dst[keep]], 1), fmt="%d", delimiter=",")
    np.savetxt(raw / "node-feat.csv", rng.standard_normal((n, f)).astype(np.float32), fmt="%.5f", delimiter=",")
    np.savetxt(raw / "node-label.csv", rng.integers(0, c, (n, 1)), fmt="%d", delimiter=",")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-line", "--dataset", str(tmp_path / "products"), "--hidden", "64",
                          "--steps", "2", "--warmup", "1", "--cpu-sample-rows", "500", "--no-extra-graphs"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _last_json(res.stdout)
    assert d["data"].startswith("real: ") and d["config"]["nodes"] == n and d["loss"] == d["loss"]
    assert d["config"]["nnz"] > m and "other_workloads" not in d and d["cpu_baseline"]["value"] > 0      # symmetrised, duplicates merged
