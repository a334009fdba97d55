#!/usr/bin/env python3
"""Headline benchmark: aggregated edges/s (+ epoch time) of full-graph 3-layer GraphSAGE training on an
ogbn-products-shaped synthetic graph, hidden = 256, bf16 storage / fp32 accumulation (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W        N = 1: in-process.  N > 1 without WORLD_SIZE in the environment:
                                                         the parent starts N ranks itself (before touching the GPU) through
                                                         torch.distributed.run, relays rank 0's JSON line and exits with
                                                         the children's status -- the reference's `mp.spawn(run, nprocs=N)`
                                                         (dgll/GPU Accelerator/MQGCN.py:161-163).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W        (one rank per GPU, RCCL; WORLD_SIZE must equal --gpus)

A step = one pass of the hot path over the whole graph: forward through three sageConv layers (mean neighbour
aggregation = CSR SpMM in libdgll_hip.so, then the dense transforms), cross-entropy over all nodes, backward
(SpMM on the transposed CSR for every layer whose input needs a gradient), Adam update.  One step is one epoch.
"aggregated edges" counts nnz once per SpMM-type launch (3 forward + 2 backward per step).

Workload: exactly ogbn-products' size (2 449 029 nodes, 61 859 140 undirected = 123 718 280 directed edges), 64 planted
communities holding 90 % of the edges, node ids RANDOMLY PERMUTED (what a raw dataset looks like).  The engine's own
one-off locality pass (CSRGraph.reorder, --reorder) relabels the nodes before training, as the METIS relabelling of
BASELINE config 3 does; --reorder none measures the raw order, --no-permute the generator's community-sorted order.

Rank 0 prints ONE JSON line; see DESIGN.md section 6 for the fields (`roofline`, `cpu_baseline`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=2_449_029, help="ogbn-products node count")
    ap.add_argument("--undirected-edges", type=int, default=61_859_140, help="ogbn-products undirected edge count")
    ap.add_argument("--in-feats", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--feat-align", type=int, default=64,
                    help="input feature rows are padded to a multiple of this many elements (64 bf16 = one 128-byte line)")
    ap.add_argument("--locality", type=float, default=0.9,
                    help="fraction of edges inside one of 64 planted communities; 0 = structure-free RMAT")
    ap.add_argument("--no-permute", action="store_true",
                    help="keep the generator's community-sorted node ids (the default permutes them randomly)")
    ap.add_argument("--reorder", choices=["none", "lpa"], default="lpa",
                    help="the engine's one-off locality relabelling before training (CSRGraph.reorder)")
    ap.add_argument("--inexact-edges", action="store_true", help="one draw of edges, duplicates coalesced (nnz a few % low)")
    ap.add_argument("--racom-async", action="store_true",
                    help="multi-rank: RaCoM asynchronous gradient sharing (the bucket all-reduce of step t overlaps step t+1; "
                         "applied one step late, drained every sync period) instead of the synchronous form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-graphs", "--no-worst-case", dest="no_extra", action="store_true",
                    help="skip the extra SpMM measurements on the structure-free and raw-order graphs")
    ap.add_argument("--calibrate", action="store_true",
                    help="launch the known-byte identity gather 3 times before the timed steps (PMC calibration rows)")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--dataset", default=None,
                    help="run a REAL dataset instead of the synthetic graph when its files are present: a .npz edge-list dump, a "
                         "directory with reddit_data.npz / reddit_graph.npz, or an OGB raw directory (dgll_amd/data/formats.py)")
    return ap.parse_args(argv)


def alg_bytes(nnz, n_rows, feat, x_bytes, y_bytes, weighted):
    """every edge is charged one full feature-row read."""
    return nnz * (feat * x_bytes + 4 + (4 if weighted else 0)) + n_rows * (feat * y_bytes + 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """--gpus N > 1 and no WORLD_SIZE: start the N ranks as children BEFORE anything here touches the GPU (importing torch and
    counting devices does not initialise it; a process that has initialised the GPU must never be replaced or forked)."""
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    if n_dev < args.gpus and env.get("DGLL_BENCH_BACKEND") != "gloo":
        print("bench.py: --gpus %d but only %d device(s) visible (set DGLL_BENCH_BACKEND=gloo to let ranks share a GPU for a "
              "functional check)" % (args.gpus, n_dev), file=sys.stderr)
        return 2
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if rc != 0:
        print("bench.py: a rank failed (torch.distributed.run exit code %d)" % rc, file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks finished without printing a result line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def cpu_baseline(graph, feat, sample_rows, seed):
    """BASELINE.md section 5 on the GPU box's host cores, on a bounded sample of the SAME tensors (the first `sample_rows`
    rows of the adjacency the GPU ran, the same feature width, fp32): the reference's own op -- torch.spmm on a COO tensor
    (dgll/nn/Convolution/gcnconv.py:31, restated in oracle/torch_ref.spmm_coo) -- plus torch.sparse.mm on CSR and the C
    oracle (oracle/oracle.c + OpenMP); each 1 warm-up + median of 3."""
    import numpy as np

    from oracle import cref, torch_ref

    rows = min(sample_rows, graph.n_rows)
    rowptr_t = graph.rowptr[:rows + 1].cpu()
    nnz = int(rowptr_t[-1])
    col_t = graph.col[:nnz].cpu()
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((graph.n_cols, feat), dtype=np.float32)
    xt = torch.from_numpy(x)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    cref.set_num_threads(cores)
    deg = (rowptr_t[1:] - rowptr_t[:-1])
    row_t = torch.repeat_interleave(torch.arange(rows), deg)
    val_t = (1.0 / deg.clamp(min=1).to(torch.float32))[row_t]              # mean = D^-1 A, utils.py:171
    col64 = col_t.to(torch.int64)

    def median3(fn):
        fn()                                                              # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[1]

    t_coo = median3(lambda: torch_ref.spmm_coo(row_t, col64, val_t, xt, rows))
    csr = torch.sparse_csr_tensor(rowptr_t, col64, val_t, size=(rows, graph.n_cols))
    t_csr = median3(lambda: torch.sparse.mm(csr, xt))
    rp, cc = rowptr_t.numpy(), col_t.numpy()
    t_c = median3(lambda: cref.spmm_csr(rp, cc, None, x, reduce="mean"))
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    return {
        "value": nnz / t_coo, "unit": "edges/s", "cores": cores, "kind": "port",
        "sample": "mean-SpMM of the first %d rows (%d edges) of the benchmark's own adjacency against all %d feature rows, "
                  "F=%d fp32; value = the reference's op torch.spmm(adj_coo, X) (gcnconv.py:31) as restated in "
                  "oracle/torch_ref.spmm_coo; 1 warm-up + median of 3" % (rows, nnz, graph.n_cols, feat),
        "torch_sparse_mm_csr_edges_per_s": nnz / t_csr,
        "oracle_c_openmp_csr_edges_per_s": nnz / t_c,
        "algorithmic_GBps": alg_bytes(nnz, rows, feat, 4, 4, True) / t_coo / 1e9,
        "torch_version": torch.__version__, "cpu_model": model, "threads": cores,
        "seconds_per_run": {"torch_spmm_coo": t_coo, "torch_sparse_mm_csr": t_csr, "oracle_c_openmp": t_c},
    }


def workload_signature(args, nnz):
    return {"nodes": args.nodes, "nnz": nnz, "locality": args.locality, "permuted_ids": not args.no_permute,
            "reorder": args.reorder, "hidden": args.hidden, "dtype": args.dtype}


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args))
    world = int(env_world or "1")
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if world > 1:
        import torch.distributed as dist

        # "nccl" is RCCL on ROCm.  DGLL_BENCH_BACKEND=gloo lets several ranks share one GPU (functional check of
        # the multi-rank path on a 1-GPU box; never used for reported numbers).
        backend = os.environ.get("DGLL_BENCH_BACKEND", "nccl")
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import dgll_amd
    from dgll_amd import nn as dnn
    from dgll_amd import ops, synth

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esz = 2 if dtype == torch.bfloat16 else 4
    torch.manual_seed(args.seed)

    # ---- workload: the same seeded graph on every rank -------------------------------------------------
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + 1)
    if args.dataset:
        from dgll_amd.data import formats

        dg = formats.load_node_dataset(args.dataset, symmetrise=True) if not os.path.exists(os.path.join(args.dataset, "reddit_data.npz")) \
            else formats.load_node_dataset(args.dataset)
        full = dg.to_csr(dev)
        feats_all = dg.features.to(dev)
        labels_all = dg.labels.to(dev).long()
        args.nodes, args.in_feats, args.classes = full.n_rows, int(feats_all.shape[1]), int(labels_all.max()) + 1
        del dg
    else:
        full = synth.products_like_graph(dev, seed=args.seed, n=args.nodes, n_undirected=args.undirected_edges,
                                         locality=args.locality, exact=not args.inexact_edges, permute_ids=not args.no_permute)
        labels_all = torch.randint(0, args.classes, (full.n_rows,), generator=gen, device=dev)
        feats_all = torch.randn(full.n_rows, args.in_feats, generator=gen, device=dev)
    n, nnz = full.n_rows, full.nnz
    model = dnn.GraphSage(args.in_feats, [args.hidden, args.hidden, args.classes], None).to(dev)

    # ---- the engine's one-off locality pass (outside the timed steps, like a METIS relabelling) -----------
    reorder_s = 0.0
    bounds = None
    if args.reorder != "none":
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if world > 1:
            # partition + order in one pass: communities packed into `world` parts of equal edge count (the place METIS has in
            # the reference's pipeline), each part in locality order; deterministic, so every rank derives the same relabelling
            from dgll_amd import partition as dpart
            from dgll_amd import reorder as dreorder

            perm, bounds = dpart.partition_and_order(full, world, seed=args.seed)
            full = dreorder.relabel(full, perm)
        else:
            full, perm = full.reorder(method=args.reorder, seed=args.seed)     # new row i = old row perm[i]
        torch.cuda.synchronize()
        reorder_s = time.perf_counter() - t0
        labels_all = labels_all[perm]
        feats_all = feats_all[perm]

    racom = opt_wrap = None
    if world > 1:
        from dgll_amd import dist as ddist

        if bounds is None:
            bounds = [(n * r) // world for r in range(world + 1)]
        # every rank keeps ONLY its own row block (what it would load from its part file) and learns what its peers need
        # from one exchange of halo ids
        b0, b1 = bounds[rank], bounds[rank + 1]
        e0, e1 = int(full.rowptr[b0]), int(full.rowptr[b1])
        own_rowptr, own_col = full.rowptr[b0:b1 + 1].clone(), full.col[e0:e1].clone()
        del full
        torch.cuda.empty_cache()
        part = ddist.partition_rows(own_rowptr, own_col, None, bounds, rank)
        del own_rowptr, own_col
        engine = ddist.DistGraph(part, dev)
        x_local = ops.alloc_features(part.n_own, args.in_feats, dtype, dev, pad_to=args.feat_align)
        x_local.copy_(engine.permute_to_local(feats_all[part.own_begin:part.own_end]).to(dtype))
        labels = engine.permute_to_local(labels_all[part.own_begin:part.own_end])
        placed_input = engine.place_input_halo(x_local)     # input features of halo nodes live with the partition
        graph_for_cpu = None
    else:
        engine = None
        x_local = ops.alloc_features(n, args.in_feats, dtype, dev, pad_to=args.feat_align)
        x_local.copy_(feats_all.to(dtype))
        labels = labels_all
        graph_for_cpu = full
        full.plan()
        full.transpose()[0].plan()
        full.mean_scale_transposed()
    del feats_all
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    if world > 1:
        if args.racom_async:
            opt_wrap = ddist.RaCoMOptimizer(opt, model.parameters(), dev, staleness=1,
                                            sync_every=ddist.racom_sync_period(n, world))
        else:
            racom = ddist.RaCoM(model.parameters(), dev)
    # forward: 3 layers; backward: layers 2 and 3 (the input features need no gradient).  The last layer narrows
    # (256 -> 47), so it aggregates the 47-wide product X.W_n instead of the 256-wide input (mean is linear).
    spmm_launches_per_step = 3 + 2

    def step():
        opt.zero_grad(set_to_none=True)
        if engine is None:
            out = model.forward_graph(full, x_local)
        else:
            out = engine.sage_forward(model, x_local, placed_input)
        # cross-entropy summed over this rank's nodes / global node count (x world: RaCoM averages over ranks)
        loss = ops.cross_entropy(out, labels, reduction="sum") * (world / n)     # one kernel per direction
        loss.backward()
        if opt_wrap is not None:
            opt_wrap.step()
        else:
            if racom is not None:
                racom.all_reduce_and_wait()
            opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.calibrate and world == 1:
        ident = dgll_amd.CSRGraph.fixed_fanout(n, 1, dev)
        xc = ops.alloc_features(n, args.hidden, dtype, dev)
        xc.copy_(torch.randn(n, args.hidden, device=dev).to(dtype))
        for _ in range(3):
            ops.spmm_raw(ident, xc, reduce="sum")            # streaming read + write of n*hidden*esz bytes each: known bytes
        del ident, xc
    trace = []
    for _ in range(args.warmup):
        wl = step()
        if os.environ.get("DGLL_BENCH_TRACE_LOSS"):      # debugging aid: per-step global loss (costs a sync + all-reduce)
            g = wl.detach().double() / world
            if world > 1:
                torch.distributed.all_reduce(g)
            trace.append(float(g))
            if rank == 0:
                print("warm-up loss %.6f" % float(g), file=sys.stderr)
    barrier()
    with ops.LaunchTimer() as timer:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        barrier()
        elapsed = time.perf_counter() - t0
    if opt_wrap is not None:
        opt_wrap.flush()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    global_loss = loss.detach().double() / world      # this rank's share of the mean loss
    if world > 1:
        torch.distributed.all_reduce(global_loss)
    global_loss = float(global_loss)
    ms_per_step = elapsed / args.steps * 1e3
    value = spmm_launches_per_step * nnz * args.steps / elapsed

    if rank != 0:
        return
    # ---- roofline: every SpMM-type launch of the timed steps, HIP events on the launch stream (ops.LaunchTimer) ----
    launches = timer.summary()
    local_rows = n if engine is None else engine.part.n_own
    table = {}
    for tag, (cnt, avg_ms) in launches.items():
        if tag[0] != "spmm":
            continue
        _, feat, dt, weighted, tag_nnz = tag[:5]
        extra = tag[5] if len(tag) > 5 else ""
        xb = 2 if "bfloat16" in dt else 4
        b_alg = alg_bytes(tag_nnz, local_rows, feat, xb, xb, weighted)
        name = "spmm F=%d %s %s%s nnz=%d" % (feat, dt.replace("torch.", ""), "weighted" if weighted else "unweighted",
                                             (" " + extra) if extra else "", tag_nnz)
        table[name] = {"count": cnt, "avg_ms": avg_ms, "nnz": tag_nnz, "feat": feat, "weighted": bool(weighted),
                       "epilogue": extra, "G_edges_per_s": tag_nnz / (avg_ms * 1e-3) / 1e9,
                       "algorithmic_bytes": b_alg, "algorithmic_GBps": b_alg / (avg_ms * 1e-3) / 1e9,
                       "frac_algorithmic": b_alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    # the dense kernels next to the aggregation (all hand-written MFMA: no library GEMM in the step), same HIP-event timing
    dense_table = {}
    for tag, (cnt, avg_ms) in launches.items():
        if tag[0] not in ("transform", "transform_dual", "grad_weight"):
            continue
        kind, m, k1, k2, n_out, extra = tag
        b_alg = m * (k1 + k2 + n_out) * 2 + (m * n_out * 2 if ("gate" in extra or "addend" in extra) else 0)
        name = "%s M=%d K=%d%s N=%d%s" % (kind, m, k1, ("+%d" % k2) if k2 else "", n_out, (" " + extra) if extra else "")
        dense_table[name] = {"count": cnt, "avg_ms": avg_ms, "algorithmic_bytes": b_alg,
                             "algorithmic_GBps": b_alg / (avg_ms * 1e-3) / 1e9,
                             "frac_of_hbm_peak": b_alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "frac_of_streaming_ceiling_5500": b_alg / (avg_ms * 1e-3) / 1e9 / 5500.0}
    # headline = the LONGEST hidden-width SpMM launch of the step (forward mean aggregation or the weighted, gated,
    # accumulating transposed launch of the backward pass, whichever takes longer)
    roofline = None
    wide = {k: v for k, v in table.items() if v["feat"] == args.hidden}
    if wide:
        dom_name = max(wide, key=lambda k: wide[k]["avg_ms"])
        dom = wide[dom_name]
        sig = workload_signature(args, nnz)
        traffic = load_traffic(sig, dom) if world == 1 else None
        achieved = dom["algorithmic_GBps"]
        frac_alg = achieved / HBM_PEAK_GBPS
        n_cols_touched = n if engine is None else engine.part.n_own + engine.part.n_halo
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": frac_alg, "frac_algorithmic": frac_alg, "traffic": traffic,
                    "kernel": "spmm_csr_kernel bf16 %s%s, F=%d (the longest SpMM-type launch of the step)" % (
                        "weighted" if dom["weighted"] else "unweighted", (" " + dom["epilogue"]) if dom["epilogue"] else "",
                        args.hidden),
                    "launches_timed": dom["count"], "avg_launch_ms": dom["avg_ms"],
                    "algorithmic_bytes_per_launch": dom["algorithmic_bytes"],
                    # SURVEY 8(d)'s compulsory lower bound: every index once, every feature / output row once
                    "compulsory_bytes_per_launch": dom["nnz"] * (8 if dom["weighted"] else 4) + n_cols_touched * args.hidden * esz
                                                   + local_rows * (args.hidden * esz + 8),
                    "edges_per_s_this_kernel": dom["nnz"] / (dom["avg_ms"] * 1e-3)}
        if traffic is not None:
            hbm = traffic / (dom["avg_ms"] * 1e-3) / 1e9
            roofline["achieved_hbm_counters"] = hbm
            roofline["frac_hbm_counters"] = hbm / HBM_PEAK_GBPS
        if frac_alg > 1.0:
            # the section-8(d) formula charges every edge a full row read; a value above the peak means part of those reads
            # were served by L2 / Infinity Cache.  `frac` then carries the HBM-side counter traffic (profiles/traffic.json,
            # rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of this very launch) divided by the live launch time -- or null.
            roofline["frac"] = roofline.get("frac_hbm_counters")
            roofline["note"] = ("algorithmic bytes / time exceeds the HBM peak (cache-served re-reads): frac = HBM-counter "
                                "traffic / launch time / peak; frac_algorithmic keeps the formula's value")
    result = {
        "metric": "aggregated edges/sec + epoch time, 3-layer GraphSAGE ogbn-products, 1/2/4/8 GPU",
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": ("real: " + args.dataset) if args.dataset else "synthetic",
        "config": {"workload": "full-graph 3-layer GraphSAGE (mean aggr, %d-%d-%d-%d) training step on an ogbn-products-sized "
                               "synthetic graph: %d nodes, nnz %d, 64 planted communities (locality %.2f), node ids %s, "
                               "engine reorder: %s" % (args.in_feats, args.hidden, args.hidden, args.classes, n, nnz,
                                                       args.locality, "community-sorted" if args.no_permute else "randomly permuted",
                                                       args.reorder),
                   "nodes": n, "nnz": nnz, "hidden": args.hidden, "locality": args.locality,
                   "permuted_ids": not args.no_permute, "reorder": args.reorder, "reorder_seconds_one_off": reorder_s,
                   "parallelism": "1-D row partition x%d" % world, "spmm_launches_per_step": spmm_launches_per_step,
                   "ranks": world, "backend": backend,
                   "gradient_sharing": None if world == 1 else ("RaCoM async (staleness 1)" if opt_wrap is not None else "RaCoM sync")},
        "epoch_time_s": ms_per_step / 1e3, "loss": global_loss,
        "roofline": roofline,
        "spmm_launch_table": table,
        "dense_launch_table": dense_table,   # bytes = 2 (K1 + K2 + N) per row (+ 2 N for a gate / addend operand); the 5.5 TB/s
    }                                        # "streaming ceiling" is what a trivial 2R:1W kernel reaches (tools/probes/rw_mix.hip)
    if trace:
        result["warmup_loss_trace"] = trace
    if world == 1 and not args.no_extra and not args.dataset:
        del model, opt
        result["roofline_no_locality"] = extra_roofline(args, dev, dtype, esz, locality=0.0, permute=False, reorder=args.reorder,
                                                        note="structure-free RMAT (locality 0: no communities), engine reorder '%s' as in "
                                                             "the headline run" % args.reorder)
        if args.reorder != "none":
            result["roofline_no_locality_raw_order"] = extra_roofline(
                args, dev, dtype, esz, locality=0.0, permute=False, reorder="none",
                note="structure-free RMAT in the generator's own id order (hubs at low ids), no reordering: round 1's figure")
            result["roofline_raw_order"] = extra_roofline(args, dev, dtype, esz, locality=args.locality, permute=not args.no_permute,
                                                          reorder="none", note="the headline graph WITHOUT the engine's reordering pass")
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(graph_for_cpu, args.hidden, args.cpu_sample_rows, args.seed)
    print(json.dumps(result))


def extra_roofline(args, dev, dtype, esz, locality, permute, reorder, note):
    """The forward hidden-width mean-SpMM on another variant of the graph (same size, same kernel), 10 back-to-back launches."""
    from dgll_amd import ops, synth

    g = synth.products_like_graph(dev, seed=args.seed, n=args.nodes, n_undirected=args.undirected_edges, locality=locality,
                                  exact=not args.inexact_edges, permute_ids=permute)
    if reorder != "none":
        g, _ = g.reorder(method=reorder, seed=args.seed)
    g.plan()
    x = ops.alloc_features(g.n_cols, args.hidden, dtype, dev)
    x.copy_(torch.randn(g.n_cols, args.hidden, device=dev).to(dtype))
    for _ in range(2):
        ops.spmm_raw(g, x, reduce="mean")
    torch.cuda.synchronize()
    reps = 10
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.spmm_raw(g, x, reduce="mean")
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    b_alg = alg_bytes(g.nnz, g.n_rows, args.hidden, esz, esz, weighted=False)
    achieved = b_alg / (ms * 1e-3) / 1e9
    sig = dict(workload_signature(args, g.nnz), locality=locality, permuted_ids=permute, reorder=reorder)
    traffic = load_traffic(sig, {"feat": args.hidden, "weighted": False, "epilogue": ""})
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "frac_algorithmic": achieved / HBM_PEAK_GBPS, "traffic": traffic,
           "nnz": g.nnz, "avg_launch_ms": ms, "edges_per_s_this_kernel": g.nnz / (ms * 1e-3),
           "note": "forward mean-SpMM F=%d, %s; %d back-to-back launches (timed with the long-row finalize)" % (args.hidden, note, reps)}
    if traffic is not None:
        out["frac_hbm_counters"] = traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        if out["frac"] > 1.0:
            out["frac"] = out["frac_hbm_counters"]
    return out


def load_traffic(sig, launch):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/traffic.json: entries keyed by
    the workload signature and the launch kind), or None when no pass was collected for this exact workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entries = json.load(f).get("entries", [])
    except (OSError, ValueError):
        return None
    for e in entries:
        w = e.get("workload", {})
        if all(w.get(k) == v for k, v in sig.items()) and e.get("feat") == launch["feat"] and \
                bool(e.get("weighted")) == bool(launch["weighted"]) and e.get("epilogue", "") == launch.get("epilogue", ""):
            return e.get("hbm_bytes_per_launch")
    return None


if __name__ == "__main__":
    main()
