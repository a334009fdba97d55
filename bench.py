#!/usr/bin/env python3
"""Benchmark of the aggregation hot path.  The default run is BASELINE.json's headline: aggregated edges/s (+ epoch time)
of full-graph 3-layer GraphSAGE training on an ogbn-products-sized synthetic graph, hidden = 256, bf16 storage / fp32
accumulation.

    python bench.py --gpus N --steps K --warmup W        N = 1: in-process.  N > 1 without WORLD_SIZE in the environment:
                                                         the parent starts N ranks itself (before touching the GPU) through
                                                         torch.distributed.run, relays rank 0's JSON line and exits with
                                                         the children's status -- the reference's `mp.spawn(run, nprocs=N)`
                                                         (dgll/GPU Accelerator/MQGCN.py:161-163).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W        (one rank per GPU, RCCL; WORLD_SIZE must equal --gpus)

--workload picks the BASELINE.json config that is measured (one JSON line each, same fields):
    sage       (default) config 3: full-graph 3-layer GraphSAGE, products size.  A step = forward through three sageConv
               layers (mean aggregation = CSR SpMM in libdgll_hip.so, then the MFMA transforms), cross-entropy over all
               nodes, backward (SpMM on the transposed CSR), Adam.  One step is one epoch.  5 SpMM-type launches per step.
    gat        config 4: full-graph 2-layer SpGAT, 8 heads x 32 -> 47 classes, the same graph + self-loops.  6 gather passes
               per step (forward, backward over rows, backward over the transposed rows; two layers).
    minibatch  config 2: Reddit-shaped graph, sampled 3-layer GraphSAGE (fan-out 25-10-10, batch 1024, hidden 256 bf16)
               through the bit-exact host sampler, the hot-node feature cache (PARTIAL: pinned-host leg carries traffic) and
               the mini-batch queue.  A step = one batch.
    rmat27     config 5: RMAT-27 (134 M nodes, > 2^31 nonzeros), F = 128 bf16: a step = one mean-SpMM pass + the dense X.W of
               the same layer (gcnconv.py:30-31); --gpus N: cost-balanced contiguous row blocks, X replicated, no exchange in
               the step (dgll_amd.dist.RowBlockShard).  --scale S for a smaller RMAT.

Workload (sage / gat): exactly ogbn-products' size (2 449 029 nodes, 61 859 140 undirected = 123 718 280 directed edges), 64
planted communities holding 90 % of the edges, node ids RANDOMLY PERMUTED (what a raw dataset looks like).  The engine's own
one-off locality pass (CSRGraph.reorder, --reorder) relabels the nodes before training, as the METIS relabelling of BASELINE
config 3 does; --reorder none measures the raw order, --no-permute the generator's community-sorted order.

Rank 0 prints ONE JSON line; DESIGN.md section 7 explains every field.  `roofline.achieved` is the section-8(d) algorithmic rate
(every edge charged one full feature row) of the dominant launch, timed live with HIP events on the launch stream.
`roofline.frac` is a fraction (<= 1) of the 8 TB/s peak, defined in `roofline.frac_definition`: counter traffic of that very
launch kind (profiles/traffic.json -- rocprofv3 FETCH_SIZE / WRITE_SIZE over this program, corrected by the ratios calibrated in
the same pass, used only when the entry was collected with THIS build of libdgll_hip.so; Infinity-Cache hits included: an upper
bound on HBM utilisation) / the live launch time / peak; without a matching entry min(1, frac_algorithmic).  `frac_algorithmic`
(may exceed 1: caches serve re-reads), `traffic_over_compulsory` and `frac_conservative` sit next to it.
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
FP32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32: 64 flops / cycle / SIMD x 4 SIMDs x 256 CUs x 2.4 GHz
LINE_LIMIT = 6000        # characters of the stdout line (the driver keeps an 8 000-character tail and parses the last line)
FULL_RECORD = os.path.join(ROOT, "bench_full.json")
METRIC = "aggregated edges/sec + epoch time, 3-layer GraphSAGE ogbn-products, 1/2/4/8 GPU"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["sage", "gat", "minibatch", "rmat27"], default="sage")
    ap.add_argument("--nodes", type=int, default=2_449_029, help="ogbn-products node count")
    ap.add_argument("--undirected-edges", type=int, default=61_859_140, help="ogbn-products undirected edge count")
    ap.add_argument("--in-feats", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--heads", type=int, default=8, help="gat: attention heads of the hidden layer (hidden / heads columns each)")
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
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the one-launch flat Adam (dgll_amd/optim.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-graphs", "--no-worst-case", dest="no_extra", action="store_true",
                    help="skip the extra SpMM measurements on the structure-free and raw-order graphs")
    ap.add_argument("--other-workloads", choices=["auto", "on", "off"], default="auto",
                    help="append compact records of the other BASELINE configs (gat = config 4, minibatch = config 2, rmat27 = config 5 on "
                         "one GPU) to the default line as `other_workloads`, each measured by a child `bench.py --workload X` run after the "
                         "headline; auto = on for the default single-GPU products-sized sage run")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL record (definitions, launch tables, per-rank diagnostics, the complete child records) as the stdout "
                         "line, as rounds 1-5 did; default: a compact line of at most %d characters -- the full record goes to "
                         "bench_full.json next to bench.py and to stderr" % LINE_LIMIT)
    ap.add_argument("--calibrate", action="store_true",
                    help="launch the known-byte identity gather 3 times before the timed steps (PMC calibration rows)")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--dataset", default=None,
                    help="run a REAL dataset instead of the synthetic graph when its files are present: a .npz edge-list dump, a "
                         "directory with reddit_data.npz / reddit_graph.npz, or an OGB raw directory (dgll_amd/data/formats.py)")
    # minibatch (config 2) / rmat27 (config 5)
    ap.add_argument("--mb-nodes", type=int, default=232_965)
    ap.add_argument("--mb-undirected-edges", type=int, default=57_300_000, help="Reddit: 114.6 M directed edges")
    ap.add_argument("--mb-feats", type=int, default=602)
    ap.add_argument("--mb-classes", type=int, default=41)
    ap.add_argument("--mb-batch", type=int, default=1024)
    ap.add_argument("--mb-fanouts", default="25,10,10")
    ap.add_argument("--mb-cache-frac", type=float, default=0.5,
                    help="fraction of the nodes whose features sit in the HBM cache; -1 = the reference's capacity rule (storage.py:64-98: "
                         "every row that fits in free device memory)")
    ap.add_argument("--mb-no-fused-last-hop", action="store_true",
                    help="minibatch: fetch the outermost hop's rows and reduce them in the model instead of reducing them straight out of the cache")
    ap.add_argument("--mb-sampler-threads", type=int, default=-1,
                    help="minibatch: K native sampler threads, each drawing whole batches under per-batch seeds (bit-equal to the reference "
                         "loop under random.seed(batch_seed(seed, epoch, b))); 0 = ONE sequential stream on the interpreter's generator "
                         "(rounds 1-3); -1 = min(8, host threads / 4)")
    ap.add_argument("--mb-loader-blocks-per-cu", type=int, default=4,
                    help="minibatch: cap of the feature-loading kernels' grids in workgroups per CU (0 = the kernels' own 16 / 32): they "
                         "run on the loading stream next to the training kernels")
    ap.add_argument("--gat-unfused-loss", action="store_true",
                    help="gat: F.nll_loss on the model's log_softmax output (torch kernels) instead of dgll_amd.ops.cross_entropy on its activations")
    ap.add_argument("--mb-hip-graph", choices=["auto", "on", "off"], default="auto",
                    help="minibatch: the consumer's forward + loss + backward as ONE HIP graph on padded static block shapes "
                         "(dgll_amd.graphs.GraphedSampledStep), replayed per batch")
    ap.add_argument("--mb-copy-inputs", action="store_true",
                    help="minibatch: the captured step copies every batch (260 MB) into ONE static input set (rounds 3-4) instead of the "
                         "loading stage writing batches IN PLACE into one of six sets through one native call per batch "
                         "(MiniBatchPipeline.use_static_sets, dgll_hip_load_sampled_batch).  Interleaved A/B on one box (round 5, "
                         "tools/minibatch_inplace_ab.sh): in place 649-655 batches/s, GPU side 1.42-1.46 ms; copies 615-620, 1.58 ms")
    ap.add_argument("--mb-dense-kernel", choices=["auto", "4wave"], default="4wave",
                    help="minibatch: 4wave = every bf16 transform on the 4-wavefront MFMA kernel (dgll_hip_debug_tune(4, 1)) instead of the "
                         "persistent resident-weights one, whose workgroups need a whole CU's registers and LDS and wait for the loading "
                         "stream's wavefronts to drain")
    ap.add_argument("--mb-host-translate", action="store_true",
                    help="minibatch: turn the outermost hop's positions into ids on the host (16 threads) instead of by a device gather")
    ap.add_argument("--scale", type=int, default=27, help="rmat27: RMAT scale (27 = config 5; smaller for a quick run)")
    ap.add_argument("--rmat-order", choices=["auto", "transform-first", "aggregate-first"], default="auto",
                    help="rmat27: the layer as A.(X.W) (the reference's order; N = 1 default) or as (A.X).W (N > 1: own rows only)")
    ap.add_argument("--rmat-no-rebalance", action="store_true",
                    help="rmat27, N > 1: keep the built-in cost ratio instead of re-fitting it on the live ranks and re-cutting once")
    ap.add_argument("--rmat-no-allgather", action="store_true", help="rmat27, N > 1: skip the separately timed all-gather of the outputs")
    return ap.parse_args(argv)


def alg_bytes(nnz, n_rows, feat, x_bytes, y_bytes, weighted):
    """every edge is charged one full feature-row read."""
    return nnz * (feat * x_bytes + 4 + (4 if weighted else 0)) + n_rows * (feat * y_bytes + 8)


def gat_alg_bytes(nnz, n_rows, feat, esz, heads):
    """SURVEY.md section 8(d), fused GAT pass: per edge one feature row + the column id + one score per head; per row the
    output row, the row pointer, one score and one row sum per head."""
    return nnz * (feat * esz + 4 + 4 * heads) + n_rows * (feat * esz + 8 + 8 * heads)


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


# ---------------------------------------------------------------------------------------------------- host / CPU side
def host_info():
    """Threads torch will use, physical cores, CPU model (the GPU box's host; printed with every cpu_baseline)."""
    model, pairs = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    threads = os.cpu_count() or 1
    return {"threads": threads, "physical_cores": len(pairs) or threads, "cpu_model": model}


def median3(fn):
    fn()                                                              # warm-up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[1]


def sample_rows_csr(graph, sample_rows, seed):
    """(rows, host rowptr, host col, nnz) of `sample_rows` rows of `graph` drawn at random without replacement (seeded, kept in
    ascending order).  The engine's reorder puts the hubs of every community first, so a PREFIX of the rows over-represents long
    rows (round 3's sample: 56 edges per row against 50.5 on average); a random draw has the graph's own degree mix."""
    n = graph.n_rows
    if sample_rows >= n:
        rp = graph.rowptr.cpu()
        return n, rp, graph.col.cpu(), int(rp[-1])
    gen = torch.Generator(device=graph.device)
    gen.manual_seed(seed + 12345)
    rows = torch.randperm(n, generator=gen, device=graph.device)[:sample_rows].sort().values
    deg = graph.rowptr[rows + 1] - graph.rowptr[rows]
    rowptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=graph.device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    pos = torch.repeat_interleave(graph.rowptr[rows] - rowptr[:-1], deg, output_size=nnz) + torch.arange(nnz, device=graph.device)
    return int(rows.numel()), rowptr.cpu(), graph.col[pos].cpu(), nnz


def cpu_baseline_spmm(graph, feat, sample_rows, seed):
    """BASELINE.md section 5 on the GPU box's host cores, on a bounded sample of the SAME tensors (the first `sample_rows`
    rows of the adjacency the GPU ran, the same feature width, fp32): the reference's own op -- torch.spmm on a COO tensor
    (dgll/nn/Convolution/gcnconv.py:31, restated in oracle/torch_ref.spmm_coo) -- plus torch.sparse.mm on CSR and the C
    oracle (oracle/oracle.c + OpenMP); each 1 warm-up + median of 3."""
    import numpy as np

    from oracle import cref, torch_ref

    host = host_info()
    rows, rowptr_t, col_t, nnz = sample_rows_csr(graph, sample_rows, seed)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((graph.n_cols, feat), dtype=np.float32)
    xt = torch.from_numpy(x)
    torch.set_num_threads(host["threads"])
    cref.set_num_threads(host["threads"])
    deg = (rowptr_t[1:] - rowptr_t[:-1])
    row_t = torch.repeat_interleave(torch.arange(rows), deg)
    val_t = (1.0 / deg.clamp(min=1).to(torch.float32))[row_t]              # mean = D^-1 A, utils.py:171
    col64 = col_t.to(torch.int64)
    t_coo = median3(lambda: torch_ref.spmm_coo(row_t, col64, val_t, xt, rows))
    csr = torch.sparse_csr_tensor(rowptr_t, col64, val_t, size=(rows, graph.n_cols))
    t_csr = median3(lambda: torch.sparse.mm(csr, xt))
    rp, cc = rowptr_t.numpy(), col_t.numpy()
    t_c = median3(lambda: cref.spmm_csr(rp, cc, None, x, reduce="mean"))
    return {
        "value": nnz / t_coo, "unit": "edges/s", "cores": host["threads"], "kind": "port",
        "sample": "mean-SpMM of %d rows DRAWN AT RANDOM (seeded; %d edges) from the benchmark's own (reordered) adjacency against all %d feature rows, "
                  "F=%d fp32; value = the reference's op torch.spmm(adj_coo, X) (gcnconv.py:31) as restated in "
                  "oracle/torch_ref.spmm_coo; 1 warm-up + median of 3" % (rows, nnz, graph.n_cols, feat),
        "torch_sparse_mm_csr_edges_per_s": nnz / t_csr,
        "oracle_c_openmp_csr_edges_per_s": nnz / t_c,
        "algorithmic_GBps": alg_bytes(nnz, rows, feat, 4, 4, True) / t_coo / 1e9,
        "torch_version": torch.__version__, "cpu_model": host["cpu_model"], "threads": host["threads"],
        "physical_cores": host["physical_cores"],
        "seconds_per_run": {"torch_spmm_coo": t_coo, "torch_sparse_mm_csr": t_csr, "oracle_c_openmp": t_c},
    }


def cpu_baseline_gat(graph, heads, fo, alpha, sample_rows, seed):
    """The fused GAT forward pass on the host cores, first `sample_rows` rows of the benchmark's adjacency (with its
    self-loops) against all nodes' transformed features, fp32: the reference's op sequence of sparseGatConv.forward
    (gatconv.py:111-148: per-edge scores, exp(-leakyrelu), two sparse products, division, ELU) restated on the index lists
    (oracle/torch_ref.py, segment sums by index_add) -- the reference itself needs a dense N x N adjacency (gatconv.py:115)
    and cannot run at this size -- plus the C oracle (oracle/oracle.c, OpenMP)."""
    import ctypes as C

    import numpy as np

    from oracle import cref

    host = host_info()
    rows, rowptr_t, col_t, nnz = sample_rows_csr(graph, sample_rows, seed)
    rng = np.random.default_rng(seed)
    width = heads * fo
    h = rng.standard_normal((graph.n_cols, width), dtype=np.float32)
    s = rng.standard_normal((graph.n_cols, heads), dtype=np.float32)
    t = rng.standard_normal((graph.n_cols, heads), dtype=np.float32)
    torch.set_num_threads(host["threads"])
    cref.set_num_threads(host["threads"])
    ht, st, tt = torch.from_numpy(h), torch.from_numpy(s), torch.from_numpy(t)
    row_t = torch.repeat_interleave(torch.arange(rows), rowptr_t[1:] - rowptr_t[:-1])
    col64 = col_t.to(torch.int64)

    def torch_heads():
        outs = []
        for k in range(heads):                                            # the reference runs its heads one by one (gatconv.py:196)
            hk = ht[:, k * fo:(k + 1) * fo]
            e = torch.exp(-torch.nn.functional.leaky_relu(st[:rows, k][row_t] + tt[:, k][col64], alpha))
            den = torch.zeros(rows).index_add_(0, row_t, e)
            num = torch.zeros(rows, fo).index_add_(0, row_t, e[:, None] * hk[col64])
            outs.append(torch.nn.functional.elu(num / den[:, None]))
        return torch.cat(outs, 1)

    t_torch = median3(torch_heads)
    out = np.empty((rows, width), dtype=np.float32)
    rp, cc = np.ascontiguousarray(rowptr_t.numpy()), np.ascontiguousarray(col_t.numpy())
    lib = cref.lib()

    def oracle_pass():
        lib.oracle_gat_fwd_f32(rp.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p),
                               C.c_int64(width), s.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p), C.c_int64(width), None, None, C.c_int64(rows), C.c_int(heads),
                               C.c_int(fo), C.c_float(alpha), C.c_int(1), C.c_int(0))

    t_c = median3(oracle_pass)
    return {
        "value": nnz / t_torch, "unit": "edges/s", "cores": host["threads"], "kind": "port",
        "sample": "fused GAT forward pass (%d heads x %d, fp32) over %d rows drawn at random (seeded; %d edges) from the benchmark's adjacency "
                  "against all %d nodes; value = sparseGatConv.forward's op sequence (gatconv.py:111-148) restated on index lists "
                  "with torch CPU ops, heads one by one as gatconv.py:196; 1 warm-up + median of 3" % (heads, fo, rows, nnz, graph.n_cols),
        "oracle_c_openmp_edges_per_s": nnz / t_c,
        "algorithmic_GBps": gat_alg_bytes(nnz, rows, width, 4, heads) / t_torch / 1e9,
        "torch_version": torch.__version__, "cpu_model": host["cpu_model"], "threads": host["threads"],
        "physical_cores": host["physical_cores"],
        "seconds_per_run": {"torch_index_add_heads": t_torch, "oracle_c_openmp": t_c},
    }


# ---------------------------------------------------------------------------------------------------- roofline record
def build_stamp():
    from dgll_amd import build as hip_build

    try:
        with open(hip_build.STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def spmm_kernel_fragment(feat, dtype_name, weighted, extra, avg_len=None):
    """Name fragment of the kernel instantiation a launch of this kind runs (16-byte rows): the wave-per-row kernel, the
    row-per-slot kernel for short rows (spmm.hip's rule: <= 24 edges per row on average at up to 16 lanes per row, <= 4.5 at 32 --
    RMAT-27's 16.8 edges per row at F = 128), the flattened kernel for fp32 rows of 16 / 32 lanes (>= 8 edges per row)."""
    bf = "bfloat16" in dtype_name
    epv = 8 if bf else 4
    vecs = -(-feat // epv)
    lpr = 4
    while lpr < 64 and lpr < vecs:
        lpr <<= 1
    t = "unsigned short" if bf else "float"
    w, e = "true" if weighted else "false", "true" if extra else "false"
    if avg_len is not None and ((lpr <= 16 and avg_len <= 24.0) or (lpr == 32 and avg_len <= 4.5)):
        return "spmm_rowslot_kernel<%s, %s, %d, %d, %s, %s>" % (t, t, epv, lpr, w, e)
    if avg_len is not None and not bf and 8 < vecs <= 32 and avg_len >= 8.0:
        return "spmm_csr_flat_kernel<%s, %s, %d, %d, %s, 4, %s>" % (t, t, epv, 16 if vecs <= 16 else 32, w, e)
    return "spmm_csr_kernel<%s, %s, %d, %d, %s, 4, %s, false>" % (t, t, epv, lpr, w, e)


def gat_kernel_fragment(heads, fo, dtype_name, kind, packed=False, rowscore=False):
    """gat2_kernel instantiation of a pass (kind 0 forward, 3 rows in the exact-dd form -- 1: its stored-output form --, 2 transposed rows) -- edge.hip's gat2_pick / gat2_inrow."""
    bf = "bfloat16" in dtype_name
    epv = 8 if bf else 4
    vph = fo // epv
    lph = 1
    while lph < vph:
        lph <<= 1
    nh = 1
    for cand in (8, 4, 2, 1):
        if cand * lph > 64 or (cand > 2 and heads % cand) or (cand > 1 and cand // 2 >= heads):
            continue
        nh = cand
        break
    while nh * lph < 4:
        lph <<= 1
    t = "unsigned short" if bf else "float"
    inrow = packed and heads == 1 and nh == 1 and vph < lph        # scores behind the row's last column, an idle lane to fetch them
    # (last argument: the row-score form -- t_j formed from the gathered row, dgll_hip_gat_fwd_rowscore)
    return "gat2_kernel<%s, %s, %d, %d, %d, 4, %d, %s, %s>" % (t, t, epv, nh * lph, nh, kind, "true" if inrow else "false",
                                                               "true" if rowscore else "false")


def load_traffic(sig, fragment):
    """Counter traffic of one launch kind from profiles/traffic.json: the entry must match the workload signature, the kernel
    instantiation AND the build stamp of the libdgll_hip.so that is loaded now -- a stale entry yields None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entries = json.load(f).get("entries", [])
    except (OSError, ValueError):
        return None
    stamp = build_stamp()
    for e in entries:
        w = e.get("workload", {})
        if e.get("kernel_fragment") == fragment and e.get("build_stamp") == stamp and stamp and \
                all(w.get(k) == v for k, v in sig.items()):
            return e
    return None


def roofline_record(dom, sig, kernel_desc, compulsory, world, extra=None):
    """One `roofline` object.  `achieved` = algorithmic bytes (section 8(d)) / average live launch time.  `frac`: ONE definition,
    named in frac_definition."""
    achieved = dom["algorithmic_GBps"]
    rec = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac_algorithmic": achieved / HBM_PEAK_GBPS, "traffic": None,
           "kernel": kernel_desc, "kernel_fragment": dom["kernel_fragment"], "launches_timed": dom["count"],
           "avg_launch_ms": dom["avg_ms"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes"],
           "compulsory_bytes_per_launch": compulsory, "edges_per_s_this_kernel": dom["nnz"] / (dom["avg_ms"] * 1e-3),
           "build_stamp": build_stamp(), "traffic_signature": sig}
    # `frac` is a FRACTION (<= 1) of the HBM peak:
    #   * when profiles/traffic.json holds counter traffic for this workload, kernel instantiation AND this build of libdgll_hip.so:
    #     frac = counter bytes per launch / live launch time / peak  (= frac_l2_miss_path).  The bytes are what the L2s requested from
    #     the fabric (rocprofv3 FETCH_SIZE / WRITE_SIZE, corrected by the ratios calibrated in the same pass); Infinity-Cache hits are
    #     INCLUDED (this box's rocprofv3 exposes no DRAM-side counter: profiles/r05_rocprofv3_counter_names.txt), so it is an UPPER
    #     BOUND on HBM utilisation;
    #   * otherwise frac = min(1, frac_algorithmic) and frac_definition says that no counter entry matched.
    # Always next to it: frac_algorithmic (section 8(d): every edge charged one full feature row / time / peak; > 1 = cache-served
    # re-reads), traffic_over_compulsory (counter bytes / compulsory bytes: the remaining headroom is THIS ratio), and for the sage
    # headline frac_conservative (structure-free graph in raw id order, formula; filled in by the caller).
    rec["frac"] = min(1.0, rec["frac_algorithmic"])
    rec["frac_definition"] = ("no counter entry in profiles/traffic.json for this workload / kernel instantiation / build: min(1, "
                              "frac_algorithmic) -- algorithmic bytes (SURVEY 8(d)) / live launch time / peak, capped at 1 (caches serve re-reads)")
    rec["frac_l2_miss_path"] = None
    rec["traffic_over_compulsory"] = None
    entry = load_traffic(sig, dom["kernel_fragment"]) if world == 1 else None
    if entry is not None:
        rec["traffic"] = entry["hbm_bytes_per_launch"]
        rec["traffic_source"] = {k: entry.get(k) for k in ("round", "fetch_size_kib", "write_size_kib", "ratio_read", "ratio_write",
                                                           "l2_hit_rate", "avg_ns_under_pmc")}
        hbm = entry["hbm_bytes_per_launch"] / (dom["avg_ms"] * 1e-3) / 1e9
        rec["achieved_l2_miss_path"] = hbm
        rec["frac_l2_miss_path"] = hbm / HBM_PEAK_GBPS
        rec["frac"] = min(1.0, hbm / HBM_PEAK_GBPS)
        rec["frac_definition"] = ("counter bytes per launch (profiles/traffic.json, same build: FETCH_SIZE / ratio_read + WRITE_SIZE / "
                                  "ratio_write = what the L2s requested from the fabric; Infinity-Cache hits included, so an UPPER BOUND on "
                                  "HBM utilisation) / live launch time / peak.  `achieved` stays the section-8(d) algorithmic rate "
                                  "(frac_algorithmic = achieved / peak, > 1 when caches serve re-reads)")
        rec["frac_l2_miss_path_definition"] = rec["frac_definition"]
        if compulsory:
            rec["traffic_over_compulsory"] = entry["hbm_bytes_per_launch"] / compulsory
    if extra:
        rec.update(extra)
    return rec


def launch_tables(launches, local_rows, heads_of=None):
    """Per launch kind of the timed steps (HIP events on the launch stream, ops.LaunchTimer): gather passes and dense kernels."""
    table, dense_table = {}, {}
    for tag, (cnt, avg_ms) in launches.items():
        if tag[0] == "spmm":
            _, feat, dt, weighted, tag_nnz = tag[:5]
            extra = tag[5] if len(tag) > 5 else ""
            xb = 2 if "bfloat16" in dt else 4
            b_alg = alg_bytes(tag_nnz, local_rows, feat, xb, xb, weighted)
            name = "spmm F=%d %s %s%s nnz=%d" % (feat, dt.replace("torch.", ""), "weighted" if weighted else "unweighted",
                                                 (" " + extra) if extra else "", tag_nnz)
            table[name] = {"count": cnt, "avg_ms": avg_ms, "nnz": tag_nnz, "feat": feat, "weighted": bool(weighted),
                           "epilogue": extra,
                           "kernel_fragment": spmm_kernel_fragment(feat, dt, weighted, bool(extra), tag_nnz / max(local_rows, 1))}
        elif tag[0] == "fused_sage":
            # aggregate -> transform in one launch: the SpMM's section-8(d) bytes (every edge one feature row; the aggregated row is
            # still written for the backward pass) plus the transform's own operands (self rows read, output rows written)
            _, rows, feat, k1, n_out, tag_nnz = tag
            b_alg = alg_bytes(tag_nnz, local_rows, feat, 2, 2, False) + rows * (k1 + n_out) * 2
            name = "fused aggregate->transform F=%d K1=%d N=%d bfloat16 nnz=%d" % (feat, k1, n_out, tag_nnz)
            vecs = -(-feat // 8)
            lpr = 8
            while lpr < 32 and lpr < vecs:
                lpr <<= 1
            table[name] = {"count": cnt, "avg_ms": avg_ms, "nnz": tag_nnz, "feat": feat, "weighted": False, "epilogue": "fused transform",
                           "kernel_fragment": "fused_sage_kernel<%d, false, %d, 4>" % (lpr, 2 if n_out > 128 else 1)}
        elif tag[0] == "gat":
            _, kind, heads, fo, dt, tag_nnz, packed = tag
            xb = 2 if "bfloat16" in dt else 4
            b_alg = gat_alg_bytes(tag_nnz, local_rows, heads * fo, xb, heads)
            name = "gat %s %d heads x %d %s%s nnz=%d" % (kind, heads, fo, dt.replace("torch.", ""), (" " + packed) if packed else "", tag_nnz)
            table[name] = {"count": cnt, "avg_ms": avg_ms, "nnz": tag_nnz, "feat": heads * fo, "heads": heads, "pass": kind,
                           "scores_in_row_padding": packed == "packed", "scores_from_gathered_rows": packed == "rowscore",
                           "kernel_fragment": gat_kernel_fragment(heads, fo, dt, {"fwd": 0, "bwd_rows": 3, "bwd_cols": 2}[kind],
                                                                  packed == "packed", packed == "rowscore")}
        elif tag[0] in ("transform", "transform_dual", "grad_weight"):
            kind, m, k1, k2, n_out, extra = tag
            parts = extra.split("+") if extra else []
            b_alg = m * (k1 + k2 + n_out) * 2 + (m * n_out * 2 if ("gate" in parts or "addend" in parts) else 0)
            # a gate read / sign bits written as one bit per element: 4 bytes per 32 columns, rows padded to 16 bytes
            b_alg += m * 16 * -(-n_out // 128) * (("gatebits" in parts) + ("signbits" in parts))
            name = "%s M=%d K=%d%s N=%d%s" % (kind, m, k1, ("+%d" % k2) if k2 else "", n_out, (" " + extra) if extra else "")
            dense_table[name] = {"count": cnt, "avg_ms": avg_ms, "algorithmic_bytes": b_alg,
                                 "algorithmic_GBps": b_alg / (avg_ms * 1e-3) / 1e9,
                                 "frac_of_hbm_peak": b_alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                 "frac_of_streaming_ceiling_5500": b_alg / (avg_ms * 1e-3) / 1e9 / 5500.0}
            continue
        elif tag[0] in ("transform_f32", "grad_weight_f32"):
            # fp32 products (the reference's own arithmetic, dgll/__init__.py:1) on v_mfma_f32_32x32x2_f32: matrix-core bound at these
            # widths (157.3 TFLOP/s dense fp32 MFMA peak of 256 CUs at 2.4 GHz: MI355X_MICROARCH.md), bytes next to it
            kind, m, k1, k2, n_out, extra = tag
            flops = 2.0 * m * (k1 + k2) * n_out
            b_alg = m * (k1 + k2 + n_out) * 4 + (m * n_out * 4 if extra else 0)
            name = "%s M=%d K=%d%s N=%d%s" % (kind, m, k1, ("+%d" % k2) if k2 else "", n_out, (" " + extra) if extra else "")
            dense_table[name] = {"count": cnt, "avg_ms": avg_ms, "algorithmic_bytes": b_alg, "flops": flops,
                                 "TFLOPs": flops / (avg_ms * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": flops / (avg_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                 "algorithmic_GBps": b_alg / (avg_ms * 1e-3) / 1e9,
                                 "frac_of_hbm_peak": b_alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            continue
        else:
            continue
        table[name].update({"G_edges_per_s": tag_nnz / (avg_ms * 1e-3) / 1e9, "algorithmic_bytes": b_alg,
                            "algorithmic_GBps": b_alg / (avg_ms * 1e-3) / 1e9,
                            "frac_algorithmic": b_alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    return table, dense_table


def workload_signature(args, nnz, **over):
    sig = {"workload": args.workload, "nodes": args.nodes, "nnz": nnz, "locality": args.locality,
           "permuted_ids": not args.no_permute, "reorder": args.reorder, "hidden": args.hidden, "dtype": args.dtype}
    sig.update(over)
    return sig


# ---------------------------------------------------------------------------------------------------- process set-up
class Ctx:
    pass


def setup(args):
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args))
    c = Ctx()
    c.world = int(env_world or "1")
    if c.world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, c.world), file=sys.stderr)
        sys.exit(2)
    c.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    c.backend = None
    if c.world > 1:
        import torch.distributed as dist

        # "nccl" is RCCL on ROCm.  DGLL_BENCH_BACKEND=gloo lets several ranks share one GPU (functional check of
        # the multi-rank path on a 1-GPU box; never used for reported numbers).
        c.backend = os.environ.get("DGLL_BENCH_BACKEND", "nccl")
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=c.backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    c.dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(c.dev)
    c.dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    c.esz = 2 if c.dtype == torch.bfloat16 else 4
    torch.manual_seed(args.seed)
    return c


def barrier(c):
    if c.world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def timed_steps(args, c, step):
    """W untimed steps, then exactly K steps between barrier + synchronize; MAX over ranks.  Returns (elapsed s, last loss,
    launch summary, warm-up loss trace)."""
    from dgll_amd import ops

    trace = []
    for _ in range(args.warmup):
        wl = step()
        if os.environ.get("DGLL_BENCH_TRACE_LOSS"):      # debugging aid: per-step global loss (costs a sync + all-reduce)
            g = wl.detach().double() / c.world
            if c.world > 1:
                torch.distributed.all_reduce(g)
            trace.append(float(g))
            if c.rank == 0:
                print("warm-up loss %.6f" % float(g), file=sys.stderr)
    barrier(c)
    loss = None
    marks = []                                      # one event per step boundary: the timed steps' own GPU durations (diagnostics)

    def mark():
        if c.dev.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)

    # The interpreter's cyclic garbage collector is switched off for the timed steps (as timeit does): right after the barrier the host
    # is not yet ahead of the GPU, and a full collection that falls into the first steps -- 8 ... 76 ms, measured as one 97 ms step among
    # 21 ms ones -- is idle GPU time in the measurement of a loop that is otherwise GPU-bound.
    import gc

    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        with ops.LaunchTimer() as timer:
            t0 = time.perf_counter()
            mark()
            for _ in range(args.steps):
                loss = step()
                mark()
            barrier(c)
            elapsed = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    timer.per_step_gpu_ms = [round(a.elapsed_time(b), 3) for a, b in zip(marks[:-1], marks[1:])]
    if c.world > 1:
        t = torch.tensor([elapsed], device=c.dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, loss, timer, trace


def build_products_graph(args, c, self_loops=False):
    """The products-sized graph (or --dataset), the engine's reordering, labels and input features in engine order."""
    from dgll_amd import synth

    gen = torch.Generator(device=c.dev)
    gen.manual_seed(args.seed + 1)
    if args.dataset:
        from dgll_amd.data import formats

        dg = formats.load_node_dataset(args.dataset, symmetrise=True) if not os.path.exists(os.path.join(args.dataset, "reddit_data.npz")) \
            else formats.load_node_dataset(args.dataset)
        full = dg.to_csr(c.dev)
        if self_loops:
            import dgll_amd

            row = torch.cat([full.row_index(), torch.arange(full.n_rows, device=c.dev)])
            col = torch.cat([full.col.long(), torch.arange(full.n_rows, device=c.dev)])
            full = dgll_amd.CSRGraph.from_coo(row, col, None, (full.n_rows, full.n_cols))
            full.val = None
        feats_all = dg.features.to(c.dev)
        labels_all = dg.labels.to(c.dev).long()
        args.nodes, args.in_feats, args.classes = full.n_rows, int(feats_all.shape[1]), int(labels_all.max()) + 1
        del dg
    else:
        full = synth.products_like_graph(c.dev, seed=args.seed, n=args.nodes, n_undirected=args.undirected_edges,
                                         locality=args.locality, exact=not args.inexact_edges, permute_ids=not args.no_permute,
                                         self_loops=self_loops)
        labels_all = torch.randint(0, args.classes, (full.n_rows,), generator=gen, device=c.dev)
        feats_all = torch.randn(full.n_rows, args.in_feats, generator=gen, device=c.dev)
    reorder_s, bounds = 0.0, None
    if args.reorder != "none":
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if c.world > 1:
            # partition + order in one pass: communities packed into `world` parts of equal edge count (the place METIS has in
            # the reference's pipeline), each part in locality order; deterministic, so every rank derives the same relabelling
            from dgll_amd import partition as dpart
            from dgll_amd import reorder as dreorder

            perm, bounds = dpart.partition_and_order(full, c.world, seed=args.seed)
            full = dreorder.relabel(full, perm)
        else:
            full, perm = full.reorder(method=args.reorder, seed=args.seed)     # new row i = old row perm[i]
        torch.cuda.synchronize()
        reorder_s = time.perf_counter() - t0
        labels_all = labels_all[perm]
        feats_all = feats_all[perm]
    return full, feats_all, labels_all, reorder_s, bounds


def make_engine(args, c, full, feats_all, labels_all, bounds):
    """Multi-rank: every rank keeps ONLY its own row block (what it would load from its part file) and learns what its peers
    need from one exchange of halo ids.  Returns (engine, x_local, labels)."""
    from dgll_amd import dist as ddist
    from dgll_amd import ops

    n = full.n_rows
    if bounds is None:
        bounds = [(n * r) // c.world for r in range(c.world + 1)]
    b0, b1 = bounds[c.rank], bounds[c.rank + 1]
    e0, e1 = int(full.rowptr[b0]), int(full.rowptr[b1])
    own_rowptr, own_col = full.rowptr[b0:b1 + 1].clone(), full.col[e0:e1].clone()
    del full
    torch.cuda.empty_cache()
    part = ddist.partition_rows(own_rowptr, own_col, None, bounds, c.rank)
    del own_rowptr, own_col
    # which form of the first two layers (recompute on halo rows / exchange hidden-width rows) wins depends on the fabric: unless the
    # caller fixed it, let the live ranks decide during warm-up (DistGraph.resolve_halo_mode, called by the workload)
    os.environ.setdefault("DGLL_HALO_MODE", "auto")
    engine = ddist.DistGraph(part, c.dev)
    engine.verify()          # exchange lists agree across ranks + the start-up self-test of the chosen exchange form (DGLL_EXCHANGE)
    x_local = ops.alloc_features(part.n_own, args.in_feats, c.dtype, c.dev, pad_to=args.feat_align)
    x_local.copy_(engine.permute_to_local(feats_all[part.own_begin:part.own_end]).to(c.dtype))
    labels = engine.permute_to_local(labels_all[part.own_begin:part.own_end])
    return engine, x_local, labels


def base_result(args, c, value, elapsed, workload_text, config):
    ms = elapsed / args.steps * 1e3
    cfg = {"workload": workload_text, "workload_id": args.workload}
    cfg.update(config)
    cfg.update({"ranks": c.world, "backend": c.backend})
    return {"metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": c.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": ("real: " + args.dataset) if args.dataset else "synthetic", "config": cfg, "epoch_time_s": ms / 1e3}


def calibrate_launches(args, c, n):
    """Three launches with KNOWN bytes in the gather kernels' own access pattern (identity gather of n rows: streaming read +
    write of n * hidden * esz bytes each): the PMC passes turn them into the FETCH_SIZE / WRITE_SIZE correction ratios."""
    import dgll_amd
    from dgll_amd import ops

    from dgll_amd import _lib

    ident = dgll_amd.CSRGraph.fixed_fanout(n, 1, c.dev)
    xc = ops.alloc_features(n, args.hidden, c.dtype, c.dev)
    xc.copy_(torch.randn(n, args.hidden, device=c.dev).to(c.dtype))
    # one-edge rows would be handed to the row-per-slot kernel: the calibration must run on the wave-per-row kernel the measured
    # launches use (tools/pmc_parse.py looks for it by name)
    _lib.check(_lib.lib.dgll_hip_debug_tune(5, 1), "tune")
    try:
        for _ in range(3):
            ops.spmm_raw(ident, xc, reduce="sum")
    finally:
        _lib.check(_lib.lib.dgll_hip_debug_tune(5, 0), "tune")
    del ident, xc


# ---------------------------------------------------------------------------------------------------- workload: sage
def run_sage(args, c):
    from dgll_amd import dist as ddist
    from dgll_amd import nn as dnn
    from dgll_amd import ops

    full, feats_all, labels_all, reorder_s, bounds = build_products_graph(args, c)
    n, nnz = full.n_rows, full.nnz
    model = dnn.GraphSage(args.in_feats, [args.hidden, args.hidden, args.classes], None).to(c.dev)
    racom = opt_wrap = engine = None
    if c.world > 1:
        engine, x_local, labels = make_engine(args, c, full, feats_all, labels_all, bounds)
        placed_input = engine.place_input_halo(x_local)     # input features of halo nodes live with the partition
        graph_for_cpu = None
    else:
        x_local = ops.alloc_features(n, args.in_feats, c.dtype, c.dev, pad_to=args.feat_align)
        x_local.copy_(feats_all.to(c.dtype))
        labels = labels_all
        graph_for_cpu = full
        full.plan()
        full.transpose()[0].plan()
        full.mean_scale_transposed()
    del feats_all
    # Adam as ONE launch over flat parameter / gradient buffers (dgll_amd/optim.py: torch.optim.Adam's arithmetic): the weight-gradient
    # kernels write into the gradient buffer, RaCoM all-reduces that buffer in place, the update kernel also emits the packed bf16
    # weights of the MFMA transforms.  --torch-adam: torch.optim.Adam + per-parameter bucket copies, as in rounds 1-3.
    from dgll_amd.optim import FlatAdam

    params = list(model.parameters())
    opt = torch.optim.Adam(params, lr=1e-3) if args.torch_adam else FlatAdam(params, lr=1e-3)
    flat = None if args.torch_adam else opt
    if c.world > 1:
        if args.racom_async:
            opt_wrap = ddist.RaCoMOptimizer(opt, params, c.dev, staleness=1, sync_every=ddist.racom_sync_period(n, c.world))
        else:
            racom = ddist.RaCoM(params, c.dev, flat=flat)
    # forward: 3 layers; backward: layers 2 and 3 (the input features need no gradient).  The last layer narrows
    # (256 -> 47), so it aggregates the 47-wide product X.W_n instead of the 256-wide input (mean is linear).
    passes = 3 + 2
    world = c.world

    def step():
        opt.zero_grad(set_to_none=True)
        if engine is None:
            out = model.forward_graph(full, x_local)
        else:
            out = engine.sage_forward(model, x_local, placed_input)
        # cross-entropy summed over this rank's nodes / global node count (x world: RaCoM averages over ranks)
        loss = ops.cross_entropy(out, labels, reduction="sum", fold_relu=True) * (world / n)     # one kernel per direction
        loss.backward()
        if opt_wrap is not None:
            opt_wrap.step()
        else:
            if racom is not None:
                racom.all_reduce_and_wait()
            opt.step()
        return loss

    if args.calibrate and c.world == 1:
        calibrate_launches(args, c, n)
    halo_mode = None
    if engine is not None:        # DGLL_HALO_MODE=auto: both forms of the first two layers timed on the live ranks, the faster kept
        halo_mode = {"mode": engine.resolve_halo_mode(step), "requested": os.environ.get("DGLL_HALO_MODE", "recompute"),
                     "timings_ms": engine.halo_mode_timings}
    elapsed, loss, timer, trace = timed_steps(args, c, step)
    if opt_wrap is not None:
        opt_wrap.flush()
    global_loss = loss.detach().double() / c.world      # this rank's share of the mean loss
    per_rank = None
    if c.world > 1:
        torch.distributed.all_reduce(global_loss)
        per_rank = per_rank_diagnostics(c, engine, step, racom=racom, opt_wrap=opt_wrap)
    if c.rank != 0:
        return None
    local_rows = n if engine is None else engine.part.n_own
    table, dense_table = launch_tables(timer.summary(), local_rows)
    # headline = the LONGEST hidden-width SpMM launch of the step (forward mean aggregation or the weighted, gated,
    # accumulating transposed launch of the backward pass, whichever takes longer)
    roofline = None
    wide = {k: v for k, v in table.items() if v["feat"] == args.hidden}
    if wide:
        dom = wide[max(wide, key=lambda k: wide[k]["avg_ms"])]
        n_cols_touched = n if engine is None else engine.part.n_own + engine.part.n_halo
        compulsory = dom["nnz"] * (8 if dom["weighted"] else 4) + n_cols_touched * args.hidden * c.esz + local_rows * (args.hidden * c.esz + 8)
        roofline = roofline_record(dom, workload_signature(args, nnz),
                                   "spmm_csr_kernel %s %s%s, F=%d (the longest SpMM-type launch of the step)" % (
                                       args.dtype, "weighted" if dom["weighted"] else "unweighted",
                                       (" " + dom["epilogue"]) if dom["epilogue"] else "", args.hidden), compulsory, c.world)
    result = base_result(
        args, c, passes * nnz * args.steps / elapsed, elapsed,
        "full-graph 3-layer GraphSAGE (mean aggr, %d-%d-%d-%d) training step on an ogbn-products-sized synthetic graph: %d nodes, "
        "nnz %d, 64 planted communities (locality %.2f), node ids %s, engine reorder: %s" % (
            args.in_feats, args.hidden, args.hidden, args.classes, n, nnz, args.locality,
            "community-sorted" if args.no_permute else "randomly permuted", args.reorder),
        {"nodes": n, "nnz": nnz, "hidden": args.hidden, "locality": args.locality, "permuted_ids": not args.no_permute,
         "reorder": args.reorder, "reorder_seconds_one_off": reorder_s, "parallelism": "1-D row partition x%d" % c.world,
         "spmm_launches_per_step": passes,
         "gradient_sharing": None if c.world == 1 else ("RaCoM async (staleness 1)" if opt_wrap is not None else "RaCoM sync")})
    result.update({"loss": float(global_loss), "roofline": roofline, "spmm_launch_table": table,
                   # bytes = 2 (K1 + K2 + N) per row (+ 2 N for a bf16 gate / addend operand, + 16 ceil(N / 128) for a gate read or sign
                   # bits written as bits); the 5.5 TB/s "streaming ceiling" is what
                   "dense_launch_table": dense_table})      # a trivial 2R:1W kernel reaches (tools/probes/rw_mix.hip)
    # this rank's GPU time of every timed step (events at the step boundaries): the first step after the barrier runs ~1.5 ms longer
    # (the GPU idled through it), the rest sit at the steady state -- what a longer K converges to
    result["per_step_gpu_ms"] = getattr(timer, "per_step_gpu_ms", None)
    if trace:
        result["warmup_loss_trace"] = trace
    if per_rank is not None:
        result["per_rank"] = per_rank
        result["config"]["exchange_form"] = engine.exchange.form
        result["config"]["halo_mode"] = halo_mode
    if c.world == 1 and not args.no_extra and not args.dataset:
        del model, opt
        result["roofline_no_locality"] = extra_roofline(args, c, locality=0.0, permute=False, reorder=args.reorder,
                                                        note="structure-free RMAT (locality 0: no communities), engine reorder '%s' as in "
                                                             "the headline run" % args.reorder)
        if args.reorder != "none":
            raw = extra_roofline(args, c, locality=0.0, permute=False, reorder="none",
                                 note="structure-free RMAT in the generator's own id order (hubs at low ids), no reordering: round 1's figure")
            result["roofline_no_locality_raw_order"] = raw
            if roofline is not None:      # the most conservative reading: no structure, no reordering, section-8(d) formula
                roofline["frac_conservative"] = raw["frac_algorithmic"]
            result["roofline_raw_order"] = extra_roofline(args, c, locality=args.locality, permute=not args.no_permute,
                                                          reorder="none", note="the headline graph WITHOUT the engine's reordering pass")
    if c.world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_spmm(graph_for_cpu, args.hidden, args.cpu_sample_rows, args.seed)
    return result


def extra_roofline(args, c, locality, permute, reorder, note):
    """The forward hidden-width mean-SpMM on another variant of the graph (same size, same kernel), 10 back-to-back launches."""
    from dgll_amd import ops, synth

    g = synth.products_like_graph(c.dev, seed=args.seed, n=args.nodes, n_undirected=args.undirected_edges, locality=locality,
                                  exact=not args.inexact_edges, permute_ids=permute)
    if reorder != "none":
        g, _ = g.reorder(method=reorder, seed=args.seed)
    g.plan()
    x = ops.alloc_features(g.n_cols, args.hidden, c.dtype, c.dev)
    x.copy_(torch.randn(g.n_cols, args.hidden, device=c.dev).to(c.dtype))
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
    b_alg = alg_bytes(g.nnz, g.n_rows, args.hidden, c.esz, c.esz, weighted=False)
    dom = {"count": reps, "avg_ms": ms, "nnz": g.nnz, "algorithmic_bytes": b_alg, "algorithmic_GBps": b_alg / (ms * 1e-3) / 1e9,
           "kernel_fragment": spmm_kernel_fragment(args.hidden, str(c.dtype), False, False)}
    sig = workload_signature(args, g.nnz, locality=locality, permuted_ids=permute, reorder=reorder)
    compulsory = g.nnz * 4 + g.n_cols * args.hidden * c.esz + g.n_rows * (args.hidden * c.esz + 8)
    return roofline_record(dom, sig, "spmm_csr_kernel %s unweighted, F=%d" % (args.dtype, args.hidden), compulsory, 1,
                           extra={"nnz": g.nnz, "note": "forward mean-SpMM F=%d, %s; %d back-to-back launches (timed with the long-row "
                                                        "finalize)" % (args.hidden, note, reps)})


# ---------------------------------------------------------------------------------------------------- workload: gat
def run_gat(args, c):
    """BASELINE config 4: 2-layer SpGAT (gatconv.py:174-199), `heads` x (hidden / heads) -> classes, products-sized graph with
    self-loops (rows without edges are NaN in the reference, gatconv.py:139-141), attention dropout inactive."""
    from dgll_amd import dist as ddist
    from dgll_amd import nn as dnn
    from dgll_amd import ops

    heads, fo = args.heads, args.hidden // args.heads
    full, feats_all, labels_all, reorder_s, bounds = build_products_graph(args, c, self_loops=True)
    n, nnz = full.n_rows, full.nnz
    model = dnn.SpGAT(args.in_feats, fo, args.classes, dropout=0.0, alpha=0.2, nheads=heads).to(c.dev)
    racom = engine = None
    if c.world > 1:
        engine, x_local, labels = make_engine(args, c, full, feats_all, labels_all, bounds)
        placed_input = engine.place_input_halo(x_local)     # input features of halo nodes live with the partition
        graph_for_cpu = None
    else:
        x_local = ops.alloc_features(n, args.in_feats, c.dtype, c.dev, pad_to=args.feat_align)
        x_local.copy_(feats_all.to(c.dtype))
        labels = labels_all
        graph_for_cpu = full
        full.plan()
        full.transpose()[0].plan()
    del feats_all
    from dgll_amd.optim import FlatAdam

    params = list(model.parameters())
    opt = torch.optim.Adam(params, lr=1e-3) if args.torch_adam else FlatAdam(params, lr=1e-3)
    if c.world > 1:
        racom = ddist.RaCoM(params, c.dev, flat=None if args.torch_adam else opt)
    passes = 2 * 3                # per layer: forward, backward over the rows of A, backward over the rows of A^T
    world = c.world

    def step():
        opt.zero_grad(set_to_none=True)
        if args.gat_unfused_loss:
            out = model(x_local, full) if engine is None else engine.spgat_forward(model, x_local, placed_input)     # log_softmax
            # F.nll_loss of the reference's training loops, as its definition (torch's nll_loss kernels take 10 ms at this size)
            loss = -out.gather(1, labels.unsqueeze(1)).float().sum() * (world / n)
        else:
            # the same loss, nll_loss(log_softmax(a)) = cross_entropy(a), on the model's activations before its log_softmax: one
            # kernel per direction instead of log_softmax + gather (+ their backward passes) over an fp32 copy of [N, 47]
            act = model.forward_activations(x_local, full) if engine is None else engine.spgat_forward(model, x_local, placed_input,
                                                                                                       activations=True)
            loss = ops.cross_entropy(act, labels, reduction="sum") * (world / n)
        loss.backward()
        if racom is not None:
            racom.all_reduce_and_wait()
        opt.step()
        return loss

    if args.calibrate and c.world == 1:
        calibrate_launches(args, c, n)
    if engine is not None:
        engine.resolve_halo_mode(step)          # DGLL_HALO_MODE=auto: recompute / exchange of the first layer timed on the live ranks
    elapsed, loss, timer, trace = timed_steps(args, c, step)
    global_loss = loss.detach().double() / c.world
    if c.world > 1:
        torch.distributed.all_reduce(global_loss)
    if c.rank != 0:
        return None
    local_rows = n if engine is None else engine.part.n_own
    table, dense_table = launch_tables(timer.summary(), local_rows)
    roofline = None
    wide = {k: v for k, v in table.items() if v["feat"] == args.hidden and "pass" in v}
    if wide:
        dom = wide[max(wide, key=lambda k: wide[k]["avg_ms"])]
        n_cols_touched = n if engine is None else engine.part.n_own + engine.part.n_halo
        compulsory = dom["nnz"] * 4 + n_cols_touched * (args.hidden * c.esz + 4 * heads) + local_rows * (args.hidden * c.esz + 8 + 4 * heads)
        roofline = roofline_record(dom, workload_signature(args, nnz, heads=heads),
                                   "gat2_kernel %s, %d heads x %d, pass %s (the longest gather pass of the step)" % (
                                       args.dtype, heads, fo, dom["pass"]), compulsory, c.world)
    result = base_result(
        args, c, passes * nnz * args.steps / elapsed, elapsed,
        "BASELINE config 4: full-graph 2-layer SpGAT (%d -> %d heads x %d -> %d, alpha 0.2, attention dropout off) training step on "
        "the ogbn-products-sized synthetic graph + self-loops: %d nodes, nnz %d, locality %.2f, node ids %s, engine reorder: %s" % (
            args.in_feats, heads, fo, args.classes, n, nnz, args.locality,
            "community-sorted" if args.no_permute else "randomly permuted", args.reorder),
        {"nodes": n, "nnz": nnz, "hidden": args.hidden, "heads": heads, "locality": args.locality, "permuted_ids": not args.no_permute,
         "reorder": args.reorder, "reorder_seconds_one_off": reorder_s, "parallelism": "1-D row partition x%d" % c.world,
         "gather_passes_per_step": passes})
    result.update({"loss": float(global_loss), "roofline": roofline, "spmm_launch_table": table, "dense_launch_table": dense_table,
                   "per_step_gpu_ms": getattr(timer, "per_step_gpu_ms", None)})
    if trace:
        result["warmup_loss_trace"] = trace
    if c.world == 1 and not args.no_extra:
        # the SpMM at the same width on the same graph, same build: what the GAT passes are judged against
        x = ops.alloc_features(n, args.hidden, c.dtype, c.dev)
        x.copy_(torch.randn(n, args.hidden, device=c.dev).to(c.dtype))
        for _ in range(2):
            ops.spmm_raw(full, x, reduce="mean")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.spmm_raw(full, x, reduce="mean")
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        result["spmm_same_width_same_graph_ms"] = ms
        result["gat_pass_over_spmm"] = {v["pass"]: v["avg_ms"] / ms for v in wide.values()}
        del x
    if c.world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_gat(graph_for_cpu, heads, fo, 0.2, min(args.cpu_sample_rows, 20_000), args.seed)
    return result


# ---------------------------------------------------------------------------------------------------- workload: rmat27
def run_rmat27(args, c):
    """BASELINE config 5: one GCN layer Y = A . X . W (gcnconv.py:30-31) at RMAT-27 size, F = 128 bf16, mean weights, > 2^31
    nonzeros (int64 row pointers).

    N = 1: a step = the MFMA transform S = X.W, then the SpMM Y = A.S -- the reference's order.
    N > 1 (process shape MQGCN.py:161-163): A is cut into N contiguous COST-balanced row blocks (cut
    points on the prefix sum of bytes per row, dgll_amd.dist.cost_balanced_bounds: on the hubs-first order equal-row blocks are
    not equal-nnz blocks), X is REPLICATED on every rank (34 GB of 288 at bf16), and every rank runs the same two kernels on its
    block in the order (A.X).W: the SpMM over its rows against the full X, then the transform of its own rows -- 1/N of both, no
    exchange inside the step (A.(X.W) would need S = X.W for ALL rows on every rank: N times the transform, or an all-gather of
    34 GB per step).  value = sum of the ranks' nonzeros / the slowest rank's time; `per_rank` carries every rank's rows, nonzeros,
    launch times and roofline; the all-gather of the outputs a FOLLOWING layer would need is timed separately (`output_allgather`)."""
    from dgll_amd import dense, dist as ddist, ops, synth

    feat = 128
    t0 = time.time()
    g = synth.rmat_graph(args.scale, 16, seed=args.seed, device=c.dev, symmetric=False, weighted=False, self_loops=True)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    t0 = time.time()
    reorder = "degree" if g.nnz >= (1 << 30) else "lpa"
    if args.reorder != "none":
        g, _ = g.reorder(method=reorder, seed=args.seed)
    torch.cuda.synchronize()
    reorder_s = time.time() - t0
    torch.cuda.empty_cache()
    n, nnz_total = g.n_rows, g.nnz
    order = args.rmat_order
    if order == "auto":
        order = "transform-first" if c.world == 1 else "aggregate-first"
    if c.world > 1 and order != "aggregate-first":
        raise SystemExit("bench.py --workload rmat27 --gpus N > 1 runs the layer as (A.X).W on replicated X (--rmat-order aggregate-first)")
    # cost of one rank's step per edge and per row, in picoseconds on one MI355X (tools/scaling_model.py rmat27, least squares over
    # the 15 blocks of N = 1, 2, 4, 8 at scale 27: 0.035 ns per edge, 0.183 ns per row -- profiles/r05_rmat27_scaling_model.log).  The
    # byte counts (260 B per edge, 776 B per row: ratio 3.0) under-charge the rows: a block of degree-1 rows is bound by rows per
    # second, not bytes (predicted 8-GPU speed-up 5.3x with the byte ratio, 7.4x with the fitted one).  --rmat-rebalance (default)
    # re-fits both on the live ranks after a first measurement and re-cuts once.
    edge_cost, row_cost = 35, 183
    shard = ddist.RowBlockShard(g, c.world, c.rank, edge_cost, row_cost)
    gen = torch.Generator(device=c.dev)
    gen.manual_seed(args.seed + 1)                       # the same X and W on every rank: X is replicated, not exchanged
    x = ops.alloc_features(n, feat, c.dtype, c.dev)
    chunk = 1 << 24
    for lo in range(0, n, chunk):                        # (a one-shot randn of [2^27, 128] fp32 is 69 GB of scratch)
        hi = min(n, lo + chunk)
        x[lo:hi] = torch.randn(hi - lo, feat, device=c.dev, generator=gen).to(c.dtype)
    w = (torch.randn(feat, feat, device=c.dev, generator=gen) / feat ** 0.5).to(c.dtype)
    rebalance = None
    if c.world > 1:
        shard.own_copy()
        if not args.rmat_no_rebalance:
            # one measured pass per rank -> (nnz, rows, ms) of every rank -> least-squares cost per edge and per row -> new cut points
            # (every rank computes the same ones from the same gathered numbers); kept when the model predicts a step >= 2 % shorter
            yb = ops.alloc_features(shard.n_own, feat, c.dtype, c.dev)

            def probe():
                out = ops.spmm_raw(shard.block, x, reduce="mean", out=yb)
                return dense.transform_bf16(out, w.t()) if c.dtype == torch.bfloat16 else out @ w

            for _ in range(2):
                probe()
            barrier(c)
            t0 = time.perf_counter()
            for _ in range(3):
                probe()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            del yb
            got = [None] * c.world
            torch.distributed.all_gather_object(got, (shard.block.nnz, shard.n_own, ms))
            a_ = torch.tensor([[t[0], t[1]] for t in got], dtype=torch.float64)
            sol = torch.linalg.lstsq(a_, torch.tensor([[t[2]] for t in got], dtype=torch.float64)).solution.flatten()
            per_edge, per_row = float(sol[0]), float(sol[1])
            rebalance = {"measured_ms": [t[2] for t in got], "fitted_ns_per_edge": per_edge * 1e6, "fitted_ns_per_row": per_row * 1e6,
                         "applied": False}
            if per_edge > 0 and per_row > 0:
                e2, r2 = max(1, int(round(per_edge * 1e9))), max(1, int(round(per_row * 1e9)))
                b2 = ddist.cost_balanced_bounds(g.rowptr, c.world, e2, r2)
                at = g.rowptr[torch.tensor(b2, device=c.dev)].tolist()
                pred = max(per_edge * (at[r + 1] - at[r]) + per_row * (b2[r + 1] - b2[r]) for r in range(c.world))
                rebalance["predicted_ms_after"] = pred
                if pred < 0.98 * max(t[2] for t in got):
                    shard = ddist.RowBlockShard(g, c.world, c.rank, bounds=b2).own_copy()
                    edge_cost, row_cost = e2, r2
                    rebalance["applied"] = True
        bounds, block_nnz = shard.bounds, shard.block_nnz
        del g
        torch.cuda.empty_cache()
        blk = shard.block
    else:
        bounds, block_nnz = shard.bounds, shard.block_nnz
        blk = g
    blk.plan()
    rows, nnz = blk.n_rows, blk.nnz
    y = ops.alloc_features(rows, feat, c.dtype, c.dev)

    def transform(a):
        return dense.transform_bf16(a, w.t()) if c.dtype == torch.bfloat16 else a @ w

    def step_transform_first():
        ops.spmm_raw(blk, transform(x), reduce="mean", out=y)                # S = X.W (gcnconv.py:30), Y = A.S (:31)
        return y

    calibrate_rows = None
    if args.calibrate and c.world == 1:       # the PMC passes' known-byte launches, in this workload's kernel instantiation (F = 128)
        import copy

        a2 = copy.copy(args)
        a2.hidden = feat
        calibrate_rows = min(n, 1 << 22)
        calibrate_launches(a2, c, calibrate_rows)

    def step_aggregate_first():
        return transform(ops.spmm_raw(blk, x, reduce="mean", out=y))         # the same layer as (A.X).W: own rows only

    step = step_transform_first if order == "transform-first" else step_aggregate_first

    def timed(fn, steps):
        for _ in range(args.warmup):
            fn()
        barrier(c)
        with ops.LaunchTimer() as timer:
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            own = time.perf_counter() - t0
            barrier(c)
            elapsed = time.perf_counter() - t0
        return own, elapsed, timer

    own_s, elapsed, timer = timed(step, args.steps)
    if c.world > 1:
        t = torch.tensor([elapsed], device=c.dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    table, dense_table = launch_tables(timer.summary(), rows)
    dom = table[max(table, key=lambda k: table[k]["avg_ms"])]
    n_src = n
    compulsory = nnz * 4 + (n_src + rows) * feat * c.esz + rows * 8
    mine = {"rank": c.rank, "rows": rows, "nnz": nnz, "row_range": [shard.own_begin, shard.own_end],
            "own_ms_per_step": own_s / args.steps * 1e3, "spmm_avg_ms": dom["avg_ms"], "spmm_algorithmic_GBps": dom["algorithmic_GBps"],
            "spmm_frac_algorithmic": dom["frac_algorithmic"], "spmm_G_edges_per_s": dom["G_edges_per_s"],
            "dense_ms": {k: v["avg_ms"] for k, v in dense_table.items()}}
    per_rank = [mine]
    if c.world > 1:
        per_rank = [None] * c.world
        torch.distributed.all_gather_object(per_rank, mine)
    allgather = None
    if c.world > 1 and not args.rmat_no_allgather:
        # what a FOLLOWING layer would need (its source rows = this layer's outputs of all ranks): N broadcasts into the slices of
        # one [n, F] buffer (blocks differ in rows: no padding, no staging copy); NOT part of the timed step
        try:
            y_all = torch.empty(n, feat, dtype=c.dtype, device=c.dev)
            out_rows = step()
            times = []
            for _ in range(3):
                barrier(c)
                t0 = time.perf_counter()
                shard.gather_output(out_rows, out=y_all)
                barrier(c)
                times.append(time.perf_counter() - t0)
            ms = sorted(times)[1] * 1e3
            nbytes = n * feat * c.esz
            allgather = {"ms": ms, "bytes_received_per_rank": nbytes - rows * feat * c.esz,
                         "GBps_ingress_per_rank": (nbytes - rows * feat * c.esz) / (ms * 1e-3) / 1e9,
                         "form": "dist.RowBlockShard.gather_output: N broadcasts into the row slices of one [n, F] buffer", "backend": c.backend}
            del y_all
        except Exception as exc:  # noqa: BLE001  (a diagnostic: the timed step does not depend on it)
            allgather = {"ms": None, "error": "%s: %s" % (type(exc).__name__, exc)}
    slow = max(per_rank, key=lambda r: r["own_ms_per_step"])
    sig = workload_signature(args, nnz_total, nodes=n, hidden=feat, scale=args.scale)
    roofline = roofline_record(dom, sig, "spmm_csr_kernel %s unweighted mean, F=%d, int64 row pointers%s" % (
        args.dtype, feat, "" if c.world == 1 else " (rank 0's row block; per_rank lists every rank)"), compulsory, c.world)
    result = base_result(
        args, c, nnz_total * args.steps / elapsed, elapsed,
        "BASELINE config 5%s: GCN layer Y = A.X.W (gcnconv.py:30-31) on RMAT-%d (%d nodes, nnz %d%s, self-loops), F = %d, "
        "engine reorder: %s; %s" % (
            " on one GPU" if c.world == 1 else " on %d ranks: cost-balanced contiguous row blocks, X replicated, no exchange in the step" % c.world,
            args.scale, n, nnz_total, " > 2^31" if nnz_total > 2 ** 31 else "", feat, reorder if args.reorder != "none" else "none",
            "step = X.W then A.(.)" if order == "transform-first" else "step = A.X on own rows then (.).W on own rows"),
        {"nodes": n, "nnz": nnz_total, "hidden": feat, "scale": args.scale, "reorder": reorder if args.reorder != "none" else "none",
         "graph_build_seconds": build_s, "reorder_seconds_one_off": reorder_s,
         "parallelism": "single GPU" if c.world == 1 else "row blocks x%d, X replicated" % c.world, "layer_order": order,
         "row_bounds": bounds, "block_nnz": block_nnz, "edge_cost": edge_cost, "row_cost": row_cost, "rebalance": rebalance,
         "calibrate_rows": calibrate_rows,
         "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9})
    result.update({"roofline": roofline, "spmm_launch_table": table, "dense_launch_table": dense_table, "per_rank": per_rank,
                   "slowest_rank": slow["rank"],
                   "imbalance_max_over_mean": slow["own_ms_per_step"] / (sum(r["own_ms_per_step"] for r in per_rank) / len(per_rank)),
                   "output_allgather": allgather})
    if c.world == 1 and order == "transform-first":
        # the N > 1 ranks run the layer as (A.X).W: the same launch kinds in the other order, measured here so that a scaling
        # ratio can be formed between equal steps
        _o, el2, _t = timed(step_aggregate_first, max(2, args.steps // 2))
        result["aggregate_first_ms_per_step"] = el2 / max(2, args.steps // 2) * 1e3
    if not args.no_cpu_baseline and c.rank == 0 and c.world == 1:
        del x, y
        result["cpu_baseline"] = cpu_baseline_rows_only(blk, feat, min(args.cpu_sample_rows, 1_000_000), args.seed)
    return result


def cpu_baseline_rows_only(graph, feat, sample_rows, seed):
    """Config 5's CPU baseline on a bounded sample: `sample_rows` rows drawn at random (the engine's reorder puts the hubs first:
    the first rows alone would hold half the edges), their column ids folded into the first 2^22 nodes so that the fp32 source
    matrix is 2 GB instead of 69 GB -- the same edges and row lengths, a smaller source matrix."""
    import dgll_amd

    fold = 1 << 22
    gen = torch.Generator(device=graph.device)
    gen.manual_seed(seed)
    rows = torch.randperm(graph.n_rows, generator=gen, device=graph.device)[:min(sample_rows, graph.n_rows)].sort().values
    deg = graph.rowptr[rows + 1] - graph.rowptr[rows]
    rowptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=graph.device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    pos = torch.repeat_interleave(graph.rowptr[rows] - rowptr[:-1], deg) + torch.arange(nnz, device=graph.device)
    col = (graph.col[pos].long() % fold).to(torch.int32)
    sub = dgll_amd.CSRGraph(rowptr, col, None, rows.numel(), fold, check=False)
    out = cpu_baseline_spmm(sub, feat, rows.numel(), seed)
    out["sample"] = ("%d rows of the benchmark's adjacency drawn at random (%d edges), column ids folded into the first 2^22 nodes "
                     "(fp32 source matrix 2 GB); " % (rows.numel(), nnz)) + out["sample"]
    return out


# ---------------------------------------------------------------------------------------------------- workload: minibatch
def run_minibatch(args, c):
    """BASELINE config 2 shape: Reddit-sized synthetic graph (232 965 nodes, 114.6 M directed edges, F = 602, 41 classes),
    3-layer GraphSAGE hidden 256 bf16, mini-batches of 1024 seeds with fan-out 25-10-10 (graphage.py:47-63):
      host: bit-exact native sampler -> bounded queue on a side stream (buffer_queues.py:22-46)
      GPU : one-launch hit/miss gather from the HBM hot-node cache + pinned host memory (storage.py:151-198), CSR blocks,
            hop-pyramid forward / backward / Adam.
    A step = one batch; W warm-up batches, K timed batches."""
    import gc
    import random

    import numpy as np

    from dgll_amd import nn as dnn
    from dgll_amd import ops, ranges, synth
    from dgll_amd.cache import GraphCacheServer
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler

    if c.world > 1:
        raise SystemExit("bench.py --workload minibatch runs one GPU (data-parallel replicas would each run this loop on a share of the seeds)")
    fanouts = [int(v) for v in args.mb_fanouts.split(",")]
    L = len(fanouts)
    g = synth.products_like_graph(c.dev, seed=1, n=args.mb_nodes, n_undirected=args.mb_undirected_edges, locality=0.0, exact=True)
    indptr, indices = g.rowptr.cpu().numpy(), g.col.cpu().numpy().astype(np.int64)
    deg = g.degrees().cpu()
    nnz_graph = g.nnz
    del g
    torch.manual_seed(args.seed)
    feats = torch.randn(args.mb_nodes, args.mb_feats).to(c.dtype)
    labels = torch.randint(0, args.mb_classes, (args.mb_nodes,))
    dg = DGraph.from_csr(indptr, indices, labels=labels, features=feats)
    cache = GraphCacheServer(feats, gpuid=c.dev.index or 0)
    cache.log = True
    if args.mb_cache_frac < 0:        # the reference's rule, storage.py:64-98: as many rows as fit in the free device memory (minus a reserve)
        cache.auto_cache(deg)
        cache_rule = "storage.py:64-98 capacity rule (free device memory / row bytes): %d of %d rows cached" % (
            min(cache.capability, args.mb_nodes), args.mb_nodes)
        args.mb_cache_frac = min(cache.capability, args.mb_nodes) / args.mb_nodes
    else:
        cache.auto_cache(deg, capacity=int(args.mb_cache_frac * args.mb_nodes))
        cache_rule = "fixed fraction %.2f of the nodes (hottest by out-degree)" % args.mb_cache_frac
    # The producers run up to six batches ahead (queue of 4 loaded + 2 sampled): a short timed region would be served from that
    # backlog and report the consumer's speed, not the pipeline's.  Steady state needs the backlog to be a small share of the batches
    # timed: at least 16 warm-up and 192 timed batches here (a 64-batch window, 0.15 s, moved by +-30 % from run to run), whatever --steps / --warmup say (the JSON line carries the counts used).
    args.warmup = max(args.warmup, 16)
    args.steps = max(args.steps, 192)
    # ... + a TAIL of 16 batches outside the timed window: the per-launch HIP events behind the launch tables cost the consumer thread
    # ~0.2 ms per batch (two event records per launch), so they are switched on for the tail only
    tail = 16
    n_batches = args.warmup + args.steps + tail
    train = torch.randperm(args.mb_nodes)[:n_batches * args.mb_batch]

    from dgll_amd import _lib as _dl

    _dl.check(_dl.lib.dgll_hip_debug_tune(12, int(args.mb_loader_blocks_per_cu)), "tune")
    if args.mb_dense_kernel == "4wave":
        _dl.check(_dl.lib.dgll_hip_debug_tune(4, 1), "tune")
    # native pool (the default): the workers never take the interpreter lock, so they scale with the cores -- 16 of them draw ~1 600
    # batches/s at the Reddit shape, well above what the GPU consumes; DGLL_NATIVE_SAMPLER_POOL=0: Python threads (8: more of those ran slower)
    pool_on = os.environ.get("DGLL_NATIVE_SAMPLER_POOL", "1") != "0"
    k_threads = args.mb_sampler_threads if args.mb_sampler_threads >= 0 else max(1, min(16 if pool_on else 8, (os.cpu_count() or 4) // 4))
    lock = __import__("threading").Lock()

    class TimedSampler(FastNeighborSampler):
        seconds, calls = 0.0, 0

        def _timed(self, fn, *a, **kw):
            t = time.perf_counter()
            out = fn(*a, **kw)
            dt = time.perf_counter() - t
            with lock:
                TimedSampler.seconds += dt
                TimedSampler.calls += 1
            return out

        def sample(self, g_, seeds):
            return self._timed(super().sample, g_, seeds)

        def sample_seeded(self, g_, seeds, seed, **kw):
            return self._timed(super().sample_seeded, g_, seeds, seed, **kw)

    loader = DataLoader(dg, train, TimedSampler(fanouts, defer_last_hop=True), batch_size=args.mb_batch)
    # the graph's index arrays in HBM (0.9 GB of int64): the outermost hop leaves the host as neighbour POSITIONS and becomes ids by a
    # device gather on the loading stream
    device_graph = None if args.mb_host_translate else (torch.from_numpy(indptr).to(c.dev), torch.from_numpy(indices).to(c.dev))
    # the outermost hop's mean is formed straight out of the feature cache (its 2.2 M gathered rows are never written): the loading
    # stage moves half the bytes, the consumer skips the widest block reduction, a queued batch is ten times smaller
    fuse_last = not args.mb_no_fused_last_hop
    pipe = MiniBatchPipeline(loader, cache=cache, labels=labels, queue_size=4, device=c.dev, hops="sampled",
                             reduce_last_hop="mean" if fuse_last else None, sampler_threads=k_threads, base_seed=args.seed, epoch=0,
                             device_graph=device_graph, build_blocks=True)
    model = dnn.GraphSage(args.mb_feats, [args.hidden] * (L - 1) + [args.mb_classes], fanouts).to(c.dev)
    from dgll_amd.optim import FlatAdam

    opt = torch.optim.Adam(model.parameters(), lr=1e-3) if args.torch_adam else FlatAdam(list(model.parameters()), lr=1e-3)
    graphed = None
    want_graph = args.mb_hip_graph == "on" or (args.mb_hip_graph == "auto" and fuse_last and not args.torch_adam)
    seen_rows = [0] * L          # the warm-up batches run launch by launch and show how far the hops fill their upper bounds
    graph_misses = in_place = 0

    def capture():
        # bounds of the captured step: what the warm-up batches reached + 10 % (a multiple of 64 rows), at most batch x prod(fan-outs);
        # a batch beyond them runs launch by launch (counted in the record).  Captured while the pipeline's threads keep working
        # (capture_error_mode = thread_local).
        from dgll_amd.graphs import GraphedSampledStep

        rows = [args.mb_batch] + [-(-int(r * 1.1) // 64) * 64 for r in seen_rows[1:]]
        try:
            # queue of 4 loaded batches + the one being consumed + the one being loaded = 6 in-place input sets (+ set 0, the copy path's)
            n_sets = 1 if args.mb_copy_inputs else 7
            step_ = GraphedSampledStep(model, opt, args.mb_batch, fanouts, args.mb_feats, args.mb_classes, dtype=c.dtype, device=c.dev, rows=rows,
                                       n_sets=n_sets)
            if n_sets > 1:
                pipe.use_static_sets(step_)       # batches loaded from now on land in a set; the ones already queued take the copy path
            return step_
        except Exception as exc:  # noqa: BLE001  (auto: a stack that cannot capture the step runs it launch by launch, and says so)
            if args.mb_hip_graph == "on":
                raise
            print("bench.py: HIP-graph capture of the sampled step failed (%s: %s); running launch by launch" % (type(exc).__name__, exc),
                  file=sys.stderr, flush=True)
            opt.zero_grad(set_to_none=True)
            return None
    random.seed(args.seed)
    timer_cm = ops.LaunchTimer()
    done = edges = 0
    gpu_ms = 0.0
    t0 = None
    events = []
    loss = None
    # the consumer's kernels go to a HIGH-priority stream: the loading stream's gathers (normal priority) overlap them and, left to
    # themselves, stretched half of the batches from 2.2 to 4.5-5 ms (round 3's six records)
    compute = torch.cuda.Stream(c.dev, priority=-1)
    compute.wait_stream(torch.cuda.current_stream(c.dev))
    stream_ctx = torch.cuda.stream(compute)
    stream_ctx.__enter__()
    cpu_busy = 0.0
    prof = None
    if os.environ.get("DGLL_MB_PROFILE"):      # diagnostics: where the consumer thread's host time goes (cProfile, printed to stderr)
        import cProfile

        prof = cProfile.Profile()
    for b in pipe:
        if prof is not None and done == args.warmup:
            prof.enable()
        if done == args.warmup:
            if want_graph:
                graphed = capture()
            torch.cuda.synchronize()
            gc.collect()
            gc.disable()                       # as in timed_steps: no full collection inside the timed window
            t0 = time.perf_counter()
            s0 = TimedSampler.seconds
            l0, cpu_busy = pipe.load_seconds, 0.0
        if done == args.warmup + args.steps:                  # the timed window ends here; the tail runs with per-launch events
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            gc.enable()
            sampler_s = TimedSampler.seconds - s0
            loader_s, busy_s = pipe.load_seconds - l0, cpu_busy
            if prof is not None:
                prof.disable()
            timer_cm.__enter__()
        t_body = time.perf_counter()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        blocks = b.blocks          # built by the loading stage on its own stream (the outermost one is None: reduced out of the cache)
        if blocks[L - 1] is None and b.last_hop_reduced is None:
            blocks[L - 1] = b.subgraphs[0].to_block(c.dev)
        in_place += b.static_set is not None and args.warmup <= done < args.warmup + args.steps
        use_graph = graphed is not None and done < args.warmup + args.steps
        if use_graph:
            try:
                loss = graphed(b)             # copies into the static inputs, one graph replay, the optimizer's launch
            except ValueError:                # a batch beyond the captured bounds
                use_graph = False
                graph_misses += 1
        if done < args.warmup:
            for h in range(L):
                seen_rows[h] = max(seen_rows[h], int(b.features[h].shape[0]) if b.features[h] is not None else 0)
        if not use_graph:                     # (also the tail after the timed window: launch by launch, it feeds the launch tables)
            with ranges.rng("consume"):       # (DGLL_PROFILE_RANGES=1: the reference's 'gpu-compute' range, FeatureCache/gs.py:93)
                if graphed is not None and b.static_set is not None:
                    loss = graphed.eager(b)   # a batch that was loaded in place: the same step on its static set, launch by launch
                else:
                    out = model.forward_sampled(b.features, blocks, last_hop_reduced=b.last_hop_reduced)
                    loss = ops.cross_entropy(out, b.labels)
                    opt.zero_grad(set_to_none=True)
                    loss.backward()
                    opt.step()
        ev1.record()
        if args.warmup <= done < args.warmup + args.steps:
            events.append((ev0, ev1))
            # edges aggregated per batch: layer l runs over hops 0..L-l-1, each a gather over that hop's sampled edges (fwd + bwd)
            per_hop = [b.subgraphs[L - 1 - h].num_src_nodes() for h in range(L)]
            edges += sum(sum(per_hop[:L - l]) for l in range(L)) * 2
        done += 1
        cpu_busy += time.perf_counter() - t_body
    torch.cuda.synchronize()
    if prof is not None:
        import pstats

        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
    stream_ctx.__exit__(None, None, None)
    timer_cm.__exit__(None, None, None)
    per_batch = sorted(a.elapsed_time(b_) for a, b_ in events)
    gpu_ms = sum(per_batch) / max(len(per_batch), 1)
    pct = lambda q: per_batch[min(len(per_batch) - 1, int(q * len(per_batch)))] if per_batch else None    # noqa: E731
    # sampled blocks differ in size from batch to batch: launches are grouped by kind (width, dtype, weights), their edges and
    # times summed; algorithmic bytes count the per-edge term only (the row counts of the blocks are not in the tags)
    torch.cuda.synchronize()
    groups, dense_groups = {}, {}
    for tag, a_, b_ in timer_cm.records:
        ms = a_.elapsed_time(b_)
        if tag[0] == "spmm":
            key = (tag[1], tag[2], bool(tag[3]), tag[5] if len(tag) > 5 else "")
            g_ = groups.setdefault(key, [0, 0.0, 0])
            g_[0] += 1; g_[1] += ms; g_[2] += tag[4]
        elif tag[0] in ("transform", "transform_dual", "grad_weight"):
            key = (tag[0], tag[2], tag[3], tag[4], tag[5])
            g_ = dense_groups.setdefault(key, [0, 0.0, 0])
            g_[0] += 1; g_[1] += ms; g_[2] += tag[1]
    table, dense_table = {}, {}
    for (feat, dt, weighted, extra), (cnt, ms, edges_) in groups.items():
        xb = 2 if "bfloat16" in dt else 4
        b_alg = edges_ * (feat * xb + 4 + (4 if weighted else 0))
        table["spmm F=%d %s %s%s (all block sizes)" % (feat, dt.replace("torch.", ""), "weighted" if weighted else "unweighted",
                                                      (" " + extra) if extra else "")] = {
            "count": cnt, "total_ms": ms, "avg_ms": ms / cnt, "nnz": edges_ // cnt, "edges_total": edges_, "feat": feat,
            "weighted": weighted, "epilogue": extra, "kernel_fragment": spmm_kernel_fragment(feat, dt, weighted, bool(extra)),
            "algorithmic_bytes": b_alg // cnt, "algorithmic_GBps": b_alg / (ms * 1e-3) / 1e9,
            "frac_algorithmic": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "G_edges_per_s": edges_ / (ms * 1e-3) / 1e9}
    for (kind, k1, k2, n_out, extra), (cnt, ms, rows_) in dense_groups.items():
        b_alg = rows_ * (k1 + k2 + n_out) * 2
        dense_table["%s K=%d%s N=%d%s (all block sizes)" % (kind, k1, ("+%d" % k2) if k2 else "", n_out, (" " + extra) if extra else "")] = {
            "count": cnt, "total_ms": ms, "avg_ms": ms / cnt, "rows_total": rows_, "algorithmic_GBps": b_alg / (ms * 1e-3) / 1e9}
    roofline = None
    if table:
        dom = table[max(table, key=lambda k: table[k]["total_ms"])]
        roofline = roofline_record(dom, workload_signature(args, nnz_graph, nodes=args.mb_nodes, batch=args.mb_batch),
                                   "the gather launch kind with the largest total time in the timed batches (%s), all block sizes pooled; "
                                   "these launches are short (tens of microseconds): launch-bound, not bandwidth-bound" % dom["kernel_fragment"],
                                   None, 1)
    steps = len(events)
    args.steps = steps
    result = base_result(
        args, c, edges / elapsed, elapsed,
        "BASELINE config 2 shape: sampled 3-layer GraphSAGE (hidden %d, %s) on a Reddit-sized synthetic graph (%d nodes, %d directed "
        "edges, F = %d, %d classes), batch %d, fan-out %s, bit-exact host sampler + %d %% hot-node HBM cache + mini-batch queue; "
        "a step = one batch" % (args.hidden, args.dtype, args.mb_nodes, nnz_graph, args.mb_feats, args.mb_classes, args.mb_batch,
                                args.mb_fanouts, int(args.mb_cache_frac * 100)),
        {"nodes": args.mb_nodes, "nnz": nnz_graph, "hidden": args.hidden, "batch": args.mb_batch, "fanouts": fanouts,
         "cache_fraction": args.mb_cache_frac, "outermost_hop": "reduced out of the cache" if fuse_last else "fetched, reduced in the model",
         "parallelism": "single GPU"})
    result.update({"loss": float(loss.detach()), "batches_per_s": steps / elapsed, "gpu_side_ms_per_batch": gpu_ms,
                   "gpu_side_ms_per_batch_p50": pct(0.5), "gpu_side_ms_per_batch_p95": pct(0.95), "gpu_side_ms_per_batch_max": per_batch[-1],
                   # per batch and sampler thread: the Python-thread path times sample_seeded around its native call; the native pool reports
                   # its workers' own time per batch (whole epoch)
                   "host_sampler_ms_per_batch": (getattr(pipe, "pool_stats", None) or {}).get("sample_ms_per_batch", sampler_s / max(steps, 1) * 1e3),
                   "host_sampler_threads": k_threads,
                   "native_sampler_pool": getattr(pipe, "pool_stats", None) is not None,
                   # where the batch period goes on the host: the consumer thread's own time per batch (issuing ~110 launches through
                   # Python), the loading thread's, and what is left of the period = the consumer waiting for a batch
                   "consumer_host_ms_per_batch": busy_s / max(steps, 1) * 1e3,
                   "loader_host_ms_per_batch": loader_s / max(steps, 1) * 1e3,
                   "launch_tables_from": "%d further batches after the timed window (per-launch HIP events on)" % tail,
                   "loaded_queue_starved_s": pipe.queue._starved,
                   "sampler_mode": ("per-batch seeds, %d %s (batch b under random.seed(batch_seed(%d, 0, b)))" % (
                       k_threads, "native pool threads (csrc/sampler.hip: no interpreter in the producers)" if getattr(pipe, "pool_stats", None)
                       else "Python threads around the native draw", args.seed)) if k_threads > 0 else "one sequential stream on the interpreter's generator",
                   "outermost_hop_translation": "host" if device_graph is None else "device gather from pinned positions",
                   "consumer_step": ("one HIP graph on padded static block shapes (rows per hop %s, %d of %d timed batches beyond them ran "
                                     "launch by launch) + the optimizer's launch; %d of the timed batches were written IN PLACE into one of %d "
                                     "static input sets by the loading stage (no copy into the graph's inputs)" % (
                                         graphed.rows, graph_misses, steps, in_place, len(graphed.sets) - 1)) if graphed is not None
                   else "launch by launch",
                   "batches_loaded_in_place": in_place,
                   "cache_miss_rate": cache.get_miss_rate(), "cache_rule": cache_rule,
                   "cache_rows": int(min(cache.capability, args.mb_nodes)),
                   "epoch_time_s_153431_train_nodes": elapsed / steps * (153_431 / args.mb_batch),
                   "roofline": roofline, "spmm_launch_table": table, "dense_launch_table": dense_table})
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_sampler(indptr, indices, fanouts, args.mb_batch, args.seed)
    return result


def cpu_baseline_sampler(indptr, indices, fanouts, batch, seed):
    """Config 2's host path timed alone: the reference's pure-Python sampler loop (base_sampler.py:30-58: random.sample per
    seed, hops in reversed(fanouts) order) as restated in oracle/sampler.py, on a few batches, against which the native
    bit-exact sampler of the pipeline is reported."""
    import random

    import numpy as np

    from oracle import sampler as osampler

    host = host_info()
    rng = np.random.default_rng(seed)
    seeds = rng.integers(0, len(indptr) - 1, size=batch).tolist()

    class Adj:          # adjacency lists on demand (the reference holds python lists per node, dgraph.py:49-62)
        def __getitem__(self, v):
            return indices[indptr[v]:indptr[v + 1]].tolist()

    random.seed(seed)
    t0 = time.perf_counter()
    _, _, layers = osampler.sample(Adj(), seeds, fanouts)
    dt = time.perf_counter() - t0
    n_edges = sum(len(src) for src, _ in layers)
    return {"value": n_edges / dt, "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": "one batch of %d seeds, fan-out %s, through the reference's sampler loop (base_sampler.py:30-58, dgllsampler.py:"
                      "10-21) restated in oracle/sampler.py: %d sampled edges in %.2f s, single Python thread (the reference's own "
                      "CPU path of config 2 is this host loop; the GPU-side figures are in the line's other fields)" % (
                          batch, fanouts, n_edges, dt),
            "cpu_model": host["cpu_model"], "threads": 1, "physical_cores": host["physical_cores"]}


def per_rank_diagnostics(c, engine, step, racom=None, opt_wrap=None):
    """N > 1, after the timed steps: what every rank did per step -- its own wall time, how long its compute stream sat waiting for
    an exchange (the EXPOSED part of the halo traffic), bytes it sent, kernels it ran.  Gathered to rank 0 for the line, so that the
    first run on real links says where the time went."""
    import torch.distributed as dist
    from torch.profiler import ProfilerActivity, profile

    reps = 3
    engine.exchange.bytes_sent = engine.exchange.bytes_received = engine.exchange.calls = 0
    engine.wait_events = []
    if racom is not None:
        racom.bytes_reduced = 0
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / reps * 1e3
    reduced = (racom.bytes_reduced / reps) if racom is not None else 0
    wait_ms = engine.exposed_wait_ms() / reps
    engine.wait_events = None
    sent, calls = engine.exchange.bytes_sent / reps, engine.exchange.calls / reps
    dist.barrier()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    kern = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    p = engine.part
    mine = {"rank": c.rank, "step_ms": wall_ms, "exchange_wait_ms": wait_ms, "compute_ms": wall_ms - wait_ms,
            "bytes_sent_per_step": sent, "exchanges_per_step": calls, "kernels_per_step": len(kern),
            "kernel_ms_per_step": sum(e.time_range.end - e.time_range.start for e in kern) / 1e3,
            "own_rows": p.n_own, "halo_rows": p.n_halo, "local_edges": p.local.nnz, "halo_edges": p.halo.nnz,
            "gradient_bytes_all_reduced_per_step": reduced}
    everyone = [None] * c.world
    dist.all_gather_object(everyone, mine)
    return everyone


OTHER_WORKLOADS = {        # record name -> (--workload, child command line tail, wall-clock bound in seconds)
    "gat": ("gat", ["--steps", "10", "--warmup", "3", "--no-extra-graphs"], 420),
    "minibatch": ("minibatch", [], 420),
    "rmat27": ("rmat27", ["--steps", "5", "--warmup", "2"], 600),
    # the headline step in the reference's own arithmetic (fp32 storage, dgll/__init__.py:1, gcnconv.py:30-31): BASELINE.md section 3
    # quotes a 4.65 G edges/s fp32-weighted target next to the bf16 one
    "sage_f32": ("sage", ["--dtype", "f32", "--steps", "10", "--warmup", "3", "--no-extra-graphs", "--no-cpu-baseline", "--other-workloads", "off"], 300),
    # config 2 with the reference's own capacity rule for the feature cache (storage.py:64-98: everything that fits in free device
    # memory -- all of a Reddit-sized matrix in 288 GB) next to the 50 % cache the `minibatch` record keeps from the earlier rounds
    "minibatch_capacity_rule_cache": ("minibatch", ["--mb-cache-frac", "-1", "--no-cpu-baseline"], 420),
}


def compact_record(d):
    """What the default line keeps of a child run: the contract fields, `roofline` and `cpu_baseline` trimmed to their contract keys
    (+ the kernel and the frac definitions), and the workload's own headline figures.  `bench.py --workload X` prints the full record."""
    keep = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "data", "epoch_time_s", "loss")}
    keep["workload"] = d.get("config", {}).get("workload")
    r = d.get("roofline") or {}
    keep["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_definition", "frac_algorithmic",
                                              "frac_l2_miss_path", "traffic_over_compulsory", "kernel", "kernel_fragment", "avg_launch_ms", "algorithmic_bytes_per_launch", "build_stamp")} if r else None
    cb = d.get("cpu_baseline") or {}
    keep["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "oracle_c_openmp_csr_edges_per_s",
                                                   "oracle_c_openmp_edges_per_s", "cpu_model")} if cb else None
    for k in ("gat_pass_over_spmm", "spmm_same_width_same_graph_ms", "batches_per_s", "gpu_side_ms_per_batch", "gpu_side_ms_per_batch_p50",
              "gpu_side_ms_per_batch_p95", "host_sampler_ms_per_batch", "host_sampler_threads", "sampler_mode", "consumer_host_ms_per_batch",
              "loader_host_ms_per_batch", "cache_miss_rate", "epoch_time_s_153431_train_nodes", "cache_rows", "cache_rule",
              "aggregate_first_ms_per_step"):
        if k in d:
            keep[k] = d[k]
    cfg = d.get("config", {})
    for k in ("nodes", "nnz", "hidden", "heads", "batch", "fanouts", "scale", "peak_memory_GB", "cache_fraction"):
        if k in cfg:
            keep.setdefault("config", {})[k] = cfg[k]
    launches = {name: round(v["avg_ms"], 4) for name, v in (d.get("spmm_launch_table") or {}).items()}
    if launches:
        keep["gather_launch_ms"] = launches
    return keep


def run_other_workloads(args):
    """The other BASELINE configs, each in a CHILD process (`bench.py --workload X`, started after this process has finished its own
    measurement and released its device memory), one after the other; a child that fails or overruns leaves an `error` record and
    never takes the headline with it."""
    torch.cuda.empty_cache()
    overrides = json.loads(os.environ.get("DGLL_BENCH_OTHER_ARGS", "{}"))      # tests: small shapes
    out = {}
    for name, (workload, tail, limit) in OTHER_WORKLOADS.items():
        extra = overrides.get(name, tail if not overrides else None)      # tests name the records they want (small shapes)
        if extra is None:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--seed", str(args.seed), "--full-line"] + list(extra)
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=limit)
            line = next((ln for ln in reversed(res.stdout.splitlines()) if ln.startswith("{") and '"metric"' in ln), None)
            if res.returncode != 0 or line is None:
                out[name] = {"error": "exit code %d" % res.returncode, "stderr_tail": res.stderr[-600:]}
            else:
                out[name] = compact_record(json.loads(line))
        except subprocess.TimeoutExpired:
            out[name] = {"error": "no result within %d s" % limit}
        out[name]["wall_seconds"] = time.perf_counter() - t0
        out[name]["command"] = "python bench.py --workload %s %s" % (workload, " ".join(extra))
    return out


WORKLOADS = {"sage": run_sage, "gat": run_gat, "rmat27": run_rmat27, "minibatch": run_minibatch}


def _sig(x, digits=6):
    """Numbers of the compact line carry `digits` significant figures (the full record keeps every digit)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    return x


def _short(text, limit):
    text = " ".join(str(text).split())
    return text if len(text) <= limit else text[:limit - 3] + "..."


def compact_roofline(r):
    """`roofline` of the compact line: numbers + two short strings.  frac_kind = "counter" when `frac` comes from the counter traffic
    of profiles/traffic.json (same build), "algorithmic_capped" when no entry matched and frac = min(1, frac_algorithmic)."""
    if not r:
        return None
    out = {k: _sig(r.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "traffic", "traffic_over_compulsory",
                                       "avg_launch_ms", "algorithmic_bytes_per_launch")}
    out["kernel_fragment"] = _short(r.get("kernel_fragment"), 96)
    out["build_stamp"] = (r.get("build_stamp") or "")[:16] or None
    out["frac_kind"] = "counter" if r.get("traffic") is not None and r.get("frac_l2_miss_path") is not None else "algorithmic_capped"
    if r.get("frac_conservative") is not None:
        out["frac_conservative"] = _sig(r["frac_conservative"])
    return out


def compact_cpu_baseline(cb):
    if not cb:
        return None
    out = {k: _sig(cb.get(k)) for k in ("value", "unit", "cores", "threads", "kind")}
    out["cpu_model"] = _short(cb.get("cpu_model"), 48)
    out["sample"] = _short(cb.get("sample"), 110)
    for k in ("torch_sparse_mm_csr_edges_per_s", "oracle_c_openmp_csr_edges_per_s", "oracle_c_openmp_edges_per_s"):
        if cb.get(k) is not None:
            out[k] = _sig(cb[k])
    return out


def compact_other(rec):
    """At most ten numeric keys of a child record (the full child record is in bench_full.json)."""
    if "error" in rec:
        return {"error": _short(rec["error"], 80), "wall_seconds": _sig(rec.get("wall_seconds"), 4)}
    r = rec.get("roofline") or {}
    cb = rec.get("cpu_baseline") or {}
    out = {"ms_per_step": rec.get("ms_per_step"), "value": rec.get("value"), "roofline_frac": r.get("frac"),
           "roofline_frac_algorithmic": r.get("frac_algorithmic"), "cpu_baseline_value": cb.get("value")}
    if "batches_per_s" in rec:
        out.update({k: rec.get(k) for k in ("batches_per_s", "gpu_side_ms_per_batch", "gpu_side_ms_per_batch_p95",
                                            "loader_host_ms_per_batch", "cache_miss_rate")})
    passes = {}
    for name, ms in (rec.get("gather_launch_ms") or {}).items():       # the three GAT passes of the hidden (8-head) layer
        for kind in ("fwd", "bwd_rows", "bwd_cols"):
            if name.startswith("gat %s " % kind) and " packed" not in name and kind not in passes:
                passes[kind] = ms
    for kind, ms in passes.items():
        out["gat_%s_ms" % kind] = ms
    if "aggregate_first_ms_per_step" in rec:
        out["aggregate_first_ms_per_step"] = rec["aggregate_first_ms_per_step"]
    out["wall_seconds"] = rec.get("wall_seconds")
    return {k: _sig(v, 5) for k, v in list(out.items())[:10] if v is not None}


def compact_line(result, full_path=None):
    """The ONE stdout line: the contract fields, `roofline` and `cpu_baseline` as numbers + short strings, `other_workloads` as at most
    ten numbers per config -- at most LINE_LIMIT characters.  Definitions, `sample` prose, launch tables, per-step times, the extra-graph
    rooflines, per-rank diagnostics and the child commands stay in the full record (bench_full.json / stderr / --full-line)."""
    cfg = result.get("config", {})
    line = {k: _sig(result.get(k), 10) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                 "scaling", "vs_baseline", "dtype", "data")}
    keep_cfg = {"workload": _short(cfg.get("workload"), 260)}
    for k in ("workload_id", "nodes", "nnz", "hidden", "heads", "scale", "batch", "fanouts", "locality", "permuted_ids", "reorder",
              "parallelism", "spmm_launches_per_step", "gradient_sharing", "ranks", "backend", "exchange_form"):
        if cfg.get(k) is not None:
            keep_cfg[k] = _sig(cfg[k])
    if isinstance(cfg.get("halo_mode"), dict):
        keep_cfg["halo_mode"] = cfg["halo_mode"].get("mode")
    line["config"] = keep_cfg
    for k in ("epoch_time_s", "loss", "batches_per_s", "gpu_side_ms_per_batch", "gpu_side_ms_per_batch_p95", "loader_host_ms_per_batch",
              "cache_miss_rate", "aggregate_first_ms_per_step", "allgather_ms"):
        if result.get(k) is not None:
            line[k] = _sig(result[k])
    line["roofline"] = compact_roofline(result.get("roofline"))
    if "cpu_baseline" in result:
        line["cpu_baseline"] = compact_cpu_baseline(result["cpu_baseline"])
    gather = {_short(name, 60): _sig(v.get("avg_ms"), 4) for name, v in (result.get("spmm_launch_table") or {}).items()}
    dense = {_short(name, 60): [_sig(v.get("avg_ms"), 4), _sig(v.get("frac_of_hbm_peak"), 3)]
             for name, v in (result.get("dense_launch_table") or {}).items()}
    if gather:
        line["gather_launch_ms"] = gather
    if dense:
        line["dense_launch_ms_and_frac_of_hbm_peak"] = dense
    if result.get("other_workloads"):
        line["other_workloads"] = {name: compact_other(rec) for name, rec in result["other_workloads"].items()}
    line["full_record"] = os.path.basename(full_path) if full_path else None
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("dense_launch_ms_and_frac_of_hbm_peak", "gather_launch_ms"):      # never over the limit: the optional tables go first
        if len(text) > LINE_LIMIT and drop in line:
            del line[drop]
            text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        line["config"]["workload"] = _short(line["config"]["workload"], 80)
        line.pop("other_workloads", None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def write_full_record(result):
    """The full record next to bench.py (bench_full.json; DGLL_BENCH_FULL_RECORD overrides the path); None when the directory is read-only."""
    path = os.environ.get("DGLL_BENCH_FULL_RECORD", FULL_RECORD)
    try:
        with open(path + ".tmp", "w") as f:
            json.dump(result, f, indent=1)
        os.replace(path + ".tmp", path)
        return path
    except OSError:
        return None


def main():
    args = parse_args()
    if os.environ.get("DGLL_BENCH_DUMP_AFTER"):       # diagnostics: every thread's Python stack after that many seconds (a rank that hangs)
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["DGLL_BENCH_DUMP_AFTER"]), exit=False)
    c = setup(args)
    import dgll_amd  # noqa: F401  (loads libdgll_hip.so; raises if it cannot be built)

    result = WORKLOADS[args.workload](args, c)
    if c.rank == 0 and result is not None:
        default_shape = (args.workload == "sage" and c.world == 1 and not args.dataset and args.nodes == 2_449_029
                         and args.undirected_edges == 61_859_140 and args.dtype == "bf16")
        if args.other_workloads == "on" or (args.other_workloads == "auto" and default_shape):
            result["other_workloads"] = run_other_workloads(args)
        if args.full_line:
            print(json.dumps(result), flush=True)
            return
        full_path = write_full_record(result)
        sys.stderr.write("bench.py full record%s:\n%s\n" % ((" (also in %s)" % full_path) if full_path else "", json.dumps(result)))
        sys.stderr.flush()
        print(compact_line(result, full_path), flush=True)


if __name__ == "__main__":
    main()
