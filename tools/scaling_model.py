#!/usr/bin/env python3
"""What ONE rank of the N-GPU bench computes and exchanges per step, measured on one GPU.

For N in 2, 4, 8 the bench graph is partitioned exactly as bench.py does; rank 0's engine runs real training steps with a
stand-in transport that moves nothing (received buffers keep whatever they hold): the kernels, launch counts and buffer
sizes are those of the real run, only the wire time is missing.  Printed: the rank's compute per step, the bytes it
receives and sends per step, and the step time / speed-up that follows for a given per-GPU exchange bandwidth under two
bounds -- exchange fully hidden behind compute, and not hidden at all.  (xGMI: 7 links x ~64 GB/s per direction per
GPU at the guide's 153 GB/s bidirectional per link; RCCL's point-to-point efficiency decides where in between it lands.)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dist as ddist, nn as dnn, ops, partition as dpart, reorder as dreorder, synth  # noqa: E402
from dgll_amd.optim import FlatAdam  # noqa: E402


class NullExchange:
    """Counts the bytes a grouped send/recv would move; moves nothing."""

    def __init__(self, part):
        self.part, self.group = part, None
        self.sent = self.received = 0

    def start(self, send_buf, recv_buf, reverse=False, more=()):
        for sb, rb in ((send_buf, recv_buf),) + tuple(more):
            self.sent += sb.numel() * sb.element_size()
            self.received += rb.numel() * rb.element_size()
        return ([], [])

    @staticmethod
    def wait(handle):
        return None


RANK = int(os.environ.get("MODEL_RANK", "0"))      # which rank's share is run (0 holds the hub communities: the heaviest)


def rmat27_model(scale, feat=128, dtype=torch.bfloat16, reps=5):
    """Config 5 at N = 1, 2, 4, 8 on ONE GPU: the step has no exchange (row blocks, X replicated: dist.RowBlockShard), so a rank's
    step time on its own GPU is its block's time here.  EVERY block of every N is run through the real kernels (the bench's
    aggregate-first step: SpMM over the block against the full X, transform of the block's rows); predicted step = the slowest
    block.  Then the per-edge / per-row costs are FITTED to the measured block times (least squares over all blocks of all N) and
    the cut points recomputed with them: what the byte-count guess leaves on the table."""
    from dgll_amd import dense

    dev = torch.device("cuda:0")
    g = synth.rmat_graph(scale, 16, seed=0, device=dev, symmetric=False, weighted=False, self_loops=True)
    g, _ = g.reorder(method="degree" if g.nnz >= (1 << 30) else "lpa", seed=0)
    torch.cuda.empty_cache()
    n, esz = g.n_rows, 2 if dtype == torch.bfloat16 else 4
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    x = ops.alloc_features(n, feat, dtype, dev)
    for lo in range(0, n, 1 << 24):
        hi = min(n, lo + (1 << 24))
        x[lo:hi] = torch.randn(hi - lo, feat, device=dev, generator=gen).to(dtype)
    w = (torch.randn(feat, feat, device=dev, generator=gen) / feat ** 0.5).to(dtype)
    print("RMAT-%d: %d nodes, %d nonzeros, max degree %d, F = %d %s" % (scale, n, g.nnz, int(g.degrees().max()), feat, dtype))

    def run_block(blk):
        y = ops.alloc_features(blk.n_rows, feat, dtype, dev)

        def step():
            return dense.transform_bf16(ops.spmm_raw(blk, x, reduce="mean", out=y), w.t()) if dtype == torch.bfloat16 \
                else ops.spmm_raw(blk, x, reduce="mean", out=y) @ w

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        with ops.LaunchTimer() as timer:
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
        kinds = {k[0]: v[1] for k, v in timer.summary().items()}
        return ms, kinds.get("spmm", 0.0), kinds.get("transform", 0.0)

    def sweep(edge_cost, row_cost, label):
        samples, single = [], None
        for world in (1, 2, 4, 8):
            bounds = ddist.cost_balanced_bounds(g.rowptr, world, edge_cost, row_cost)
            rows_out = []
            for r in range(world):
                shard = ddist.RowBlockShard(g, world, r, bounds=bounds)
                if world > 1:
                    shard.own_copy()
                else:
                    shard.block = g
                    g.plan()
                ms, spmm_ms, dense_ms = run_block(shard.block)
                rows_out.append((r, shard.n_own, shard.block.nnz, ms, spmm_ms, dense_ms))
                samples.append((shard.block.nnz, shard.n_own, ms))
                del shard
                torch.cuda.empty_cache()
            slow = max(rows_out, key=lambda t: t[3])
            mean = sum(t[3] for t in rows_out) / world
            if world == 1:
                single = slow[3]
            print("[%s] N=%d: predicted step %.2f ms (slowest block: rank %d), mean %.2f, max/mean %.3f, speed-up vs N=1 %.2fx" % (
                label, world, slow[3], slow[0], mean, slow[3] / mean, single / slow[3]))
            for r, rows, nnz, ms, sp, de in rows_out:
                print("      rank %d: %11d rows %12d nnz | step %.2f ms = spmm %.2f + transform %.2f | %.1f G edges/s, spmm %.2f TB/s algorithmic" % (
                    r, rows, nnz, ms, sp, de, nnz / (ms * 1e-3) / 1e9,
                    (nnz * (feat * esz + 4) + rows * (feat * esz + 8)) / max(sp, 1e-9) / 1e9))
        return samples

    edge_cost, row_cost = feat * esz + 4, feat * esz + 8 + 2 * feat * esz
    samples = sweep(edge_cost, row_cost, "byte costs: %d B/edge, %d B/row" % (edge_cost, row_cost))
    a = torch.tensor([[s[0], s[1]] for s in samples], dtype=torch.float64)
    b = torch.tensor([s[2] for s in samples], dtype=torch.float64).unsqueeze(1)
    sol = torch.linalg.lstsq(a, b).solution.flatten()
    per_edge_ns, per_row_ns = float(sol[0]) * 1e6, float(sol[1]) * 1e6
    print("fitted: %.4f ns per edge, %.4f ns per row (ratio row/edge %.2f; the byte costs assume %.2f)" % (
        per_edge_ns, per_row_ns, per_row_ns / per_edge_ns, row_cost / edge_cost))
    if per_edge_ns > 0 and per_row_ns > 0:
        sweep(max(1, int(round(per_edge_ns * 1000))), max(1, int(round(per_row_ns * 1000))), "fitted costs")



def gat_model(worlds=(2, 4), heads=8, fo=32, classes=47, in_feats=100):
    """Config 4 ("2-layer GAT (8 heads) on ogbn-products, 1 -> 4 GPUs"): EVERY rank of N = 2, 4 of bench.py's `--workload gat` step
    through its real kernels on one GPU with the stand-in transport -- the partitioned SpGAT (DistGraph.spgat_forward: first layer's
    transform and scores evaluated on the placed halo rows, nothing exchanged for it; the out head exchanges its 47-wide rows and
    scores), the fused cross-entropy, RaCoM's flat buffer, Adam.  MODEL_HALO_MODES=recompute,exchange runs both forms."""
    dev = torch.device("cuda:0")
    raw = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True, self_loops=True)     # bench.py --workload gat
    n, nnz = raw.n_rows, raw.nnz
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    feats = torch.randn(n, in_feats, generator=gen, device=dev)
    labels_all = torch.randint(0, classes, (n,), generator=gen, device=dev)

    def timed(step, reps=10, warm=3):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def single_gpu_step_ms():
        torch.manual_seed(0)
        model = dnn.SpGAT(in_feats, fo, classes, dropout=0.0, alpha=0.2, nheads=heads).to(dev)
        opt = FlatAdam(list(model.parameters()), lr=1e-3)
        full, perm = raw.reorder(seed=0)
        full.plan(); full.transpose()[0].plan()
        x = ops.alloc_features(n, in_feats, torch.bfloat16, dev, pad_to=64)
        x.copy_(feats[perm])
        lab = labels_all[perm]

        def step():
            opt.zero_grad(set_to_none=True)
            loss = ops.cross_entropy(model.forward_activations(x, full), lab, reduction="sum") * (1.0 / n)
            loss.backward()
            opt.step()

        return timed(step)

    single_ms = float(os.environ["SINGLE_MS"]) if "SINGLE_MS" in os.environ else single_gpu_step_ms()
    torch.cuda.empty_cache()
    print("config 4, single-GPU SpGAT step (%d heads x %d -> %d; %d nodes, nnz %d with self-loops; measured in this run): %.2f ms" % (
        heads, fo, classes, n, nnz, single_ms))
    halo_modes = os.environ.get("MODEL_HALO_MODES", "recompute").split(",")
    summary = []
    for world in worlds:
        stats = {}
        perm, bounds = dpart.partition_and_order(raw, world, seed=0, stats=stats)
        full = dreorder.relabel(raw, perm)
        q = stats.get("balanced", stats["after"])
        print("N=%d partition: cut %.2f %%, edges max/mean %.3f" % (world, 100 * q["cut"], q["balance"]))
        feats_p, labels_p = feats[perm], labels_all[perm]
        ranks = list(range(world)) if "MODEL_RANK" not in os.environ else [RANK % world]
        for mode in halo_modes:
            per_rank = []
            for rank in ranks:
                torch.manual_seed(0)
                model = dnn.SpGAT(in_feats, fo, classes, dropout=0.0, alpha=0.2, nheads=heads).to(dev)
                params = list(model.parameters())
                opt = FlatAdam(params, lr=1e-3)
                racom = ddist.RaCoM(params, dev, flat=opt)
                part = ddist.partition_contiguous(full, world, rank, bounds)
                engine = ddist.DistGraph(part, dev)
                engine.halo_recompute = mode == "recompute"
                engine.exchange = NullExchange(part)
                x = ops.alloc_features(part.n_own, in_feats, torch.bfloat16, dev, pad_to=64)
                x.copy_(feats_p[part.own_begin:part.own_end])
                labels = labels_p[part.own_begin:part.own_end]
                placed = engine.place_input_halo(x)

                def step():
                    opt.zero_grad(set_to_none=True)
                    act = engine.spgat_forward(model, x, placed, activations=True)
                    loss = ops.cross_entropy(act, labels, reduction="sum") * (world / n)
                    loss.backward()
                    racom.all_reduce_and_wait()
                    opt.step()

                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                engine.exchange.sent = engine.exchange.received = 0
                reps = 10
                with ops.LaunchTimer() as timer:
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        step()
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / reps * 1e3
                gather_ms = sum(v[1] * v[0] for k, v in timer.summary().items() if k[0] in ("gat", "spmm")) / reps
                rx, tx = engine.exchange.received / reps, engine.exchange.sent / reps
                per_rank.append((rank, ms, rx, tx))
                print("   N=%d [%s] rank %d: %7d rows, %9d + %8d (halo) edges, %7d halo rows | compute %.2f ms/step (gather passes %.2f) | "
                      "receives %.0f MB, sends %.0f MB per step" % (world, mode, rank, part.n_own, part.local.nnz, part.halo.nnz, part.n_halo,
                                                                     ms, gather_ms, rx / 1e6, tx / 1e6))
                del engine, part, model, opt, racom, placed, x
                torch.cuda.empty_cache()
            slow = max(per_rank, key=lambda t: t[1])
            mean = sum(t[1] for t in per_rank) / len(per_rank)
            rx, tx = max(t[2] for t in per_rank), max(t[3] for t in per_rank)
            print("N=%d [%s]: slowest rank %d %.2f ms, mean %.2f ms, spread %+.1f %% / %+.1f %% of the mean; ideal %.2f ms" % (
                world, mode, slow[0], slow[1], mean, 100 * (slow[1] / mean - 1), 100 * (min(t[1] for t in per_rank) / mean - 1), single_ms / world))
            for bw in (150e9, 300e9, 450e9):
                wire = max(rx, tx) / bw * 1e3
                print("      at %3.0f GB/s per direction: wire %.2f ms -> step %.2f (hidden) .. %.2f ms (exposed): speed-up %.2fx .. %.2fx" % (
                    bw / 1e9, wire, max(slow[1], wire), slow[1] + wire, single_ms / max(slow[1], wire), single_ms / (slow[1] + wire)))
            summary.append((world, mode, slow[1], mean, single_ms / slow[1]))
        del full
        torch.cuda.empty_cache()
    print("summary, config 4 (single-GPU step %.2f ms):" % single_ms)
    for world, mode, slow, mean, sp in summary:
        print("   N=%d %-9s slowest rank %.2f ms (mean %.2f): %.2fx with the exchange hidden" % (world, mode, slow, mean, sp))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "rmat27":
        return rmat27_model(int(sys.argv[2]) if len(sys.argv) > 2 else 27)
    if len(sys.argv) > 1 and sys.argv[1] == "gat":
        return gat_model(tuple(int(a) for a in sys.argv[2:]) or (2, 4))
    dev = torch.device("cuda:0")
    raw = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True)     # bench.py's default graph
    n, nnz = raw.n_rows, raw.nnz
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    feats = torch.randn(n, 100, generator=gen, device=dev)
    labels_all = torch.randint(0, 47, (n,), generator=gen, device=dev)
    # the single-GPU step of bench.py, measured here and now on the same box (the engine's reorder, same model, same loss)
    def single_gpu_step_ms():
        torch.manual_seed(0)
        model = dnn.GraphSage(100, [256, 256, 47], None).to(dev)
        opt = FlatAdam(list(model.parameters()), lr=1e-3)
        full, perm = raw.reorder(seed=0)
        full.plan(); full.transpose()[0].plan(); full.mean_scale_transposed()
        x = ops.alloc_features(n, 100, torch.bfloat16, dev, pad_to=64)
        x.copy_(feats[perm])
        lab = labels_all[perm]

        def step():
            opt.zero_grad(set_to_none=True)
            loss = ops.cross_entropy(model.forward_graph(full, x), lab, reduction="sum", fold_relu=True) * (1.0 / n)
            loss.backward()
            opt.step()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3

    single_ms = float(os.environ["SINGLE_MS"]) if "SINGLE_MS" in os.environ else single_gpu_step_ms()
    torch.cuda.empty_cache()
    print("single-GPU step (same model and graph, measured in this run): %.2f ms" % single_ms)
    samples = []
    balance = os.environ.get("MODEL_BALANCE", "cost")            # cost: partition.rebalance_parts (default); edges: round 4's equal-edge parts
    halo_modes = os.environ.get("MODEL_HALO_MODES", "recompute").split(",")
    summary = []
    for world in (2, 4, 8):
        stats = {}
        perm, bounds = dpart.partition_and_order(raw, world, seed=0, stats=stats, balance_cost=(balance == "cost"))   # as bench.py
        full = dreorder.relabel(raw, perm)
        q = stats.get("balanced", stats["after"])
        print("N=%d partition (%s-balanced): cut %.2f %% (after refinement %.2f %%), edges max/mean %.3f%s" % (
            world, balance, 100 * q["cut"], 100 * stats["after"]["cut"], q["balance"],
            (", %d rebalancing passes: modelled cost max/mean %.3f -> %.3f" % (
                len(stats["rebalance"]), stats["rebalance"][0]["max_over_mean"], stats["rebalance"][-1]["max_over_mean"])) if "rebalance" in stats else ""))
        ranks = list(range(world)) if "MODEL_RANK" not in os.environ else [RANK % world]
        feats_p, labels_p = feats[perm], labels_all[perm]
        for mode in halo_modes:
            per_rank = []
            for rank in ranks:
                torch.manual_seed(0)
                model = dnn.GraphSage(100, [256, 256, 47], None).to(dev)
                params = list(model.parameters())
                opt = FlatAdam(params, lr=1e-3)
                racom = ddist.RaCoM(params, dev, flat=opt)       # as bench.py; no process group here: the all-reduce itself is skipped
                part = ddist.partition_contiguous(full, world, rank, bounds)
                engine = ddist.DistGraph(part, dev)
                engine.halo_recompute = mode == "recompute"
                engine.exchange = NullExchange(part)
                x = ops.alloc_features(part.n_own, 100, torch.bfloat16, dev, pad_to=64)
                x.copy_(feats_p[part.own_begin:part.own_end])
                labels = labels_p[part.own_begin:part.own_end]
                placed = engine.place_input_halo(x)

                def step():
                    opt.zero_grad(set_to_none=True)
                    out = engine.sage_forward(model, x, placed)
                    loss = ops.cross_entropy(out, labels, reduction="sum", fold_relu=True) * (world / n)
                    loss.backward()
                    racom.all_reduce_and_wait()
                    opt.step()

                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                engine.exchange.sent = engine.exchange.received = 0
                t0 = time.perf_counter()
                reps = 10
                for _ in range(reps):
                    step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / reps * 1e3
                rx, tx = engine.exchange.received / reps, engine.exchange.sent / reps
                per_rank.append((rank, ms, rx, tx, part.n_own, part.local.nnz + part.halo.nnz, part.n_halo))
                if mode == "recompute":
                    samples.append((part.local.nnz + part.halo.nnz, part.n_halo, part.n_own, ms))
                print("   N=%d [%s] rank %d: %7d rows, %9d + %8d (halo) edges, %7d halo rows | compute %.2f ms/step | receives %.0f MB, sends %.0f MB per step" % (
                    world, mode, rank, part.n_own, part.local.nnz, part.halo.nnz, part.n_halo, ms, rx / 1e6, tx / 1e6))
                del engine, part, model, opt, racom, placed, x
                torch.cuda.empty_cache()
            slow = max(per_rank, key=lambda t: t[1])
            mean = sum(t[1] for t in per_rank) / len(per_rank)
            rx, tx = max(t[2] for t in per_rank), max(t[3] for t in per_rank)
            print("N=%d [%s]: slowest rank %d %.2f ms, mean %.2f ms, spread %+.1f %% / %+.1f %% of the mean" % (
                world, mode, slow[0], slow[1], mean, 100 * (slow[1] / mean - 1), 100 * (min(t[1] for t in per_rank) / mean - 1)))
            for bw in (150e9, 300e9, 450e9):
                wire = max(rx, tx) / bw * 1e3
                print("      at %3.0f GB/s per direction: wire %.2f ms -> step %.2f (hidden) .. %.2f ms (exposed): speed-up %.2fx .. %.2fx" % (
                    bw / 1e9, wire, max(slow[1], wire), slow[1] + wire, single_ms / max(slow[1], wire), single_ms / (slow[1] + wire)))
            summary.append((world, mode, slow[1], mean, single_ms / slow[1]))
        del full
        torch.cuda.empty_cache()
    if len(samples) >= 4:
        a_ = torch.tensor([[s_[0], s_[1], s_[2], 1.0] for s_ in samples], dtype=torch.float64)
        b_ = torch.tensor([[s_[3]] for s_ in samples], dtype=torch.float64)
        sol = torch.linalg.lstsq(a_, b_).solution.flatten()
        resid = (a_ @ sol.unsqueeze(1) - b_).abs().max()
        print("cost fit over %d ranks (recompute mode): %.4f ns per own edge, %.3f ns per halo row, %.3f ns per own row, %.3f ms constant; "
              "largest residual %.3f ms   [partition.STEP_COST_NS = %s]" % (
                  len(samples), sol[0] * 1e6, sol[1] * 1e6, sol[2] * 1e6, sol[3], resid, dpart.STEP_COST_NS))
    print("summary (single-GPU step %.2f ms):" % single_ms)
    for world, mode, slow, mean, sp in summary:
        print("   N=%d %-9s slowest rank %.2f ms (mean %.2f): %.2fx with the exchange hidden" % (world, mode, slow, mean, sp))


if __name__ == "__main__":
    main()
