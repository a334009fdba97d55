"""The partitioned engine with the REAL kernels: two ranks sharing the one GPU of the test box (gloo transport, which
accepts device tensors), full-graph GraphSage forward + backward vs the single-process result.  (RCCL itself needs one
GPU per rank and is exercised by the driver's multi-GPU bench.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, dtype_name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import nn as dnn
        from dgll_amd import ops, synth

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dtype = getattr(torch, dtype_name)
        torch.manual_seed(0)
        full = synth.products_like_graph(dev, seed=3, n=6000, n_undirected=60000, locality=0.8, n_blocks=8)
        n = full.n_rows
        model = dnn.GraphSage(40, [64, 64, 10], None).to(dev)
        x = torch.randn(n, 40, device=dev)
        gout = torch.randn(n, 10, device=dev)
        xin = ops.alloc_features(n, 40, dtype, dev)
        xin.copy_(x)
        # single-process reference
        ref = model.forward_graph(full, xin)
        (ref.float() * gout).sum().backward()
        ref_grads = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        # partitioned
        part = ddist.partition_contiguous(full, world, rank)
        engine = ddist.DistGraph(part, dev)
        engine.verify()
        blk = slice(part.own_begin, part.own_end)
        x_local = ops.alloc_features(part.n_own, 40, dtype, dev)
        x_local.copy_(x[blk])
        placed = engine.place_input_halo(x_local)
        out = engine.sage_forward(model, x_local, placed)
        tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=1e-1)   # bf16: the local partial is rounded once more
        torch.testing.assert_close(out.float(), ref[blk].float(), **tol)
        (out.float() * gout[blk]).sum().backward()
        racom = ddist.RaCoM(model.parameters(), dev)
        racom.all_reduce_and_wait()                      # averages: compare with ref / world
        for p, r in zip(model.parameters(), ref_grads):
            torch.testing.assert_close(p.grad * world, r, rtol=2e-2 if dtype == torch.bfloat16 else 2e-3,
                                       atol=(1.5e-1 if dtype == torch.bfloat16 else 2e-3) * float(r.abs().max()))   # bf16: ReLU-mask flips near 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_two_ranks_on_one_gpu_match_single_process(dtype_name):
    mp.spawn(_worker, args=(2, _free_port(), dtype_name), nprocs=2, join=True)


def _gat_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import nn as dnn
        from dgll_amd import synth

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        torch.manual_seed(0)
        full = synth.products_like_graph(dev, seed=4, n=5000, n_undirected=40000, locality=0.7, n_blocks=8, self_loops=True)
        n = full.n_rows
        model = dnn.SpGAT(24, 8, 7, dropout=0.0, alpha=0.2, nheads=4).to(dev)
        x = torch.randn(n, 24, device=dev)
        gout = torch.randn(n, 7, device=dev)
        ref = model(x, full)
        (ref * gout).sum().backward()
        ref_grads = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        part = ddist.partition_contiguous(full, world, rank)
        engine = ddist.DistGraph(part, dev)
        engine.verify()
        blk = slice(part.own_begin, part.own_end)
        out = engine.spgat_forward(model, x[blk].contiguous())
        torch.testing.assert_close(out, ref[blk], rtol=1e-4, atol=1e-4)
        (out * gout[blk]).sum().backward()
        ddist.RaCoM(model.parameters(), dev).all_reduce_and_wait()
        for p, r in zip(model.parameters(), ref_grads):
            torch.testing.assert_close(p.grad * world, r, rtol=2e-3, atol=2e-3 * float(r.abs().max()))
    finally:
        dist.destroy_process_group()


def test_partitioned_gat_two_ranks_match_single_process():
    """Config 4's multi-GPU leg: the fused edge-softmax kernels over the split adjacency, forward and backward."""
    mp.spawn(_gat_worker, args=(2, _free_port()), nprocs=2, join=True)
