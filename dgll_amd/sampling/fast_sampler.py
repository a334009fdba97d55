"""FastNeighborSampler -- the reference's multi-hop sampler (dgllsampler.py:10-21) with the per-seed loop in native
code (dgll_host_sample_neighbors), bit-exact with DGLLNeighborSampler / the reference under `random.seed(s)`.

The Python side keeps a CSR copy of DGraph.edges, hands the interpreter's Mersenne-Twister state to the native call and
stores the advanced state back, so it can be freely interleaved with other users of the global `random` module.
"""
import ctypes as C
import math
import random
import threading

import numpy as np
import torch

from .. import _lib
from .base_sampler import Base_sampler, sugbraph


def _setsize(k):
    """CPython 3.10 random.sample: threshold between the pool and the set algorithm."""
    setsize = 21
    if k > 5:
        setsize += 4 ** math.ceil(math.log(k * 3, 4))
    return setsize


def batch_seed(base_seed, epoch, batch):
    """The seed of batch `batch` of epoch `epoch` in the per-batch-seeded mode:  (base_seed << 40) | (epoch << 20) | batch
    (base_seed >= 0; epoch, batch < 2**20).  The reference's loop under random.seed(batch_seed(...)) right before that batch draws
    the same ids (oracle/sampler.py; tests/test_sampler.py)."""
    if base_seed < 0 or not (0 <= epoch < (1 << 20)) or not (0 <= batch < (1 << 20)):
        raise ValueError("batch_seed needs base_seed >= 0 and epoch, batch in [0, 2**20)")
    return (int(base_seed) << 40) | (int(epoch) << 20) | int(batch)


def _seed_key(seed):
    """random.seed(int): the 32-bit little-endian words of abs(seed), [0] for 0 (Modules/_randommodule.c)."""
    n = abs(int(seed))
    words = []
    while n:
        words.append(n & 0xffffffff)
        n >>= 32
    return np.array(words or [0], dtype=np.uint32)


class StagedBatch:
    """What sample_seeded(staging=) packed into one buffer: `buffer` (int64 numpy view of the pinned memory), `offsets`
    {"seeds": 0, "src": [offset of hop h's source ids, h < L-1], "ptr": [offset of hop h's row pointers]}, `rows` [rows of hop h]
    (rows of hop h + 1 = edges of hop h), `token` for the ring the buffer came from."""

    def __init__(self, token, buffer, offsets):
        self.token, self.buffer, self.offsets, self.rows = token, buffer, offsets, []

    def tensor(self):
        return torch.from_numpy(self.buffer)


class _LazyInput:
    """Stands for `subgs[0].src_nodes()` (the input nodes of the batch) without forcing the deferred translation."""

    def __init__(self, subg):
        self._subg = subg

    def resolve(self):
        return self._subg.src_nodes()


class FastNeighborSampler(Base_sampler):
    def __init__(self, fanouts, defer_last_hop=False):
        super().__init__()
        self.fanouts = fanouts
        self.defer_last_hop = defer_last_hop
        self._csr_lock = threading.Lock()
        self._csr_cache = None          # (edges object, indptr, indices): published as ONE reference, never filled in place

    def _csr(self, g):
        """The CSR arrays of g.edges.  Safe to call from several threads at once (MiniBatchPipeline(sampler_threads=K) does): the
        copy of a list-of-lists adjacency (dgraph.py:18-47) is built into locals under a lock and becomes visible only as a whole
        -- a reader either finds the finished tuple for this adjacency or builds it itself; it never sees arrays that are still
        being filled (round 4 published a zero-filled indptr first: a thread reading it then sampled a graph without edges)."""
        if hasattr(g.edges, "indptr"):           # CSRAdjacency: use the arrays directly
            return (np.ascontiguousarray(g.edges.indptr, dtype=np.int64), np.ascontiguousarray(g.edges.indices, dtype=np.int64))
        cached = self._csr_cache
        if cached is not None and cached[0] is g.edges:
            return cached[1], cached[2]
        with self._csr_lock:
            cached = self._csr_cache
            if cached is not None and cached[0] is g.edges:
                return cached[1], cached[2]
            edges = g.edges
            deg = np.fromiter((len(e) for e in edges), dtype=np.int64, count=len(edges))
            indptr = np.zeros(len(edges) + 1, dtype=np.int64)
            np.cumsum(deg, out=indptr[1:])
            indices = np.fromiter((u for e in edges for u in e), dtype=np.int64, count=int(indptr[-1]))
            self._csr_cache = (edges, indptr, indices)
            return indptr, indices

    def prepare(self, g):
        """Build the CSR copy of g.edges now (the pipeline calls this before it starts its sampler threads)."""
        self._csr(g)
        return self

    def sample_neighbours(self, g, nodes, fanout=None, defer_translation=False):
        indptr, indices = self._csr(g)
        seeds = np.ascontiguousarray(nodes.numpy() if isinstance(nodes, torch.Tensor) else np.asarray(nodes), dtype=np.int64)
        deg = indptr[seeds + 1] - indptr[seeds]
        cap = int(deg.sum() if fanout is None else np.minimum(deg, fanout).sum())
        src = np.empty(cap, dtype=np.int64)
        dst = np.empty(cap, dtype=np.int64)
        counts = np.empty(len(seeds), dtype=np.int64)
        version, internal, gauss = random.getstate()
        state = np.array(internal[:624], dtype=np.uint32)
        index = C.c_int(internal[624])
        n_out = C.c_int64(0)
        code = _lib.lib.dgll_host_sample_neighbors(
            state.ctypes.data, C.byref(index), indptr.ctypes.data, indices.ctypes.data, seeds.ctypes.data, len(seeds),
            -1 if fanout is None else int(fanout), _setsize(fanout) if fanout is not None else 21, src.ctypes.data,
            None if defer_translation else dst.ctypes.data, counts.ctypes.data, cap, C.byref(n_out))
        _lib.check(code, "dgll_host_sample_neighbors")
        random.setstate((version, tuple(int(x) for x in state) + (index.value,), gauss))
        ptr = np.zeros(len(seeds) + 1, dtype=np.int64)
        np.cumsum(counts, out=ptr[1:])
        finish = None
        if defer_translation:     # positions -> ids later (no generator state involved): whoever first reads the ids pays

            def finish():
                _lib.check(_lib.lib.dgll_host_translate_neighbors(indptr.ctypes.data, indices.ctypes.data, seeds.ctypes.data,
                                                                  len(seeds), counts.ctypes.data, src.ctypes.data,
                                                                  dst.ctypes.data), "dgll_host_translate_neighbors")

        sg = sugbraph(torch.from_numpy(src), torch.from_numpy(dst), torch.from_numpy(ptr), finish=finish)
        sg.max_degree = None if fanout is None else int(fanout)      # lets to_block() skip the long-row scan (host-only plan)
        if finish is not None:        # what a device-side translation needs instead (MiniBatchPipeline(device_graph=...))
            sg.pending_positions = (torch.from_numpy(seeds), torch.from_numpy(counts))
        return sg

    @staticmethod
    def staging_entries(batch_size, fanouts):
        """int64 entries of the staging buffer of one batch (sample_seeded(staging=)): seeds, the source ids of every hop but the
        outermost, the row pointers of every hop."""
        order = [int(f) for f in reversed(fanouts)]
        rows = [int(batch_size)]
        for f in order[:-1]:
            rows.append(rows[-1] * f)
        return rows[0] + sum(rows[1:]) + sum(r + 1 for r in rows)

    def sample_seeded(self, g, seed_nodes, seed, max_threads=1, last_hop_buffer=None, staging=None):
        """The whole batch under ITS OWN generator, seeded as random.seed(seed) seeds the interpreter's: the ids the reference loop
        draws right after that call.  Touches neither the global `random` state nor any shared scratch, and the native call runs
        without the GIL: several threads may each draw whole batches at the same time (MiniBatchPipeline(sampler_threads=K)).
        Returns what `sample` returns.  last_hop_buffer: optional callable(capacity) -> (int64 numpy array of that many entries,
        token) supplying the outermost hop's id/position array (the pipeline hands out pinned host memory so that the upload is
        an asynchronous DMA); the token is attached to the outermost sugbraph as `.buffer_token`.
        staging: optional callable(entries) -> (int64 numpy array, token): ONE (pinned) buffer that receives everything else of the
        batch a device consumer needs -- [seeds | source ids of hop 0 | ... of hop L-2 | row pointers of hop 0 | ... of hop L-1],
        each block at its upper-bound offset -- so that the loading stage uploads it with one copy instead of one per array
        (ten pageable copies of ~0.1 ms each were half of that thread's time per batch).  Every sugbraph of the batch then carries
        `.staged` = StagedBatch(token, buffer as a tensor, offsets, rows per hop)."""
        if any(f is None for f in self.fanouts):
            raise ValueError("sample_seeded needs integer fan-outs")
        indptr, indices = self._csr(g)
        seeds = np.ascontiguousarray(seed_nodes.numpy() if isinstance(seed_nodes, torch.Tensor) else np.asarray(seed_nodes), dtype=np.int64)
        order = [int(f) for f in reversed(self.fanouts)]
        L = len(order)
        fan = np.array(order, dtype=np.int64)
        setsizes = np.array([_setsize(f) for f in order], dtype=np.int64)
        caps, n_prev = [], len(seeds)
        for f in order:                                  # upper bounds: every hop seed keeps at most `fanout` neighbours
            caps.append(n_prev * f)
            n_prev = n_prev * f
        cap = np.array(caps, dtype=np.int64)
        src = [np.empty(c, dtype=np.int64) for c in caps]
        staged = None
        if staging is not None:
            rows_cap = [len(seeds)] + caps[:-1]                                # rows of hop h: upper bounds
            entries = rows_cap[0] + sum(rows_cap[1:]) + sum(r + 1 for r in rows_cap)
            got = staging(entries)
            if got is not None:
                buf, stoken = got
                off, o = {"seeds": 0, "src": [], "ptr": []}, rows_cap[0]
                buf[:rows_cap[0]] = seeds
                for h in range(L - 1):
                    off["src"].append(o)
                    src[h] = buf[o:o + caps[h]]
                    o += caps[h]
                for h in range(L):
                    off["ptr"].append(o)
                    o += rows_cap[h] + 1
                staged = StagedBatch(stoken, buf, off)
        token = None
        if last_hop_buffer is not None:
            got = last_hop_buffer(caps[-1])
            if got is not None:
                src[-1], token = got
        defer = bool(self.defer_last_hop)
        dst = [np.empty(c, dtype=np.int64) for c in caps]
        counts = [np.empty(len(seeds) if h == 0 else caps[h - 1], dtype=np.int64) for h in range(L)]
        ptrs = lambda arrs: (C.c_void_p * L)(*[a.ctypes.data for a in arrs])      # noqa: E731
        n_out = np.zeros(L, dtype=np.int64)
        key = _seed_key(seed)
        code = _lib.lib.dgll_host_sample_batch_seeded(
            key.ctypes.data, len(key), indptr.ctypes.data, indices.ctypes.data, seeds.ctypes.data, len(seeds), fan.ctypes.data,
            setsizes.ctypes.data, L, ptrs(src), ptrs(dst), ptrs(counts), cap.ctypes.data, n_out.ctypes.data, int(defer), int(max_threads))
        _lib.check(code, "dgll_host_sample_batch_seeded")
        subgs = []
        hop_seeds = seeds
        for h in range(L):
            n_h = int(n_out[h])
            cnt = counts[h][:len(hop_seeds)]
            if staged is not None:
                o = staged.offsets["ptr"][h]
                ptr = staged.buffer[o:o + len(hop_seeds) + 1]
                ptr[0] = 0
            else:
                ptr = np.zeros(len(hop_seeds) + 1, dtype=np.int64)
            np.cumsum(cnt, out=ptr[1:])
            s_h, d_h = src[h][:n_h], dst[h][:n_h]
            finish = None
            if defer and h == L - 1:
                finish = self._make_finish(indptr, indices, hop_seeds, cnt, s_h, d_h)
            sg = sugbraph(torch.from_numpy(s_h), torch.from_numpy(d_h), torch.from_numpy(ptr), finish=finish)
            sg.max_degree = order[h]
            if finish is not None:     # what a device-side translation needs instead of the host one (MiniBatchPipeline(device_graph=))
                sg.pending_positions = (torch.from_numpy(hop_seeds), torch.from_numpy(cnt))
            if h == L - 1:
                sg.buffer_token = token
            if staged is not None:
                staged.rows.append(len(hop_seeds))
                sg.staged = staged
            subgs.insert(0, sg)
            hop_seeds = s_h
        return _LazyInput(subgs[0]) if defer else subgs[0].src_nodes(), seed_nodes, subgs

    @staticmethod
    def _make_finish(indptr, indices, seeds, counts, src, dst):
        def finish():
            _lib.check(_lib.lib.dgll_host_translate_neighbors(indptr.ctypes.data, indices.ctypes.data, seeds.ctypes.data, len(seeds),
                                                              counts.ctypes.data, src.ctypes.data, dst.ctypes.data),
                       "dgll_host_translate_neighbors")

        return finish

    def sample(self, g, seed_nodes):
        output_nodes = seed_nodes
        subgs = []
        order = list(reversed(self.fanouts))
        for k, fanout in enumerate(order):
            last = k == len(order) - 1
            # the outermost hop's ids feed nothing inside this call: leave their translation to the first reader (the
            # pipeline's loading thread), so this thread -- the only one allowed to touch the generator -- moves on
            subg = self.sample_neighbours(g, seed_nodes, fanout, defer_translation=last and self.defer_last_hop)
            subgs.insert(0, subg)
            if not last:
                seed_nodes = subg.src_nodes()
        return _LazyInput(subgs[0]) if self.defer_last_hop else subgs[0].src_nodes(), output_nodes, subgs
