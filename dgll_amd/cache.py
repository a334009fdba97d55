"""GraphCacheServer -- GPU-resident hot-node feature cache with a pinned-host miss path.

Mirror of /root/reference/dgll/FeatureCache/storage.py:12-221 without its DGL (NodeFlow / Frame) plumbing:
  * `auto_cache(out_degrees)`: capacity = free device memory / (total_dim * bytes) (storage.py:71-78); if everything fits,
    cache all nodes, else the top-out-degree nodes, statically (storage.py:84-98);
  * `cache_fix_data(nids, data, is_full)`: install rows + the localid -> cacheid map + the gpu_flag mask (storage.py:127-148);
  * `fetch_data(nids)`: features of a batch of node ids as ONE device tensor.  The reference splits the batch by
    `gpu_flag`, gathers hits on the GPU and misses on the CPU, copies, and merges in place (storage.py:151-198); here
    that is one HIP launch (dgll_hip_gather_rows) reading the HBM cache and the pinned host array directly;
  * `log_miss_rate` / `get_miss_rate` (storage.py:213-220).
Host features are kept in pinned memory so the GPU can read them (and `hipMemcpyAsync` them) without staging.
"""
import os
import threading

import torch

from . import _lib
from .ranges import rng
from .ops import _dtype_code


# Row pitches (bytes) of the pinned host store and of the HBM cache block; 1 = the rows' natural pitch (the default).  Measured, round 6
# (tools/probes/pcie_probe.py, tools/probes/row_pitch_ab.sh): rows that start on 128-byte boundaries are gathered over PCIe at 54.5 GB/s
# against 48.7 at the Reddit pitch of 1204 bytes, and 16-byte pitches put the outermost hop's reduction on 16-byte lanes -- but the
# PIPELINE did not get faster at the 50 % cache (627-637 against 623-645 batches/s) and lost 10 % with everything cached (727-755
# against 820-848: the 16-byte-lane reduction keeps four rows in flight per wavefront and takes more of the CUs from the step beside it).
HOST_ROW_ALIGN = int(os.environ.get("DGLL_HOST_ROW_ALIGN", "1"))
CACHE_ROW_ALIGN = int(os.environ.get("DGLL_CACHE_ROW_ALIGN", "1"))


def _padded_rows(t, align, pin=False):
    """[n, D] view of a zero-padded buffer whose rows start `align` bytes apart (a plain contiguous copy when they already do)."""
    esz = t.element_size()
    row = int(t.shape[1]) * esz
    if align <= 1 or row % align == 0 or align % esz:
        return t.contiguous().pin_memory() if pin else t.contiguous()
    ld = (-(-row // align) * align) // esz
    store = torch.zeros((int(t.shape[0]), ld), dtype=t.dtype, device=t.device, pin_memory=pin)
    store[:, :t.shape[1]] = t
    return store[:, :t.shape[1]]


class GraphCacheServer:
    def __init__(self, features, node_num=None, nid_map=None, gpuid=0):
        """features: [N, D] CPU tensor (fp32 or bf16) -- the 'remote server' of storage.py:100-125.  It is pinned here.
        nid_map: optional local -> full-graph id map (storage.py:27)."""
        self.gpuid = gpuid
        self.device = torch.device("cuda", gpuid)
        self.node_num = int(features.shape[0] if node_num is None else node_num)
        self.features = features if features.is_pinned() else _padded_rows(features, HOST_ROW_ALIGN, pin=True)
        self.nid_map = None if nid_map is None else nid_map.clone().detach().to(self.device)
        self.total_dim = int(features.shape[1])
        self.dims = {"features": self.total_dim}
        self.gpu_flag = torch.zeros(self.node_num, dtype=torch.bool, device=self.device)
        self.localid2cacheid = torch.full((self.node_num,), -1, dtype=torch.int64, device=self.device)
        self.gpu_fix_cache = None
        self.cached_num = 0
        self.capability = self.node_num
        self.full_cached = False
        self.log = False
        self.try_num = 0
        self.miss_num = 0
        self._pending = []          # (device miss counter, batch size) of logged fetches not yet read back
        self._pending_lock = threading.Lock()   # fetches come from the pipeline's loading thread

    # ---- cache population ------------------------------------------------------------------------------------
    def auto_cache(self, out_degrees, reserve_bytes=1 << 30, capacity=None):
        """storage.py:64-98.  `capacity` (rows) overrides the free-memory rule."""
        if capacity is None:
            free, _total = torch.cuda.mem_get_info(self.device)
            available = free - reserve_bytes
            capacity = max(int(available // (self.total_dim * self.features.element_size())), 0)
        self.capability = int(capacity)
        if self.capability >= self.node_num:
            full_nids = torch.arange(self.node_num, device=self.device)
            self.cache_fix_data(full_nids, self.get_feat_from_server(full_nids), is_full=True)
        else:
            sort_nid = torch.argsort(out_degrees.to(self.device), descending=True, stable=True)
            cache_nid = sort_nid[:self.capability]
            self.cache_fix_data(cache_nid, self.get_feat_from_server(cache_nid), is_full=False)

    # ---- global neighbour-sampling cache (README.md:27-29: "global neighbor sampling with caching") ---------------------
    def global_sampling_cache(self, weights, capacity=None, seed=0, reserve_bytes=1 << 30):
        """Populate the cache with a GLOBAL SAMPLE of the nodes instead of the static top-degree set: `capacity` nodes
        drawn without replacement with probability proportional to `weights` (out-degrees: the chance that a neighbour
        sampler reaches a node grows with its degree; any importance vector works).  Re-drawing periodically (`refresh`)
        rotates the tail of the distribution through the cache while the hubs stay in with near certainty.  The sampled ids
        of the mini-batches are NOT biased by the cache (they stay bit-identical to the reference sampler's)."""
        if capacity is None:
            free, _total = torch.cuda.mem_get_info(self.device)
            capacity = max(int((free - reserve_bytes) // (self.total_dim * self.features.element_size())), 0)
        self.capability = int(min(capacity, self.node_num))
        self._gs_weights = weights.to(self.device, dtype=torch.float32).clamp(min=0) + 1e-12
        self._gs_gen = torch.Generator(device=self.device)
        self._gs_gen.manual_seed(seed)
        self.refresh()

    def refresh(self):
        """Draw a new global sample (weighted, without replacement: Gumbel top-k on the device) and install it."""
        if self.capability >= self.node_num:
            full = torch.arange(self.node_num, device=self.device)
            return self.cache_fix_data(full, self.get_feat_from_server(full), is_full=True)
        u = torch.rand(self.node_num, generator=self._gs_gen, device=self.device).clamp_(min=1e-20)
        keys = torch.log(self._gs_weights) - torch.log(-torch.log(u))        # Gumbel-max: top-k == weighted sample w/o replacement
        nids = torch.topk(keys, self.capability).indices
        self.cache_fix_data(nids, self.get_feat_from_server(nids), is_full=False)

    def record_access(self, nids, stream=None):
        """Accumulate how often every node is fetched (device-side, no host sync) for `refresh_from_access`.  The counter
        vector is shared between the stream that records (the pipeline's loading stream) and the one that refreshes: every
        update is bracketed by events (`_access_event`), never by a host synchronisation."""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        # The whole enqueue -- wait for the other side's last event, the update, the new event -- happens under the lock, and so does
        # refresh_from_access's: the two streams' accesses to the counter vector are then totally ordered by the events (a recorder that
        # had snapshotted the previous event BEFORE a refresh published its own ran its index_add_ next to the refresh's top-k and
        # decay: lost increments or a torn read -- ADVICE round 3).  Nothing here synchronises the host with the GPU.
        with torch.cuda.stream(stream):                    # the id upload (it may block on a pageable source) stays outside the lock
            nids = nids.to(self.device, dtype=torch.int64, non_blocking=True)
            ones = torch.ones_like(nids)
        with self._pending_lock:
            with torch.cuda.stream(stream):
                if getattr(self, "_access", None) is None:
                    self._access = torch.zeros(self.node_num, dtype=torch.int64, device=self.device)
                prev = getattr(self, "_access_event", None)
                if prev is not None:
                    stream.wait_event(prev)
                self._access.index_add_(0, nids, ones)
                ev = torch.cuda.Event()
                ev.record(stream)
            self._access_event = ev

    def refresh_from_access(self, decay=0.5):
        """Re-populate the cache with the nodes fetched most often since the last refresh (the empirical version of the global
        sample: it follows the training set's actual neighbourhood distribution); counts are decayed IN PLACE, not cleared."""
        if getattr(self, "_access", None) is None or self.capability >= self.node_num:
            return
        cur = torch.cuda.current_stream(self.device)
        with self._pending_lock:                           # see record_access: reads and the decay are ordered against every recorder
            access, prev = self._access, getattr(self, "_access_event", None)
            if prev is not None:
                cur.wait_event(prev)                       # the loading stream's last index_add_
            nids = torch.topk(access, self.capability).indices
            access.copy_((access.to(torch.float64) * decay).to(torch.int64))
            ev = torch.cuda.Event()
            ev.record(cur)
            self._access_event = ev
        self.cache_fix_data(nids, self.get_feat_from_server(nids), is_full=False)

    def get_feat_from_server(self, nids, to_gpu=False):
        """Rows of the host feature store for local ids `nids` (storage.py:100-125)."""
        with rng("cache-cpu"):
            full = nids if self.nid_map is None else self.nid_map[nids]
            rows = self.features[full.cpu()]
            return rows.to(self.device, non_blocking=True) if to_gpu else rows

    def cache_fix_data(self, nids, data, is_full=False):
        """storage.py:127-148.  Safe DURING iteration (the pipeline's loading thread fetches on its own stream while the
        training thread refreshes): the new (slot map, cache block) pair is built off to the side on the caller's stream and
        published as ONE tuple together with the event that marks it complete; a fetch snapshots the tuple under the lock,
        makes its stream wait for that event and `record_stream`s both tensors, so the old pair is neither half-written
        nor recycled by the allocator while a gather still reads it."""
        rows = nids.size(0)
        assert rows == data.size(0)                                            # storage.py:142-143
        cur = torch.cuda.current_stream(self.device)
        slot_map = torch.full((self.node_num,), -1, dtype=torch.int64, device=self.device)
        slot_map[nids] = torch.arange(rows, device=self.device)
        block = _padded_rows(data.to(self.device), CACHE_ROW_ALIGN)
        flag = torch.zeros(self.node_num, dtype=torch.bool, device=self.device)
        flag[nids] = True
        ready = torch.cuda.Event()
        ready.record(cur)
        with self._pending_lock:
            self._state = (slot_map, block, ready)
            self.localid2cacheid, self.gpu_fix_cache, self.gpu_flag = slot_map, block, flag
            self.cached_num = rows
            self.full_cached = is_full

    # ---- the hot call ----------------------------------------------------------------------------------------
    def fetch_data(self, nids, out=None, stream=None, contiguous=False):
        """[len(nids), D] device tensor of the nodes' features (ids in the local space).  The rows sit on a 16-byte pitch (602 bf16
        columns -> 608): when D is not a whole number of vectors the result is a VIEW [n, D] of that padded buffer whose padding
        columns are uninitialised (the reference's fetch_data, storage.py:151-198, returns a dense tensor); contiguous=True returns a
        dense copy instead, for callers that flatten, export or assert contiguity.  ONE launch and no host
        synchronisation whatever the cache state: with a partition (`nid_map`) the kernel resolves hit -> cache slot of
        the LOCAL id, miss -> host row nid_map[local id] itself (dgll_hip_gather_rows_mapped).  Everything -- the id
        upload, the allocation of `out`, the launch -- is issued on `stream` (default: the current stream), so `out`
        belongs to that stream; a consumer on another stream must wait for it and `record_stream` the tensor.  The miss
        count of a logged fetch is read back lazily (get_miss_rate), not here."""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        with torch.cuda.stream(stream), rng("cache-index"):
            nids = nids.to(self.device, dtype=torch.int64, non_blocking=True)
            n = int(nids.numel())
            if out is None:
                # rows start on 16-byte boundaries (602 bf16 columns -> a leading dimension of 608): what consumes them -- the
                # MFMA transform, the split-K weight gradient -- then reads them as they are instead of re-laying them out
                epv = 16 // self.features.element_size()
                ld = -(-self.total_dim // epv) * epv
                store = torch.empty((n, ld), dtype=self.features.dtype, device=self.device)
                out = store[:, :self.total_dim] if ld != self.total_dim else store
            if n == 0:
                return out
            with self._pending_lock:
                slot_map, cache, ready = getattr(self, "_state", (None, None, None))
            use_map = cache is not None
            if ready is not None:
                stream.wait_event(ready)          # the pair may have been installed by another stream a moment ago
            host_map = self.nid_map          # full_cached: every id hits, the map is never consulted
            counter = None
            if self.log and use_map:
                counter = torch.zeros(1, dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device), rng("cache-gpu"):
                code = _lib.lib.dgll_hip_gather_rows_mapped(
                    stream.cuda_stream, cache.data_ptr() if use_map else None, cache.stride(0) if use_map else 0,
                    self.features.data_ptr(), self.features.stride(0), nids.data_ptr(),
                    slot_map.data_ptr() if use_map else None,
                    host_map.data_ptr() if host_map is not None else None, out.data_ptr(), out.stride(0), n,
                    self.total_dim, _dtype_code(out), counter.data_ptr() if counter is not None else None)
            _lib.check(code, "dgll_hip_gather_rows_mapped")
            if use_map:                           # a refresh may drop the pair while this gather still runs on `stream`
                slot_map.record_stream(stream)
                cache.record_stream(stream)
            if self.log:
                done = None
                if counter is not None:
                    done = torch.cuda.Event()
                    done.record(stream)           # the counter is final once THIS stream has passed the gather
                with self._pending_lock:
                    self._pending.append((counter, n, done))
            if contiguous and not out.is_contiguous():
                out = out.contiguous()
        return out

    def fetch_data_into(self, nids_list, outs, stream=None):
        """fetch_data for several id vectors at once, each into its own caller-owned [len(ids), D] row block (`outs`, e.g. row slices
        of a captured step's static inputs): ONE snapshot of the cache state, one miss counter and one event for the lot, one launch
        per block -- the bookkeeping of a fetch (lock, counter tensor, event, record_stream) costs more host time than its launch,
        and the loading stage of the mini-batch pipeline is host-bound.  ids: int64 device tensors (local id space)."""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        with torch.cuda.stream(stream):
            with rng("cache-index"):
                with self._pending_lock:
                    slot_map, cache, ready = getattr(self, "_state", (None, None, None))
                use_map = cache is not None
                if ready is not None:
                    stream.wait_event(ready)
                host_map = self.nid_map
                counter = torch.zeros(1, dtype=torch.int64, device=self.device) if (self.log and use_map) else None
            total = 0
            with torch.cuda.device(self.device), rng("cache-gpu"):
                for nids, out in zip(nids_list, outs):
                    n = int(nids.numel())
                    if n == 0:
                        continue
                    if out.shape[0] != n or out.shape[1] != self.total_dim or out.stride(1) != 1 or out.dtype != self.features.dtype:
                        raise ValueError("fetch_data_into: every output block is [len(ids), D] of the feature dtype, unit column stride")
                    total += n
                    code = _lib.lib.dgll_hip_gather_rows_mapped(
                        stream.cuda_stream, cache.data_ptr() if use_map else None, cache.stride(0) if use_map else 0,
                        self.features.data_ptr(), self.features.stride(0), nids.data_ptr(),
                        slot_map.data_ptr() if use_map else None, host_map.data_ptr() if host_map is not None else None,
                        out.data_ptr(), out.stride(0), n, self.total_dim, _dtype_code(out),
                        counter.data_ptr() if counter is not None else None)
                    _lib.check(code, "dgll_hip_gather_rows_mapped")
            if use_map:
                slot_map.record_stream(stream)
                cache.record_stream(stream)
            if self.log and total:
                done = None
                if counter is not None:
                    done = torch.cuda.Event()
                    done.record(stream)
                with self._pending_lock:
                    self._pending.append((counter, total, done))
        return outs

    def native_load_begin(self, stream):
        """What a native loading call (dgll_hip_load_sampled_batch) needs of the server, with fetch_data's bookkeeping done once:
        (cache ptr, ldc, host ptr, ldh, slot ptr, host_map ptr, miss-counter tensor or None, keep-alive tuple).  `stream` is made to
        wait for the current (slot map, cache block) pair; call native_load_end afterwards."""
        with self._pending_lock:
            slot_map, cache, ready = getattr(self, "_state", (None, None, None))
        use_map = cache is not None
        if ready is not None:
            stream.wait_event(ready)
        counter = None
        if self.log and use_map:
            with torch.cuda.stream(stream):          # zeroed on the stream whose kernels add to it (not on the caller's current stream)
                counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        return (cache.data_ptr() if use_map else None, cache.stride(0) if use_map else 0, self.features.data_ptr(), self.features.stride(0),
                slot_map.data_ptr() if use_map else None, self.nid_map.data_ptr() if self.nid_map is not None else None, counter,
                (slot_map, cache))

    def native_load_end(self, stream, counter, keep, tries, done=None):
        slot_map, cache = keep
        if cache is not None:
            slot_map.record_stream(stream)
            cache.record_stream(stream)
        if self.log and tries:
            if counter is not None and done is None:
                done = torch.cuda.Event()
                done.record(stream)
            with self._pending_lock:
                self._pending.append((counter, int(tries), done if counter is not None else None))

    def aggregate_data(self, nids, rowptr, reduce="mean", stream=None, out=None):
        """[len(rowptr) - 1, D] rows: row i = mean (or sum) of the features of nids[rowptr[i]:rowptr[i+1]] -- fetch_data fused with
        the neighbour reduction of the layer that consumes the rows (sageconv.py:33-36).  For the OUTERMOST hop of a sampled batch,
        whose features enter the model through that reduction only: its fan-out x batch rows are read from the HBM cache / the pinned
        host store by the reducing kernel and never written (dgll_hip_aggregate_rows_mapped).  Stream semantics, cache-refresh safety
        and miss accounting as fetch_data (every id counts as one try)."""
        if reduce not in ("mean", "sum"):
            raise ValueError("aggregate_data reduces with 'mean' or 'sum'")
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        with torch.cuda.stream(stream), rng("cache-index"):
            nids = nids.to(self.device, dtype=torch.int64, non_blocking=True)
            rowptr = rowptr.to(self.device, dtype=torch.int64, non_blocking=True)
            n_rows = int(rowptr.numel()) - 1
            esz = self.features.element_size()
            epv = 16 // esz
            ld = -(-self.total_dim // epv) * epv
            if out is None:
                store = torch.zeros((n_rows, ld), dtype=self.features.dtype, device=self.device) if ld != self.total_dim else \
                    torch.empty((n_rows, ld), dtype=self.features.dtype, device=self.device)
                out = store[:, :self.total_dim] if ld != self.total_dim else store
            elif out.shape[0] != n_rows or out.shape[1] != self.total_dim or out.stride(1) != 1:
                raise ValueError("aggregate_data(out=): [len(rowptr) - 1, D] rows, unit column stride")
            if n_rows <= 0:
                return out
            if (self.total_dim * esz) % 4 != 0 or (self.features.stride(0) * esz) % 4 != 0:
                raise ValueError("aggregate_data needs 4-byte granular feature rows (fetch_data + a block reduction handle the rest)")
            with self._pending_lock:
                slot_map, cache, ready = getattr(self, "_state", (None, None, None))
            use_map = cache is not None
            if ready is not None:
                stream.wait_event(ready)
            host_map = self.nid_map
            counter = None
            if self.log and use_map:
                counter = torch.zeros(1, dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device), rng("cache-gpu"):
                code = _lib.lib.dgll_hip_aggregate_rows_mapped(
                    stream.cuda_stream, cache.data_ptr() if use_map else None, cache.stride(0) if use_map else 0,
                    self.features.data_ptr(), self.features.stride(0), nids.data_ptr(),
                    slot_map.data_ptr() if use_map else None, host_map.data_ptr() if host_map is not None else None,
                    rowptr.data_ptr(), out.data_ptr(), out.stride(0), n_rows, self.total_dim, _dtype_code(out),
                    _lib.REDUCE_MEAN if reduce == "mean" else _lib.REDUCE_SUM, counter.data_ptr() if counter is not None else None)
            _lib.check(code, "dgll_hip_aggregate_rows_mapped")
            if use_map:
                slot_map.record_stream(stream)
                cache.record_stream(stream)
            if self.log:
                done = None
                if counter is not None:
                    done = torch.cuda.Event()
                    done.record(stream)
                with self._pending_lock:
                    self._pending.append((counter, int(nids.numel()), done))
        return out

    # ---- accounting (storage.py:213-220) -----------------------------------------------------------------------
    def log_miss_rate(self, miss_num, total_num):
        self.try_num += total_num
        self.miss_num += miss_num

    def _drain(self):
        """Fold the device-side miss counters of the logged fetches into the totals (the only host read-back)."""
        with self._pending_lock:
            pending, self._pending = self._pending, []
        for counter, n, done in pending:
            if done is not None:
                done.synchronize()                # written on the loading stream: wait for that stream, not the current one
            self.log_miss_rate(int(counter.item()) if counter is not None else 0, n)

    def get_miss_rate(self):
        self._drain()
        miss_rate = float(self.miss_num) / self.try_num
        self.miss_num = 0
        self.try_num = 0
        return miss_rate


def gather_rows(x, idx, out=None):
    """x[idx] for a device (or pinned host) matrix x through the HIP gather kernel (dgraph.py:105)."""
    dev = idx.device
    if out is None:
        epv = 16 // x.element_size()
        ld = -(-x.shape[1] // epv) * epv
        store = torch.empty((idx.numel(), ld), dtype=x.dtype, device=dev)
        out = store[:, :x.shape[1]] if ld != x.shape[1] else store
    with torch.cuda.device(dev):
        code = _lib.lib.dgll_hip_gather_rows(torch.cuda.current_stream(dev).cuda_stream, None, 0, x.data_ptr(), x.stride(0),
                                             idx.data_ptr(), None, out.data_ptr(), out.stride(0), int(idx.numel()), x.shape[1],
                                             _dtype_code(x), None)
    _lib.check(code, "dgll_hip_gather_rows")
    return out
