"""MQ-GNN style mini-batch pipeline: bounded queues between a sampling/feature-loading producer and the training
consumer, each stage on its own HIP stream.

Reference: /root/reference/dgll/GPU Accelerator/buffer_queues.py:22-46 (`sample_generator`: iterate the dataloader,
stage the batch on `d_stream`, `gpu_queue.put`), :74-113 (`sample_consumer`: `gpu_queue.get`, forward/backward on
`c_stream`, gradient sharing on `g_stream`), MQGCN.py:98,150-155 (BUFFER_SIZE = 4, one producer + one consumer thread,
`Queue(maxsize=4)` + `Condition`), README.md:27-29 (three queues: sampling -> feature loading -> training).

Here, the three queues of README.md:27-29 as two producer stages and the consumer: a SAMPLING thread draws the batches on
the host (stdlib RNG stream, bit-exact with the reference -- one thread, so the stream stays sequential) into a bounded
queue; a LOADING thread takes them, fetches the input features through the GraphCacheServer on a side stream (hot rows
from HBM, misses over PCIe from pinned memory), records an event and fills the second bounded queue; the consumer waits
on the event from its compute stream.  Sampling of batch i+2, loading of batch i+1 and training on batch i overlap.  The end of the epoch is signalled with a sentinel
(buffer_queues.py:43-46 sets a flag under the Condition).
"""
import queue
import threading

import torch

_DONE = object()


class Batch:
    __slots__ = ("input_nodes", "output_nodes", "subgraphs", "features", "labels", "ready", "step")

    def __init__(self):
        self.input_nodes = self.output_nodes = self.subgraphs = self.features = self.labels = self.ready = None
        self.step = 0


class MiniBatchPipeline:
    def __init__(self, dataloader, cache=None, labels=None, queue_size=4, device="cuda", hops=None):
        """dataloader: dgll_amd.dataloader.DataLoader; cache: GraphCacheServer (None: features come from
        dataloader.Dgraph.get_features on the host and are copied); hops: optional callable batch -> list of id tensors
        whose features are needed (default: the input nodes only, graphage.py:52)."""
        self.dataloader, self.cache, self.labels = dataloader, cache, labels
        self.device = torch.device(device)
        self.queue = queue.Queue(maxsize=queue_size)              # MQGCN.py:98 BUFFER_SIZE  (loaded batches)
        self.sampled = queue.Queue(maxsize=2)                     # sampled, not yet loaded: a short hand-over queue
        self.hops = hops
        self.load_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None   # d_stream
        self._thread = None
        self._error = None

    # ---- producer stage 1: sampling (buffer_queues.py:22-46) ---------------------------------------------------
    def _sample(self):
        try:
            for step, (inp, outp, subgs) in enumerate(self.dataloader):
                self.sampled.put((step, inp, outp, subgs))            # blocks while the queue is full
        except BaseException as exc:  # noqa: BLE001  (surface producer failures in the consumer)
            self._error = exc
        finally:
            self.sampled.put(_DONE)

    # ---- producer stage 2: feature loading --------------------------------------------------------------------
    def _load(self):
        try:
            while True:
                item = self.sampled.get()
                if item is _DONE:
                    break
                b = Batch()
                b.step, b.input_nodes, b.output_nodes, b.subgraphs = item
                if hasattr(b.input_nodes, "resolve"):      # FastNeighborSampler(defer_last_hop=True): the outermost hop's
                    b.input_nodes = b.input_nodes.resolve()    # positions become ids here, off the sampling thread
                inp, outp = b.input_nodes, b.output_nodes
                id_lists = self.hops(b) if self.hops is not None else [inp]
                if self.load_stream is not None:
                    with torch.cuda.stream(self.load_stream):
                        b.features = [self._fetch(ids) for ids in id_lists]
                        if self.labels is not None:
                            b.labels = self.labels[outp].to(self.device, non_blocking=True)
                        b.ready = torch.cuda.Event()
                        b.ready.record(self.load_stream)
                else:
                    b.features = [self._fetch(ids) for ids in id_lists]
                    if self.labels is not None:
                        b.labels = self.labels[outp]
                self.queue.put(b)                                   # blocks while the queue is full
        except BaseException as exc:  # noqa: BLE001
            self._error = exc
            while self.sampled.get() is not _DONE:                  # let the sampling thread run to its end
                pass
        finally:
            self.queue.put(_DONE)

    def _fetch(self, ids):
        if self.cache is not None:
            return self.cache.fetch_data(ids, stream=self.load_stream)
        feats = self.dataloader.Dgraph.get_features(ids)
        return feats.to(self.device, non_blocking=True)

    # ---- consumer side -----------------------------------------------------------------------------------------
    def __iter__(self):
        self._error = None
        self._thread = threading.Thread(target=self._sample, name="dgll-sample-producer", daemon=True)
        self._loader = threading.Thread(target=self._load, name="dgll-feature-loader", daemon=True)
        self._thread.start()
        self._loader.start()
        while True:
            b = self.queue.get()
            if b is _DONE:
                break
            if b.ready is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(b.ready)                      # compute stream waits for the load stream
                # The tensors were allocated on the load stream and are consumed on `cur`: tell the caching allocator, or
                # the block returns to the load stream's pool when the consumer drops the batch -- while forward/backward
                # kernels reading it may still be queued -- and the loader's next gather could be written into it.
                for t in b.features or ():
                    t.record_stream(cur)
                if b.labels is not None and b.labels.is_cuda:
                    b.labels.record_stream(cur)
            yield b
        self._thread.join()
        self._loader.join()
        if self._error is not None:
            raise self._error
