"""Host time of an asynchronous H2D copy from pinned memory (what the loading stage pays per upload): idle stream against a stream with kernels queued."""
import time, torch
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
for mb in (0.5, 4.5, 5.0, 20.0):
    n = int(mb * 1e6 // 8)
    src = torch.empty(n, dtype=torch.int64, pin_memory=True)
    dst = torch.empty(n, dtype=torch.int64, device=dev)
    busy = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    for label, load in (("idle stream", False), ("behind 1 ms of kernels on the same stream", True)):
        ts = []
        with torch.cuda.stream(s):
            for i in range(30):
                if load:
                    for _ in range(4):
                        busy.mul_(1.0001)
                t0 = time.perf_counter()
                dst.copy_(src, non_blocking=True)
                ts.append(time.perf_counter() - t0)
                if i % 5 == 4:
                    s.synchronize()
        ts = sorted(ts[5:])
        print("H2D %.1f MB from pinned memory, %s: host time of the async copy call median %.1f us, p90 %.1f us" % (
            mb, label, ts[len(ts) // 2] * 1e6, ts[int(len(ts) * 0.9)] * 1e6))

# How many async H2D copies can be queued behind unfinished GPU work before the call blocks on the host?  (The loading stage queues
# two per batch behind the consumer's replay events.)
n = int(4.5e6 // 8)
src = torch.empty(n, dtype=torch.int64, pin_memory=True)
dst = torch.empty(n, dtype=torch.int64, device=dev)
busy = torch.empty(256 << 20, dtype=torch.float32, device=dev)
other = torch.cuda.Stream()
for trial in range(2):
    torch.cuda.synchronize()
    with torch.cuda.stream(other):
        for _ in range(400):                 # ~100+ ms of GPU work the copy stream depends on
            busy.mul_(1.0001)
        gate = torch.cuda.Event()
        gate.record(other)
    ts = []
    with torch.cuda.stream(s):
        s.wait_event(gate)
        for i in range(96):
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    slow = [(i, round(t * 1e3, 2)) for i, t in enumerate(ts) if t > 5e-4]
    print("96 async 4.5 MB H2D copies queued behind ~100 ms of GPU work: calls that blocked > 0.5 ms (index, ms): %s" % (slow or "none"))
