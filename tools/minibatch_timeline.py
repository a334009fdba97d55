#!/usr/bin/env python3
"""Where a mini-batch period goes on the GPU: reads a rocprofv3 --kernel-trace CSV of `bench.py --workload minibatch` and prints, per
hardware queue, the busy time per batch, and per kernel name the launches per batch and the average duration -- inside the pipeline,
i.e. with the loading stage's kernels running beside the consumer's step.

  rocprofv3 --kernel-trace -f csv -d gpurun_out/mbt -- python3 bench.py --workload minibatch --no-cpu-baseline --full-line
  python tools/minibatch_timeline.py gpurun_out/mbt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    total, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                total += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        total += cur_b - cur_a
    return total


def main():
    root = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "aggregate_rows_kernel"      # one launch per loaded batch
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    marks = [r for r in rows if marker in r[3]]
    if len(marks) < 40:
        print("only %d launches of %s" % (len(marks), marker))
        return
    # the steady window: between the marker launches at 50 % and 90 % of the run
    lo, hi = marks[len(marks) // 2][0], marks[int(len(marks) * 0.9)][0]
    n_batches = int(len(marks) * 0.9) - len(marks) // 2
    win = [r for r in rows if lo <= r[0] < hi]
    period = (hi - lo) / n_batches / 1e6
    print("window: %d batches, %.3f ms per batch (%.1f batches/s)" % (n_batches, period, 1e3 / period))
    by_q = defaultdict(list)
    for a, b, q, name in win:
        by_q[q].append((a, b))
    for q, iv in sorted(by_q.items(), key=lambda kv: -union(kv[1])):
        print("queue %-6s busy %.3f ms per batch (%d launches per batch)" % (q, union(iv) / n_batches / 1e6, len(iv) // n_batches))
    print("all queues together busy %.3f ms per batch" % (union([(a, b) for a, b, _, _ in win]) / n_batches / 1e6))
    pcie = [(a, b) for a, b, _, n in win if "aggregate_rows_kernel" in n or "upload_kernel" in n]
    print("zero-copy kernels (outermost-hop reduction, uploads) busy %.3f ms per batch" % (union(pcie) / n_batches / 1e6))
    by_k = defaultdict(lambda: [0, 0, set()])
    for a, b, q, name in win:
        k = by_k[name]
        k[0] += 1
        k[1] += b - a
        k[2].add(q)
    print("%-110s %8s %9s %9s  queues" % ("kernel", "per batch", "avg us", "us/batch"))
    for name, (cnt, ns, qs) in sorted(by_k.items(), key=lambda kv: -kv[1][1])[:28]:
        print("%-110s %8.2f %9.1f %9.1f  %s" % (name[:110], cnt / n_batches, ns / cnt / 1e3, ns / n_batches / 1e3, ",".join(sorted(qs))))


if __name__ == "__main__":
    main()
