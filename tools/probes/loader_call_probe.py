"""Host time of the loading stage's ONE native call per batch (dgll_hip_load_sampled_batch: 2 async uploads + ~10 launches) inside the
running mini-batch bench: wall time of the ctypes call itself, per batch, against the loading thread's whole per-batch host time."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgll_amd import _lib  # noqa: E402

real = _lib.lib.dgll_hip_load_sampled_batch
times = []


def timed(stream, batch):
    t0 = time.perf_counter()
    code = real(stream, batch)
    times.append(time.perf_counter() - t0)
    return code


_lib.lib.dgll_hip_load_sampled_batch = timed
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
sys.argv = ["bench.py", "--full-line", "--workload", "minibatch", "--no-cpu-baseline"] + sys.argv[1:]
bench.main()
print("calls in order, us: " + " ".join("%d" % (t * 1e6) for t in times[:80]), file=sys.stderr)
ts = sorted(times[24:])
print("dgll_hip_load_sampled_batch: %d calls, host time median %.1f us, p90 %.1f us, max %.1f us" % (
    len(ts), ts[len(ts) // 2] * 1e6, ts[int(len(ts) * 0.9)] * 1e6, ts[-1] * 1e6), file=sys.stderr)
