"""Who holds the interpreter lock when the mini-batch pipeline's threads stall for milliseconds?  A sampler thread sleeps 1 ms at a
time; when a wake-up comes more than 5 ms late (somebody held the lock through a long call that does not release it) it prints where
every thread is right then -- the holder is usually on the line of that call."""
import importlib.util
import os
import sys
import threading
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
stop = False
reports = []


def sampler():
    last = time.perf_counter()
    while not stop:
        time.sleep(0.001)
        now = time.perf_counter()
        late = now - last - 0.001
        names = {t.ident: t.name for t in threading.enumerate()}
        if late > 0.004 and len(reports) < 14 and "dgll-feature-loader" in names.values():
            frames = sys._current_frames()
            text = ["wake-up %.1f ms late:" % (late * 1e3)]
            for ident, fr in frames.items():
                if ident == threading.get_ident():
                    continue
                st = traceback.extract_stack(fr)[-3:]
                text.append("   %-24s %s" % (names.get(ident, ident), " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(st))))
            reports.append("\n".join(text))
        last = now


threading.Thread(target=sampler, daemon=True, name="gil-sampler").start()
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
sys.argv = ["bench.py", "--full-line", "--workload", "minibatch", "--no-cpu-baseline"] + sys.argv[1:]
bench.main()
stop = True
print("\n".join(reports), file=sys.stderr)
