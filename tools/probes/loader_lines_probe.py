"""Where the loading thread's host time per batch goes inside MiniBatchPipeline._load_into_set_native: the function is re-run under
sys.setprofile-free line stamps (a copy of its source with perf_counter() stamps between statements), inside the running bench."""
import importlib.util
import inspect
import os
import re
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgll_amd import pipeline  # noqa: E402

src = textwrap.dedent(inspect.getsource(pipeline.MiniBatchPipeline._load_into_set_native))
lines = src.split("\n")
out, stamps = [], {}
depth0 = None
for i, ln in enumerate(lines):
    out.append(ln)
    m = re.match(r"^(    )(\S.*)$", ln)          # top-level statements of the function body
    if m and i > 4 and not ln.rstrip().endswith((",", "(", "\\", ":")) and not ln.strip().startswith(("#", '"""', "import", "from")):
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        if re.match(r"^    \S", nxt) or nxt.strip() == "":
            out.append("    _S[%d] = _S.get(%d, 0.0) + (_pc() - _t); _t = _pc()" % (i, i))
body = "\n".join(out)
body = body.replace('        """', '        """', 1)
body = re.sub(r"(\n        C\+\+; what stays in Python[^\n]*\n)", r"\1", body)
# first stamp initialisation right after the docstring's end: insert before the first 'import ctypes' line
body = body.replace("    import ctypes as C\n", "    _t = _pc()\n    import ctypes as C\n", 1)
ns = dict(pipeline.__dict__)
ns["_S"], ns["_pc"] = stamps, time.perf_counter
exec(compile(body, "<stamped>", "exec"), ns)
pipeline.MiniBatchPipeline._load_into_set_native = ns["_load_into_set_native"]

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
sys.argv = ["bench.py", "--full-line", "--workload", "minibatch", "--no-cpu-baseline"] + sys.argv[1:]
bench.main()
tot = sum(stamps.values())
print("stamped sections of _load_into_set_native, total %.1f ms over the run:" % (tot * 1e3), file=sys.stderr)
for i, t in sorted(stamps.items(), key=lambda kv: -kv[1])[:14]:
    print("  %6.1f ms  %5.1f %%  line %3d: %s" % (t * 1e3, 100 * t / tot, i, lines[i].strip()[:120]), file=sys.stderr)
