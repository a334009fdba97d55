#!/usr/bin/env python3
"""The long-row threshold of the CSR plans (rows with more edges are cut into chunks whose partial sums a finalize launch combines:
5 such launches of 40-60 us in the 20 ms bench step) -- A/B of dgll_hip_debug_tune(3, T) on the headline step.
    python tools/threshold_ab.py 256,512,1024,2048 [bench.py arguments]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,512,1024,2048").split(",")]
rest = sys.argv[2:]
code = ("import sys, runpy; sys.path.insert(0, %r); from dgll_amd import _lib; _lib.check(_lib.lib.dgll_hip_debug_tune(3, int(sys.argv[1])), 'tune'); "
        "sys.argv = ['bench.py', '--full-line', '--no-extra-graphs', '--no-cpu-baseline', '--other-workloads', 'off'] + sys.argv[2:]; "
        "runpy.run_path(%r, run_name='__main__')" % (ROOT, os.path.join(ROOT, "bench.py")))
for rep in range(2):
    for t in vals:
        res = subprocess.run([sys.executable, "-c", code, str(t)] + rest, capture_output=True, text=True)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print("threshold %d: failed\n%s" % (t, res.stderr[-500:]))
            continue
        d = json.loads(line[-1])
        print("threshold %5d: %.3f ms per step | %s" % (t, d["ms_per_step"], "  ".join(
            "%s %.3f" % (k.split(" nnz")[0].replace("bfloat16 ", ""), v["avg_ms"]) for k, v in d["spmm_launch_table"].items())), flush=True)
