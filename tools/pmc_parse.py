#!/usr/bin/env python3
"""Turn the counter files of tools/pmc_bench.sh into profiles/traffic.json entries.

    python tools/pmc_parse.py gpurun_out/<tag> [--round r03] [--write]

Per launch kind of bench.py's `spmm_launch_table` (matched to a kernel by its `kernel_fragment`): mean FETCH_SIZE / WRITE_SIZE
(KiB) over the launches of the timed + warm-up steps, skipping the first step (cold caches).  The three calibration launches
(`bench.py --calibrate`: an identity gather with KNOWN bytes, the FIRST three launches of the unweighted hidden-width SpMM
kernel) give the correction ratios of this very pass (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports about 1/2 of a wide
coalesced read on gfx950 -- the measured ratio is applied, not a flat 2):
    hbm_bytes = FETCH_SIZE_KiB * 1024 / ratio_read + WRITE_SIZE_KiB * 1024 / ratio_write.
Every entry carries the build stamp of the libdgll_hip.so the pass ran (bench.py prints it in roofline.build_stamp); bench.py
uses an entry only when workload, kernel AND stamp match.
"""
import argparse
import csv
import json
import os
import sys
from collections import OrderedDict


def read(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def per_kernel(rows, counter):
    out = OrderedDict()
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        out.setdefault(r["Kernel_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                     int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k in out:
        out[k].sort()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--round", default="r03")
    ap.add_argument("--write", action="store_true", help="merge the entries into profiles/traffic.json")
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = None
    for name in ("pmc_FETCH_SIZE.json", "bench_under_rocprof.json"):
        try:
            with open(os.path.join(args.dir, name)) as f:
                for line in f:
                    if line.startswith("{"):
                        bench = json.loads(line)
        except OSError:
            pass
        if bench:
            break
    if not bench:
        sys.exit("no bench JSON line found in " + args.dir)
    cfg = bench["config"]
    stamp = (bench.get("roofline") or {}).get("build_stamp")
    n, hidden = cfg["nodes"], cfg["hidden"]
    esz = 2 if bench["dtype"] == "bf16" else 4
    fetch = per_kernel(read(os.path.join(args.dir, "pmc_FETCH_SIZE.csv")), "FETCH_SIZE")
    write = per_kernel(read(os.path.join(args.dir, "pmc_WRITE_SIZE.csv")), "WRITE_SIZE")
    hit = miss = {}
    try:
        rows = read(os.path.join(args.dir, "pmc_TCC_HIT_sum_TCC_MISS_sum.csv"))
        hit, miss = per_kernel(rows, "TCC_HIT_sum"), per_kernel(rows, "TCC_MISS_sum")
    except OSError:
        pass
    # calibration: the first three launches of the unweighted F = hidden SpMM kernel read n*hidden*esz + n*12 bytes (rows,
    # rowptr, col) and write n*hidden*esz
    t = "unsigned short" if esz == 2 else "float"
    vecs = -(-hidden // (16 // esz))
    lpr = 4
    while lpr < 64 and lpr < vecs:
        lpr <<= 1
    calib_frag = "spmm_csr_kernel<%s, %s, %d, %d, false, 4, false, false>" % (t, t, 16 // esz, lpr)
    n_cal = cfg.get("calibrate_rows") or n           # rmat27 calibrates on a slice of the rows
    calib_kernel = next((k for k in fetch if calib_frag in k), None)
    ratio_r, ratio_w, calib = 0.5, 1.0, None
    if calib_kernel and len(fetch[calib_kernel]) >= 3:
        known_r = n_cal * hidden * esz + n_cal * 12
        known_w = n_cal * hidden * esz
        fr = sum(v for _, v, _ in fetch[calib_kernel][:3]) / 3 * 1024
        wr = sum(v for _, v, _ in write[calib_kernel][:3]) / 3 * 1024
        ratio_r, ratio_w = fr / known_r, wr / known_w
        calib = {"kernel": calib_kernel, "known_read_bytes": known_r, "reported_fetch_bytes": fr, "ratio_read": ratio_r,
                 "known_write_bytes": known_w, "reported_write_bytes": wr, "ratio_write": ratio_w}
        print("calibration: FETCH_SIZE reports %.3f of the known read, WRITE_SIZE %.3f of the known write" % (ratio_r, ratio_w))
        if not (0.3 <= ratio_r <= 1.2 and 0.7 <= ratio_w <= 1.3):
            raise SystemExit("implausible calibration (%.3f / %.3f): the first three launches of %s are not the known-byte launches" % (ratio_r, ratio_w, calib_kernel))
    else:
        print("NO calibration launches found: falling back to ratio_read 0.5 / ratio_write 1.0")
    sig = (bench.get("roofline") or {}).get("traffic_signature")      # what bench.py matches an entry against (round 5 on)
    if sig is None:
        sig = {"workload": cfg.get("workload_id", "sage"), "nodes": n, "nnz": cfg["nnz"], "locality": cfg["locality"],
               "permuted_ids": cfg["permuted_ids"], "reorder": cfg["reorder"], "hidden": hidden, "dtype": bench["dtype"]}
        if "heads" in cfg:
            sig["heads"] = cfg["heads"]
    entries = []
    for name, v in bench["spmm_launch_table"].items():
        frag = v.get("kernel_fragment")
        kname = next((k for k in fetch if frag and frag in k), None)
        if kname is None:
            print("no counter rows for %s (%s)" % (name, frag))
            continue
        skip = 3 if kname == calib_kernel and calib else 0
        fl, wl = fetch[kname][skip:], write.get(kname, [])[skip:]
        per_step = max(len(fl) // (bench["steps"] + bench["warmup"]), 1)
        fl, wl = fl[per_step:], wl[per_step:]          # drop the first (cold) step
        if not fl or not wl:
            continue
        fk = sum(x for _, x, _ in fl) / len(fl)
        wk = sum(x for _, x, _ in wl) / len(wl)
        ns = sum(d for _, _, d in fl) / len(fl)
        e = {"round": args.round, "kernel_name": kname, "kernel_fragment": frag, "launch": name, "workload": sig, "build_stamp": stamp,
             "launches_averaged": len(fl), "fetch_size_kib": fk, "write_size_kib": wk, "ratio_read": ratio_r, "ratio_write": ratio_w,
             "hbm_bytes_per_launch": int(fk * 1024 / ratio_r + wk * 1024 / ratio_w), "avg_ns_under_pmc": ns,
             "algorithmic_bytes_per_launch": v["algorithmic_bytes"]}
        if kname in hit and kname in miss:
            h = sum(x for _, x, _ in hit[kname][skip:][per_step:])
            m = sum(x for _, x, _ in miss[kname][skip:][per_step:])
            e["l2_hit_rate"] = h / max(h + m, 1)
        entries.append(e)
        print("%-70s fetch %.0f KiB write %.0f KiB -> %.2f GB/launch, %.3f ms under PMC" % (name[:70], fk, wk, e["hbm_bytes_per_launch"] / 1e9, ns / 1e6))
    if args.write:
        path = os.path.join(root, "profiles", "traffic.json")
        try:
            with open(path) as f:
                cur = json.load(f)
        except (OSError, ValueError):
            cur = {}
        keep = [e for e in cur.get("entries", []) if e.get("workload") != sig]
        cur = {"_comment": "HBM-side traffic per launch from rocprofv3 --pmc passes over bench.py itself (tools/pmc_bench.sh, parsed "
                           "by tools/pmc_parse.py); hbm_bytes = FETCH_SIZE KiB * 1024 / ratio_read + WRITE_SIZE KiB * 1024 / ratio_write "
                           "with the ratios calibrated in the same pass on known-byte launches (`calibrations`).  bench.py uses an "
                           "entry only when workload, kernel_fragment AND build_stamp match the running build.",
               "calibrations": [c for c in cur.get("calibrations", []) if c.get("workload") != sig] + ([dict(calib, workload=sig, round=args.round, build_stamp=stamp)] if calib else []),
               "entries": keep + entries}
        with open(path, "w") as f:
            json.dump(cur, f, indent=1)
        print("wrote", path, "(%d entries for this workload, build %s)" % (len(entries), (stamp or "?")[:12]))


if __name__ == "__main__":
    main()
