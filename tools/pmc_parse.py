#!/usr/bin/env python3
"""Turn the counter files of tools/pmc_bench.sh into profiles/traffic.json entries.

    python tools/pmc_parse.py gpurun_out/<tag> [--round r02] [--write]

Per kernel name: mean FETCH_SIZE / WRITE_SIZE (KiB) over the launches of the timed + warm-up steps, skipping the first
step (cold caches) -- except for the three calibration launches (identity gather, `--calibrate`: the FIRST three launches
of the F = hidden unweighted kernel), which give the gfx950 FETCH_SIZE correction on a known byte count in the kernel's
own access pattern (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports 1/2 of a wide coalesced read).
hbm_bytes = FETCH_SIZE_KiB * 1024 * correction + WRITE_SIZE_KiB * 1024.
"""
import argparse
import csv
import json
import os
import sys
from collections import OrderedDict, defaultdict


def read(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append(r)
    return rows


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
    ap.add_argument("--round", default="r02")
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
    # calibration: the first three launches of the unweighted F = hidden kernel read n*hidden*esz + n*12 bytes (rows,
    # rowptr, col) and write n*hidden*esz
    calib_kernel = next((k for k in fetch if "spmm_csr_kernel" in k and ", false, 4, false>" in k and ", 32," in k), None)
    correction, calib = 2.0, None
    if calib_kernel and len(fetch[calib_kernel]) >= 3:
        known_r = n * hidden * esz + n * 12
        known_w = n * hidden * esz
        fr = sum(v for _, v, _ in fetch[calib_kernel][:3]) / 3 * 1024
        wr = sum(v for _, v, _ in write[calib_kernel][:3]) / 3 * 1024
        calib = {"kernel": calib_kernel, "known_read_bytes": known_r, "reported_fetch_bytes": fr, "ratio_read": fr / known_r,
                 "known_write_bytes": known_w, "reported_write_bytes": wr, "ratio_write": wr / known_w}
        print("calibration: FETCH_SIZE reports %.3f of the known read, WRITE_SIZE %.3f of the known write" % (fr / known_r, wr / known_w))
    entries = []
    sig = {"nodes": n, "nnz": cfg["nnz"], "locality": cfg["locality"], "permuted_ids": cfg["permuted_ids"],
           "reorder": cfg["reorder"], "hidden": hidden, "dtype": bench["dtype"]}
    table = bench["spmm_launch_table"]
    for kname in fetch:
        if "spmm_csr_kernel" not in kname:
            continue
        skip = 3 if kname == calib_kernel and calib else 0
        fl, wl = fetch[kname][skip:], write.get(kname, [])[skip:]
        per_step = max(len(fl) // (bench["steps"] + bench["warmup"]), 1)
        fl, wl = fl[per_step:], wl[per_step:]          # drop the first (cold) step
        if not fl or not wl:
            continue
        fk = sum(v for _, v, _ in fl) / len(fl)
        wk = sum(v for _, v, _ in wl) / len(wl)
        ns = sum(d for _, _, d in fl) / len(fl)
        weighted = ", true, 4," in kname
        extra = kname.rstrip().endswith("true>(dgll::SpmmArgs)")
        lpr = int(kname.split("<")[1].split(",")[3])
        e = {"round": args.round, "kernel_name": kname, "workload": sig, "weighted": weighted, "lanes_per_row": lpr,
             "launches_averaged": len(fl), "fetch_size_kib": fk, "write_size_kib": wk, "fetch_correction": correction,
             "hbm_bytes_per_launch": int(fk * 1024 * correction + wk * 1024), "avg_ns_under_pmc": ns}
        if kname in hit and kname in miss:
            h = sum(v for _, v, _ in hit[kname][skip:][per_step:])
            m = sum(v for _, v, _ in miss[kname][skip:][per_step:])
            e["l2_hit_rate"] = h / max(h + m, 1)
        # match to bench's launch table: width from lanes-per-row (F = hidden -> 32 lanes of 8 bf16), weights, epilogue
        for name, v in table.items():
            vecs = -(-v["feat"] // (16 // esz))
            l = 4
            while l < 64 and l < vecs:
                l <<= 1
            if l == lpr and v["weighted"] == weighted and bool(v["epilogue"]) == extra:
                e["feat"], e["epilogue"], e["launch"] = v["feat"], v["epilogue"], name
                e["algorithmic_bytes_per_launch"] = v["algorithmic_bytes"]
                break
        entries.append(e)
        print("%-110s fetch %.0f KiB write %.0f KiB -> %.2f GB/launch (%s)" % (kname[:110], fk, wk, e["hbm_bytes_per_launch"] / 1e9, e.get("launch")))
    out = {"calibration": calib, "entries": entries}
    print(json.dumps(out)[:400] + " ...")
    if args.write:
        path = os.path.join(root, "profiles", "traffic.json")
        try:
            with open(path) as f:
                cur = json.load(f)
        except (OSError, ValueError):
            cur = {}
        keep = [e for e in cur.get("entries", []) if e.get("workload") != sig]
        cur = {"_comment": "HBM-side traffic per launch from rocprofv3 --pmc passes over bench.py itself (tools/pmc_bench.sh, "
                           "parsed by tools/pmc_parse.py); hbm_bytes = FETCH_SIZE KiB * 1024 * correction + WRITE_SIZE KiB * 1024; "
                           "the correction (2.0) is checked per pass on the known-byte calibration launches (`calibrations`).",
               "calibrations": [c for c in cur.get("calibrations", []) if c.get("workload") != sig] + ([dict(calib, workload=sig, round=args.round)] if calib else []),
               "entries": keep + entries}
        with open(path, "w") as f:
            json.dump(cur, f, indent=1)
        print("wrote", path)


if __name__ == "__main__":
    main()
