"""Host logic of bench.py and the profile tooling (no GPU): the section-8(d) byte formulas, the single definition of
roofline.frac, and the tamper evidence of profiles/traffic.json -- an entry is used only when workload signature, kernel
instantiation AND build stamp match the running build."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.fixture()
def bench(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(mod, "ROOT", str(tmp_path))
    monkeypatch.setattr(mod, "build_stamp", lambda: "stamp-A")
    return mod


def test_algorithmic_byte_formulas(bench):
    # per edge F*s + 4 (+4 weighted), per row F*s + 8
    assert bench.alg_bytes(10, 3, 256, 2, 2, False) == 10 * 516 + 3 * 520
    assert bench.alg_bytes(10, 3, 256, 4, 4, True) == 10 * 1032 + 3 * 1032
    # products layer of BASELINE.md: 65.1 GB
    assert abs(bench.alg_bytes(123_718_280, 2_449_029, 256, 2, 2, False) / 1e9 - 65.1) < 0.05
    # fused GAT pass: nnz (F s + 4 + 4 heads) + n (F s + 8 + 8 heads)
    assert bench.gat_alg_bytes(10, 3, 256, 2, 8) == 10 * (512 + 4 + 32) + 3 * (512 + 8 + 64)


def test_kernel_fragments_name_the_instantiations(bench):
    assert bench.spmm_kernel_fragment(256, "torch.bfloat16", True, True) == \
        "spmm_csr_kernel<unsigned short, unsigned short, 8, 32, true, 4, true, false>"
    assert bench.spmm_kernel_fragment(47, "torch.bfloat16", False, False) == \
        "spmm_csr_kernel<unsigned short, unsigned short, 8, 8, false, 4, false, false>"
    assert bench.spmm_kernel_fragment(100, "torch.float32", False, False) == "spmm_csr_kernel<float, float, 4, 32, false, 4, false, false>"
    # the launch's average row length picks the kernel as spmm.hip does: RMAT-27 (16.8 edges per row, F = 128) runs the row-per-slot
    # kernel, fp32 rows of 16 / 32 lanes the flattened one, the products-sized graph (50.5 per row, bf16) the wave-per-row kernel
    assert bench.spmm_kernel_fragment(128, "torch.bfloat16", False, False, 16.8) == "spmm_rowslot_kernel<unsigned short, unsigned short, 8, 16, false, false>"
    assert bench.spmm_kernel_fragment(256, "torch.bfloat16", True, False, 3.2) == "spmm_rowslot_kernel<unsigned short, unsigned short, 8, 32, true, false>"
    assert bench.spmm_kernel_fragment(100, "torch.float32", False, False, 50.5) == "spmm_csr_flat_kernel<float, float, 4, 32, false, 4, false>"
    assert bench.spmm_kernel_fragment(256, "torch.float32", False, False, 50.5) == "spmm_csr_kernel<float, float, 4, 64, false, 4, false, false>"
    assert bench.spmm_kernel_fragment(256, "torch.bfloat16", True, False, 50.5) == "spmm_csr_kernel<unsigned short, unsigned short, 8, 32, true, 4, false, false>"
    # 8 heads x 32 bf16: 4 lanes per head, 8 heads per wavefront; 1 head x 48 with scores in the row padding: the in-row form
    assert bench.gat_kernel_fragment(8, 32, "torch.bfloat16", 2) == "gat2_kernel<unsigned short, unsigned short, 8, 32, 8, 4, 2, false, false>"
    assert bench.gat_kernel_fragment(1, 48, "torch.bfloat16", 0, packed=True) == "gat2_kernel<unsigned short, unsigned short, 8, 8, 1, 4, 0, true, false>"
    assert bench.gat_kernel_fragment(1, 48, "torch.bfloat16", 0, packed=False).endswith("0, false, false>")
    assert bench.gat_kernel_fragment(3, 24, "torch.bfloat16", 1) == "gat2_kernel<unsigned short, unsigned short, 8, 8, 2, 4, 1, false, false>"
    # the 8-head forward in its row-score form (t_j from the gathered row: dgll_hip_gat_fwd_rowscore)
    assert bench.gat_kernel_fragment(8, 32, "torch.bfloat16", 0, rowscore=True) == "gat2_kernel<unsigned short, unsigned short, 8, 32, 8, 4, 0, false, true>"


def _dom(frag, ms=5.0, b_alg=65.6e9):
    return {"count": 10, "avg_ms": ms, "nnz": 123_718_280, "algorithmic_bytes": b_alg, "algorithmic_GBps": b_alg / (ms * 1e-3) / 1e9,
            "kernel_fragment": frag}


def test_frac_is_a_fraction_counter_based_when_the_build_matches_and_stale_traffic_is_refused(bench, tmp_path):
    frag = "spmm_csr_kernel<unsigned short, unsigned short, 8, 32, true, 4, true, false>"
    sig = {"workload": "sage", "nodes": 1, "nnz": 2, "locality": 0.9, "permuted_ids": True, "reorder": "lpa", "hidden": 256, "dtype": "bf16"}
    entry = {"workload": dict(sig), "kernel_fragment": frag, "build_stamp": "stamp-A", "hbm_bytes_per_launch": 31.4e9, "round": "r03",
             "ratio_read": 0.531, "ratio_write": 1.0}
    with open(tmp_path / "profiles" / "traffic.json", "w") as f:
        json.dump({"entries": [entry]}, f)
    rec = bench.roofline_record(_dom(frag), sig, "k", 3.5e9, world=1)
    # `achieved` is the section-8(d) algorithmic rate (here above the peak: caches serve re-reads); `frac` is a FRACTION: the counter
    # traffic of this launch kind / live time / peak, because the entry was collected with this build
    assert abs(rec["frac_algorithmic"] - 65.6e9 / 5e-3 / 1e9 / 8000.0) < 1e-12 and rec["frac_algorithmic"] == rec["achieved"] / rec["peak"] > 1
    assert rec["traffic"] == 31.4e9 and abs(rec["frac_l2_miss_path"] - 31.4e9 / 5e-3 / 1e9 / 8000.0) < 1e-12
    assert rec["frac"] == rec["frac_l2_miss_path"] <= 1.0
    assert abs(rec["traffic_over_compulsory"] - 31.4 / 3.5) < 1e-9
    assert "counter bytes" in rec["frac_definition"] and "Infinity-Cache" in rec["frac_definition"] and "UPPER BOUND" in rec["frac_definition"]
    # another build of the library: the entry must not be used; `frac` falls back to min(1, algorithmic) and says so
    bench.build_stamp = lambda: "stamp-B"
    rec2 = bench.roofline_record(_dom(frag), sig, "k", 3.5e9, world=1)
    assert rec2["traffic"] is None and rec2["frac_l2_miss_path"] is None and rec2["traffic_over_compulsory"] is None
    assert rec2["frac"] == 1.0 and rec2["frac_algorithmic"] == rec["frac_algorithmic"] and "no counter entry" in rec2["frac_definition"]
    rec3 = bench.roofline_record(_dom(frag, ms=20.0), sig, "k", 3.5e9, world=1)
    assert rec3["frac"] == rec3["frac_algorithmic"] < 1.0
    bench.build_stamp = lambda: "stamp-A"
    # another workload / another kernel: no match either; multi-rank runs never take counter traffic
    assert bench.roofline_record(_dom(frag), dict(sig, nnz=3), "k", 0, world=1)["traffic"] is None
    assert bench.roofline_record(_dom(frag.replace("true, 4, true", "false, 4, false")), sig, "k", 0, world=1)["traffic"] is None
    assert bench.roofline_record(_dom(frag), sig, "k", 0, world=2)["traffic"] is None


def test_pmc_parse_applies_the_ratios_calibrated_in_the_same_pass(tmp_path):
    """tools/pmc_parse.py on a synthetic counter pass: three calibration launches with known bytes give ratio_read 0.5,
    the launch kind's bytes are FETCH / ratio_read + WRITE / ratio_write, the cold first step is dropped, and the entry
    carries the bench line's build stamp."""
    n, hidden = 1000, 256
    calib = "void dgll::spmm_csr_kernel<unsigned short, unsigned short, 8, 32, false, 4, false, false>(dgll::SpmmArgs)"
    kern = "void dgll::spmm_csr_kernel<unsigned short, unsigned short, 8, 32, true, 4, true, false>(dgll::SpmmArgs)"
    known_r, known_w = n * hidden * 2 + n * 12, n * hidden * 2
    hdr = '"Correlation_Id","Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'

    def rows(counter, calib_kib, kib_list):
        out, d = hdr, 1
        for _ in range(3):
            out += '%d,%d,"%s","%s",%f,0,1000\n' % (d, d, calib, counter, calib_kib); d += 1
        for v in kib_list:
            out += '%d,%d,"%s","%s",%f,0,5000\n' % (d, d, kern, counter, v); d += 1
        return out

    d = tmp_path / "pass"
    d.mkdir()
    # 5 steps (2 warm-up + 3 timed), one launch each; the first (cold) value is an outlier that must be dropped
    (d / "pmc_FETCH_SIZE.csv").write_text(rows("FETCH_SIZE", known_r * 0.5 / 1024, [999999, 100, 100, 100, 100]))
    (d / "pmc_WRITE_SIZE.csv").write_text(rows("WRITE_SIZE", known_w / 1024, [999999, 40, 40, 40, 40]))
    frag = "spmm_csr_kernel<unsigned short, unsigned short, 8, 32, true, 4, true, false>"
    line = {"dtype": "bf16", "steps": 3, "warmup": 2, "roofline": {"build_stamp": "stamp-X"},
            "config": {"workload_id": "sage", "nodes": n, "nnz": 5, "locality": 0.9, "permuted_ids": True, "reorder": "lpa", "hidden": hidden},
            "spmm_launch_table": {"spmm F=256 weighted": {"kernel_fragment": frag, "algorithmic_bytes": 123}}}
    (d / "pmc_FETCH_SIZE.json").write_text(json.dumps(line) + "\n")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_parse.py"), str(d)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "FETCH_SIZE reports 0.500 of the known read, WRITE_SIZE 1.000" in res.stdout
    # 100 KiB / 0.5 + 40 KiB / 1.0 = 240 KiB = 0.00 GB: check through the printed per-launch line
    assert "fetch 100 KiB write 40 KiB" in res.stdout


def test_compact_record_keeps_the_contract_fields_of_a_child_run(bench):
    full = {"metric": "m", "value": 2.5e10, "unit": "edges/s", "ms_per_step": 30.0, "steps": 10, "warmup": 3, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "config 4 ...", "nodes": 5, "nnz": 7, "heads": 8, "hidden": 256},
            "roofline": {"bound": "hbm", "achieved": 11000.0, "peak": 8000.0, "unit": "GB/s", "frac": 1.375, "traffic": None,
                         "frac_definition": "d", "frac_l2_miss_path": None, "kernel": "gat2", "kernel_fragment": "f", "avg_launch_ms": 6.3,
                         "launches_timed": 10, "compulsory_bytes_per_launch": 1},
            "cpu_baseline": {"value": 2e5, "unit": "edges/s", "cores": 256, "kind": "port", "sample": "s", "seconds_per_run": {}},
            "gat_pass_over_spmm": {"fwd": 1.2}, "spmm_launch_table": {"gat fwd": {"avg_ms": 5.5, "count": 10}}, "dense_launch_table": {}}
    c = bench.compact_record(full)
    assert c["value"] == 2.5e10 and c["ms_per_step"] == 30.0 and c["workload"] == "config 4 ..."
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(c["roofline"]) and "launches_timed" not in c["roofline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c["cpu_baseline"]) and "seconds_per_run" not in c["cpu_baseline"]
    assert c["gat_pass_over_spmm"] == {"fwd": 1.2} and c["gather_launch_ms"] == {"gat fwd": 5.5} and c["config"]["heads"] == 8
    assert "dense_launch_table" not in c and "spmm_launch_table" not in c


def test_compact_line_of_a_full_default_record_fits_the_drivers_reader(bench):
    """The round-5 default record (25 KB, profiles/r05_bench_default.json: headline + five child records) through compact_line: under
    LINE_LIMIT characters, valid JSON, contract fields kept, roofline / cpu_baseline numeric, <= 10 numbers per other workload."""
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    text = bench.compact_line(full, "/somewhere/bench_full.json")
    assert len(text) <= bench.LINE_LIMIT == 6000 and "\n" not in text
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "other_workloads"):
        assert key in d, key
    assert d["metric"] == full["metric"] and d["value"] == pytest.approx(full["value"], rel=1e-9) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-9)
    r = d["roofline"]
    assert r["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5) and r["frac"] <= 1 and r["frac_kind"] == "counter" and r["traffic"] == full["roofline"]["traffic"]
    assert r["frac_conservative"] == pytest.approx(full["roofline"]["frac_conservative"], rel=1e-5) and r["build_stamp"] == full["roofline"]["build_stamp"][:16]
    assert set(r) <= {"bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "traffic", "traffic_over_compulsory", "kernel_fragment",
                      "avg_launch_ms", "algorithmic_bytes_per_launch", "build_stamp", "frac_kind", "frac_conservative"}
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 256 and cb["kind"] == "port" and cb["oracle_c_openmp_csr_edges_per_s"] > 0 and len(cb["sample"]) <= 110
    assert set(d["other_workloads"]) == set(full["other_workloads"])
    for name, rec in d["other_workloads"].items():
        assert len(rec) <= 10 and all(isinstance(v, (int, float)) for v in rec.values()), name
        assert rec["ms_per_step"] == pytest.approx(full["other_workloads"][name]["ms_per_step"], rel=1e-4)
    assert d["other_workloads"]["gat"]["gat_fwd_ms"] == pytest.approx(5.7051) and d["other_workloads"]["minibatch"]["batches_per_s"] > 600
    assert d["other_workloads"]["sage_f32"]["roofline_frac"] <= 1.0 and d["full_record"] == "bench_full.json"
    # a record that would still be too long loses its optional tables first, never the contract fields
    fat = dict(full, spmm_launch_table={("launch kind %04d " % i) + "x" * 40: {"avg_ms": 1.0} for i in range(200)})
    slim = json.loads(bench.compact_line(fat, None))
    assert len(bench.compact_line(fat, None)) <= bench.LINE_LIMIT and "gather_launch_ms" not in slim and "roofline" in slim and "cpu_baseline" in slim
    # a failed child leaves a short error record
    bad = dict(full, other_workloads={"gat": {"error": "exit code 1", "stderr_tail": "x" * 600, "wall_seconds": 3.0, "command": "c"}})
    assert json.loads(bench.compact_line(bad, None))["other_workloads"]["gat"] == {"error": "exit code 1", "wall_seconds": 3.0}
