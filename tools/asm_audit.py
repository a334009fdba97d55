#!/usr/bin/env python3
"""Static audit of the hand-waited MFMA kernels' assembly (dense.hip): between an inline-asm buffer_load and the hand-placed
s_waitcnt that retires it, no instruction may read or overwrite the load's destination registers (the compiler does not know
those registers are still being written), and no buffer_load may take an SGPR operand that a VALU instruction (v_readlane /
v_readfirstlane / v_cmp) wrote fewer than 5 wait states earlier.  Loads retire in order: `s_waitcnt vmcnt(N)` leaves the N
newest vector-memory operations in flight.
    python tools/asm_audit.py          (compiles dgll_amd/csrc/dense.hip to assembly with hipcc; no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vregs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def sregs(tok):
    out = set()
    for m in re.finditer(r"s\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dense.s")
        sys.path.insert(0, ROOT)
        from dgll_amd.build import FLAGS                        # the library's own flags: the audit reads what ships
        subprocess.run(["hipcc"] + [f for f in FLAGS if f not in ("-fPIC", "-pthread")] +
                       ["-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        os.path.join(ROOT, "dgll_amd", "csrc", "dense.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    names = re.findall(r"^(_ZN4dgll\d+gemm_bf16_res_kernel\S+):", txt, re.M)
    total = 0
    for name in names:
        i = txt.index(name + ":")
        j = txt.index("s_endpgm", i)
        lines = [l.strip() for l in txt[i:j].splitlines()]
        lines = [l for l in lines if l and not l.startswith(";") and not l.endswith(":") and not l.startswith(".")]
        order, bad, loads, scratch = [], 0, 0, 0
        for k, t in enumerate(lines):
            if t.startswith("scratch_"):
                scratch += 1
            if t.startswith("buffer_load_dwordx4"):
                parts = t.split(",")
                dst, addr = vregs(parts[0]), vregs(parts[1])
                if any(addr & s for s in order) or any(dst & s for s in order):
                    bad += 1
                    print("VGPR hazard in", name, ":", t)
                used, states = sregs(",".join(parts[2:])), 0
                for q in range(k - 1, max(k - 12, -1), -1):
                    p = lines[q]
                    if p.startswith(("v_readlane", "v_readfirstlane", "v_cmp")) and (sregs(p.split(",")[0]) & used) and states < 5:
                        bad += 1
                        print("SGPR hazard in", name, ":", p, "->", t)
                    m = re.match(r"s_nop (\d+)", p)
                    states += (int(m.group(1)) + 1) if m else 1
                    if states >= 5:
                        break
                order.append(dst)
                loads += 1
                continue
            if t.startswith(("global_load", "global_store", "buffer_store", "scratch_")):
                if any(vregs(t) & s for s in order):
                    bad += 1
                    print("hazard in", name, ":", t)
                order.append(set())
                continue
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                order = order[len(order) - n:] if n > 0 else []
                continue
            if any(vregs(t) & s for s in order):
                bad += 1
                print("hazard in", name, ":", t)
        total += bad
        if scratch:
            print("SPILLS in", name, scratch)
            total += 1
    print("%d kernels audited, %d problems" % (len(names), total))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
