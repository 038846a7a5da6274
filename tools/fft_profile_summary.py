#!/usr/bin/env python3
"""Per-PASS summary of the transform kernels from the rocprofv3 runs of tools/profile_fft.sh:

    python tools/fft_profile_summary.py <trace dir> <FETCH dir> <WRITE dir> <SQ dir> <tag>

A radix-2 transform at 2^20 is three launches of fft_pass_kernel<field> (7 + 7 + 6 layers); tools/fft_bench.py runs nothing else through
that kernel, so the k-th launch of a kernel name (in dispatch order) is pass k mod 3.  Per (kernel, pass): launches, mean duration from the
kernel trace, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH_SIZE counts 128-B requests at 64 B on gfx950:
MI355X_MICROARCH.md, HBM section), achieved GB/s = algorithmic bytes (one read + one write of the vector: 2 n s, s = 44 / 108 B) / duration,
traffic / algorithmic, and the SQ mix (VALU wave-instructions, LDS instructions, waiting and busy cycles).  Writes
gpurun_out/<tag>_fft_pass.csv and gpurun_out/traffic_fft_pass.json (stamped with the source hash bench.py checks)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = 3
N = 1 << 20


def fft_source_sha16():
    h = hashlib.sha256()
    for f in ("fft.hip.h", "fp.hip.h", "inst_field.hip"):
        h.update(open(os.path.join(ROOT, "pcd_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def rows_of(d, suffix):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def short(name):
    # fft_pass_kernel<pcd::Fp<pcd::F298B, true, false> >(...) -> fft_pass_kernel<F298B,true,false>
    n = name.split("(")[0].replace("pcd::", "").replace("void ", "").replace(" ", "")
    return n.replace("Fp<", "").replace(">>", ">")


def per_pass(rows, value_of, key="Kernel_Name", order="Dispatch_Id", counter=None):
    """{(kernel, pass): [values]} over the launches of every fft_pass kernel, pass = launch index mod 3 in dispatch order"""
    by_kernel = defaultdict(list)
    for r in rows:
        if "fft_pass_kernel" not in r.get(key, ""):
            continue
        if counter is not None and r.get("Counter_Name") != counter:
            continue
        by_kernel[short(r[key])].append((int(r.get(order) or 0), value_of(r)))
    out = defaultdict(list)
    for k, lst in by_kernel.items():
        lst.sort()
        for i, (_, v) in enumerate(lst):
            out[(k, i % PASSES)].append(v)
    return out


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    trace_d, fetch_d, write_d, sq_d, tag = sys.argv[1:6]
    dur = per_pass(rows_of(trace_d, "kernel_trace.csv"), lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)   # us
    fetch = per_pass(rows_of(fetch_d, "counter_collection.csv"), lambda r: float(r["Counter_Value"]), counter="FETCH_SIZE")
    write = per_pass(rows_of(write_d, "counter_collection.csv"), lambda r: float(r["Counter_Value"]), counter="WRITE_SIZE")
    sq_rows = rows_of(sq_d, "counter_collection.csv")
    sq = {c: per_pass(sq_rows, lambda r: float(r["Counter_Value"]), counter=c)
          for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_LDS")}
    regs = {}
    for r in sq_rows:
        if "fft_pass_kernel" in r.get("Kernel_Name", ""):
            regs[short(r["Kernel_Name"])] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    out_rows, js = [], {"n": N, "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> x2; KB -> bytes",
                        "source_sha16": fft_source_sha16(), "source": f"tools/profile_fft.sh {tag}: rocprofv3 on tools/fft_bench.py (coset FFT, n = 2^20), separate --pmc runs",
                        "kernels": {}}
    for (k, p) in sorted(dur):
        s = 44 if "F298" in k else 108
        alg = 2 * N * s
        d_us = mean(dur[(k, p)])
        f_kb, w_kb = mean(fetch.get((k, p), [])), mean(write.get((k, p), []))
        hbm = int(2 * f_kb * 1024 + w_kb * 1024) if f_kb is not None and w_kb is not None else None
        valu, cyc, wait = (mean(sq[c].get((k, p), [])) for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"))
        lds = mean(sq["SQ_INSTS_LDS"].get((k, p), []))
        row = {"kernel": k, "pass": p, "launches": len(dur[(k, p)]), "mean_us": round(d_us, 2), "algorithmic_bytes": alg,
               "achieved_GBs": round(alg / (d_us * 1e-6) / 1e9, 1), "FETCH_KB": None if f_kb is None else round(f_kb, 1),
               "WRITE_KB": None if w_kb is None else round(w_kb, 1), "hbm_bytes": hbm,
               "traffic_over_algorithmic": None if hbm is None else round(hbm / alg, 3),
               "fetch_x2_over_write": None if not w_kb else round(2 * f_kb / w_kb, 3),
               "SQ_INSTS_VALU": valu, "SQ_INSTS_LDS": lds, "wait_over_wave_cycles": None if not cyc else round(wait / cyc, 3),
               "vgpr_agpr_lds_scratch": regs.get(k)}
        out_rows.append(row)
        js["kernels"].setdefault(k, []).append({kk: row[kk] for kk in ("pass", "launches", "mean_us", "algorithmic_bytes", "achieved_GBs", "hbm_bytes",
                                                                      "traffic_over_algorithmic", "fetch_x2_over_write", "wait_over_wave_cycles")})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_fft_pass.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(out_rows[0].keys()) if out_rows else ["kernel"])
        w.writeheader()
        w.writerows(out_rows)
    json.dump(js, open(os.path.join(ROOT, "gpurun_out", "traffic_fft_pass.json"), "w"), indent=1)
    for r in out_rows:
        print(r)
    # the kernel_stats summary of the trace run, for the record
    for f in glob.glob(os.path.join(trace_d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh, open(os.path.join(ROOT, "gpurun_out", f"{tag}_fft_stats.csv"), "w") as oh:
            oh.writelines(fh.readlines()[:30])
        break


if __name__ == "__main__":
    main()
