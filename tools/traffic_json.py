#!/usr/bin/env python3
"""profiles/traffic_msm_accumulate.json from the two PMC passes of tools/profile.sh:

    python tools/traffic_json.py gpurun_out/<tag>_fetch.csv gpurun_out/<tag>_write.csv "<what was profiled>"

HBM bytes per launch of the dominant kernel (the G1 MNT4-298 bucket accumulation) = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes;
FETCH_SIZE counts 128-B requests at 64 B on gfx950, MI355X_MICROARCH.md HBM section), stamped with the hash of the kernel sources:
bench.py reports the figure only while the sources are the ones it was measured on."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha16  # noqa: E402


def mean_of(path, counter):
    best = None
    for r in csv.DictReader(open(path)):
        if r["counter"] == counter and "msm_accumulate_kernel" in r["kernel"] and "G1Cfg" in r["kernel"] and "F298A" in r["kernel"]:
            if best is None or int(r["launches"]) > best[0]:
                best = (int(r["launches"]), float(r["mean_value"]), r["kernel"])
    if best is None:
        raise SystemExit(f"no msm_accumulate_kernel<G1 MNT4-298> row with {counter} in {path}")
    return best


def main():
    fetch, write, what = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    nf, f_kb, kname = mean_of(fetch, "FETCH_SIZE")
    nw, w_kb, _ = mean_of(write, "WRITE_SIZE")
    out = {"kernel": kname, "source": what, "launches": [nf, nw], "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md HBM section) -> x2; calibrated for wide "
                         "coalesced streams, this kernel gathers 88-B points, so the absolute is approximate",
           "hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024), "algorithmic_bytes_per_launch": (1 << 20) * 120,
           "source_sha16": source_sha16()}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_msm_accumulate.json"), "w"), indent=1)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):   # (on the GPU box only gpurun_out/ travels back)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic_msm_accumulate.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
