"""Aggregate a rocprofv3 output directory into per-kernel summaries (mean per launch):
    python tools/pmc_summary.py <dir> <out.csv>
Handles --kernel-trace --stats (kernel_stats.csv: copied through, top rows) and --pmc (counter_collection.csv: mean counter value
per kernel over its launches)."""
import csv, glob, os, sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = (r.get("Kernel_Name", "")[:100], r.get("Counter_Name", ""))
            acc[k][0] += 1
            acc[k][1] += float(r.get("Counter_Value", 0) or 0)
    for (k, c), (cnt, tot) in sorted(acc.items()):
        rows.append([k, c, cnt, tot / max(cnt, 1)])
if rows:
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "launches", "mean_value"])
        w.writerows(rows)
else:
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh, open(out, "w") as oh:
            for i, line in enumerate(fh):
                if i < 40:
                    oh.write(line)
        break
print(open(out).read()[:3000] if os.path.exists(out) else "no rocprofv3 csv found in " + d)
