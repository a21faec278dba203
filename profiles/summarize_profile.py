"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
MINE = ("ms_lf_kernel", "mem_kernel", "occ_kernel", "extz_kernel", "read_totals", "occ_cnt", "occ_off", "phi_batch")


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")
            if any(m in name for m in MINE):
                print("%-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (name[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for tag in ("pmc_fetch", "pmc_write", "pmc_tcc"):
    print("== %s ==" % tag)
    agg = defaultdict(lambda: defaultdict(list))
    for f in find(tag + "/**/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if any(m in name for m in MINE):
                    agg[name.split("(")[0][:60]][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    for k, d in agg.items():
        for cname, vals in d.items():
            print("%-60s %-24s n=%d mean=%.1f min=%.1f max=%.1f" % (k, cname, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
