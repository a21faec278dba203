"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
MINE = ("ms_lf_kernel", "mem_kernel", "occ_kernel", "extz_kernel", "extz_lds_kernel", "align_kernel", "pack_kernel", "read_totals", "occ_cnt", "occ_off", "phi_batch")


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")
            if any(m in name for m in MINE):
                print("%-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (name[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
print("== full-size launches only (kernel trace; launches with the largest grid of each kernel) ==")
FULL = {}
for f in find("trace/**/*kernel_trace.csv"):
    per = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if any(m in name for m in MINE):
                per[name.split("(")[0][:60]].append((int(row["Grid_Size_X"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    for k, v in per.items():
        g = max(x[0] for x in v)
        d = [x[1] for x in v if x[0] == g]
        FULL[k] = sum(d) / len(d)
        print("%-60s grid=%d launches=%d avg_ns=%.0f min_ns=%d max_ns=%d" % (k, g, len(d), FULL[k], min(d), max(d)))
for tag in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_sq"):
    print("== %s ==" % tag)
    agg = defaultdict(lambda: defaultdict(list))
    for f in find(tag + "/**/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if any(m in name for m in MINE):
                    agg[name.split("(")[0][:60]][row.get("Counter_Name")].append((int(row.get("Grid_Size", 0)), float(row.get("Counter_Value", 0))))
    for k, d in agg.items():
        for cname, gv in d.items():
            g = max(x[0] for x in gv)
            vals = [x[1] for x in gv if x[0] == g]          # full-size launches only
            print("%-60s %-24s n=%d mean=%.1f min=%.1f max=%.1f" % (k, cname, len(vals), sum(vals) / len(vals), min(vals), max(vals)))

# HBM traffic of the dominant kernel for bench.py's roofline.traffic (bytes per launch; FETCH_SIZE/WRITE_SIZE are in KB and
# count 64-byte requests for this kernel's per-lane gathers, see profiles/README.md)
import json
vals = {}
for tag, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    v = []
    for f in find(tag + "/**/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "ms_lf_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == cname:
                    v.append((int(row.get("Grid_Size", 0)), float(row["Counter_Value"])))
    if v:
        g = max(x[0] for x in v)
        v = [x[1] for x in v if x[0] == g]
    vals[cname] = sum(v) / len(v) if v else None
try:
    b = json.load(open(os.path.join(out, "bench_trace.json")))
    import re
    mm = re.search(r"n=(\d+), r=(\d+)", b["config"]["workload"])
    doc = {"kernel": "ms_lf_kernel", "n": int(mm.group(1)), "r": int(mm.group(2)), "reads": b["config"]["reads_per_gpu"],
           "read_len": b["config"]["read_len"], "fetch_bytes": vals["FETCH_SIZE"] * 1024, "write_bytes": vals["WRITE_SIZE"] * 1024,
           "avg_launch_ms_rocprof": None, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, " + os.path.basename(out.rstrip("/"))}
    for k, v in FULL.items():
        if "ms_lf_kernel" in k:
            doc["avg_launch_ms_rocprof"] = v / 1e6
    json.dump(doc, open(os.path.join(out, "traffic_ms_lf.json"), "w"), indent=1)
    print("traffic:", doc)
except Exception as e:
    print("traffic summary failed:", e)
