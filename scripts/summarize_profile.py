#!/usr/bin/env python3
"""Summarise a scripts/profile.sh output directory: per-kernel time from the
kernel-trace stats, and HBM bytes per launch of the FIR kernel from the PMC
passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced streaming read, so it is doubled."""
import csv
import glob
import json
import os
import sys

out, wl = sys.argv[1], sys.argv[2]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
                row["Name"][:70], row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"]))

# dominant FIR kernel = the fir_* kernel with the largest total time in the trace
dominant, best = "fir_", -1.0
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "fir_" in row["Name"] and "carry" not in row["Name"] and float(row["TotalDurationNs"]) > best:
                best, dominant = float(row["TotalDurationNs"]), row["Name"].replace("(anonymous namespace)::", "").split("(")[0]
print("== dominant kernel:", dominant)
res = {}
for key, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
    vals = []
    for f in find(pat):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if dominant in row.get("Kernel_Name", "").replace("(anonymous namespace)::", "") and row.get("Counter_Name") == key:
                    vals.append(float(row["Counter_Value"]))
    if vals:
        vals = vals[1:] or vals  # drop the warm-up launch
        res[key] = sum(vals) / len(vals)
print("== PMC (per launch of the dominant kernel, raw KiB) ==", res)
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    rd = res["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE = 1/2 of streamed bytes
    wr = res["WRITE_SIZE"] * 1024
    print("HBM read bytes/launch (FETCH_SIZE KiB x1024 x2): %.4g" % rd)
    print("HBM write bytes/launch (WRITE_SIZE KiB x1024):   %.4g" % wr)
    print("HBM traffic bytes/launch: %.4g" % (rd + wr))
    with open(os.path.join(out, "traffic_%s.json" % wl), "w") as f:
        json.dump({"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr,
                   "raw_kib": res, "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)",
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --workload %s` "
                             "(scripts/profile.sh), kernel %s" % (wl, dominant)}, f)
