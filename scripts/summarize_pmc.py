#!/usr/bin/env python3
"""Per-kernel averages of the SQ / GRBM counter passes collected by scripts/pmc.sh (argument: the output directory)."""
import csv, glob, os, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if any(k in name for k in ("fir_fft4096", "fir_fft16k", "fir_direct", "fir_mfma")):
            short = re.sub(r"^void sdr_amd::", "", name.replace("(anonymous namespace)::", "").split("(")[0])
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
# one table per kernel, the one with the most wave-cycles first; per-launch averages (first launch dropped)
def avg(v):
    v = v[1:] or v
    return sum(v) / len(v)
order = sorted(acc, key=lambda k: -avg(acc[k].get("SQ_WAVE_CYCLES", [0])))
for k in order:
    print("== %s  (%d launches per pass)" % (k, len(next(iter(acc[k].values())))))
    for c in sorted(acc[k]):
        print("%-26s %.6g" % (c, avg(acc[k][c])))
    a = {c: avg(v) for c, v in acc[k].items()}
    if "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"] > 0:
        wc = a["SQ_WAVE_CYCLES"]
        print("-- of the waves' cycles: issuing %.1f %%  stalled on issue %.1f %%  parked at a wait / barrier %.1f %%" % (
            100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("SQ_WAIT_ANY", 0) / wc))
        if "SQ_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a:
            # SQ_ACTIVE_INST_* count quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs
            simd_cycles = a["GRBM_GUI_ACTIVE"] / 8 * 1024 / 4
            cu_cycles = a["GRBM_GUI_ACTIVE"] / 8 * 256   # SQ_LDS_IDX_ACTIVE counts LDS-array cycles per CU
            print("-- busy share of the kernel's time: VALU issue %.1f %% of the SIMDs, LDS array %.1f %% of the CUs "
                  "(bank conflicts %.2f %% of its cycles)" % (
                      100 * a.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles, 100 * a.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles,
                      100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
