#!/bin/bash
# scripts/profile.sh TAG WORKLOAD -- run on the GPU box (through gpurun).
# 1) rocprofv3 --kernel-trace --stats of bench.py  2) separate PMC passes for
# HBM read / write bytes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Output: gpurun_out/prof_TAG/ ; copy the summaries you want judged to profiles/.
set -u
TAG=${1:-r01}
WL=${2:-c2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the same command as the bench line it documents: default steps / warm-up
ARGS="bench.py --workload $WL --no-cpu-baseline ${EXTRA:-}"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/bench_trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/bench_trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/bench_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$OUT/bench_fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/bench_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$OUT/bench_write.log"; exit 1; }
python3 scripts/summarize_profile.py "$OUT" "$WL" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
