#!/bin/bash
# scripts/bench_all.sh -- run on the GPU box: one bench.py line per workload (steady state,
# default steps / warm-up, CPU baseline skipped), condensed into gpurun_out/bench_all.txt
set -u
OUT=$PWD/gpurun_out/bench_all.txt
: > "$OUT"
for w in c2 c2real lp255 hilbc c3 c3real c4 c4chain c5 c1 lp15 lp31; do
    timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-8s %-60s %8.0f Msamples/s %5.0f GB/s %.3f ms (min %.3f) hbm %.2f' % ('$w', d['config']['kernel'], d['value'], r['achieved'], r['kernel_ms'], r['kernel_ms_min'], r['frac']))
" >> "$OUT" || { echo "$w failed"; exit 1; }
done
cat "$OUT"
