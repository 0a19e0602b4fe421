#!/bin/bash
# scripts/bench_all.sh -- run on the GPU box: one bench.py line per workload (steady state,
# default steps / warm-up, CPU baseline skipped), condensed into gpurun_out/bench_all.txt
set -u
OUT=$PWD/gpurun_out/bench_all.txt
: > "$OUT"
for w in c2 c2real c1big lp255 hilbc c3 c3real c4 c4chain c5 c1 lp5 lp15 lp15r lp17 lp31 lp41 lp49 lp65 lp31r lp47r lp63r hilb63 hilb95; do
    timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-8s %-50s %8.0f Msamples/s %5.0f GB/s %.4f ms (min %.4f) hbm %.3f fp32 %.3f bound %s' % ('$w', d['config']['kernel'], d['value'], r['hbm_gbs'], r['kernel_ms'], r['kernel_ms_min'], r['achieved_hbm'], r['achieved_valu'], r['bound']))
" >> "$OUT" || { echo "$w failed"; exit 1; }
done
cat "$OUT"
