#!/bin/bash
# perf sweep: env assignments given as first arg list "A=1 B=2|A=2 ..." over workloads
IFS='|' read -ra ENVS <<< "$1"; shift
for wl in "$@"; do
  for e in "${ENVS[@]}"; do
    env $e timeout -k 10 120 python3 bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$wl [$e]', d['config']['kernel'], '%.0f Msamples/s' % d['value'], '%.0f GB/s' % r['achieved'], '%.3f ms' % r['kernel_ms'], 'min %.3f' % r['kernel_ms_min'], 'hbm %.2f' % r['frac'])
" || exit 1
  done
done
