#!/bin/bash
# quick perf sweep on the GPU box: prints workload, R, Msamples/s, GB/s, ms
for wl in "$@"; do
  for R in 4 8 16; do
    SDR_AMD_R=$R timeout -k 10 120 python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$wl R=$R', d['config']['kernel'], '%.0f Msamples/s' % d['value'], '%.0f GB/s' % r['achieved'], '%.3f ms' % r['kernel_ms'], 'min %.3f' % r['kernel_ms_min'], 'valu %.2f' % r['valu_frac_of_157.3'])
" || exit 1
  done
done
