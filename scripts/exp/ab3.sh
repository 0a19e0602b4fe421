#!/bin/bash
# scripts/exp/ab3.sh "LIB_A LIB_B ..." workloads... -- like ab.sh for any number of builds
set -u
LIBS=$1; shift
for round in 1 2 3; do
  for lib in $LIBS; do
    for w in "$@"; do
      SDR_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-40s %-8s %.4f ms (min %.4f) hbm %.3f' % ('$lib'.split('/')[-1], '$w', r['kernel_ms'], r['kernel_ms_min'], r['achieved_hbm']))
"
    done
  done
done
