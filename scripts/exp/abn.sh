#!/bin/bash
# scripts/exp/abn.sh WORKLOAD LIB... -- on the GPU box: several builds of the library, alternated three times on the
# same box (as ab.sh, for more than two candidates).  BENCH_ARGS passes extra bench.py flags (e.g. --path fft16k).
set -u
W=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    SDR_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $W ${BENCH_ARGS:-} --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-44s %-8s %.4f ms (min %.4f) %5.0f GB/s hbm %.3f' % ('$lib', '$W', r['kernel_ms'], r['kernel_ms_min'], r['hbm_gbs'], r['achieved_hbm']))
"
  done
done
