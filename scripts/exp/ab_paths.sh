#!/bin/bash
# scripts/exp/ab_paths.sh "PATHS" [workloads...] -- on the GPU box: kernel families of ONE build (bench.py --path), alternated
# three times on the same box.  Optional SDR_AMD_LIB picks the build.
set -u
P=$1; shift
for round in 1 2 3; do
  for path in $P; do
    for w in ${@:-c2 c1big}; do
      timeout -k 10 120 python3 bench.py --workload $w --path $path --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-8s %-8s %-40s %.4f ms (min %.4f) %5.0f GB/s hbm %.3f' % ('$path', '$w', d['config']['kernel'], r['kernel_ms'], r['kernel_ms_min'], r['hbm_gbs'], r['achieved_hbm']))
"
    done
  done
done
