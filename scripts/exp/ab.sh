#!/bin/bash
# scripts/exp/ab.sh LIB_A LIB_B [workloads...] -- on the GPU box: two builds of the library, alternated three times on
# the same box (box-to-box spread is ~1 %, more than most of the effects under test).
set -u
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for lib in $A $B; do
    for w in ${@:-c2 c1big}; do
      SDR_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $w ${BENCH_ARGS:-} --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-44s %-8s %.4f ms (min %.4f) %5.0f GB/s hbm %.3f' % ('$lib', '$w', r['kernel_ms'], r['kernel_ms_min'], r['hbm_gbs'], r['achieved_hbm']))
"
    done
  done
done
