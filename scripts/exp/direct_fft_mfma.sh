#!/bin/bash
# scripts/exp/direct_fft_mfma.sh -- same box: the three kernel families per workload (profiles/r02_direct_vs_fft_vs_mfma.txt)
for wl in ${WLS:-c1big hilbc lp31 lp49 lp65 lp63r c2 c2real}; do
  for p in direct fft mfma; do
    timeout -k 10 120 python3 bench.py --workload $wl --path $p --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | python3 -c "
import json,sys
l=sys.stdin.readline()
if not l.strip():
    print('%-7s %-7s (not available)' % ('$wl','$p')); sys.exit(0)
d=json.loads(l); r=d['roofline']
print('%-7s %-7s %-30s %.4f ms (min %.4f)  %5.0f GB/s  hbm %.3f  fp32 %.3f (%.1f TFLOP/s)' % ('$wl','$p',d['config']['kernel'],r['kernel_ms'],r['kernel_ms_min'],r['hbm_gbs'],r['achieved_hbm'],r['achieved_valu'],r['valu_tflops']))"
  done
done
