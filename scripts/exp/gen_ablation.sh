#!/bin/bash
# scripts/exp/gen_ablation.sh -- on the GPU box: C5 with diag builds of the library (make diag EXTRA=-D... DIAG_NAME=...)
run() { # lib [ENV=..]
  lib=$1; shift
  env "$@" SDR_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload c5 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-44s %-28s %.4f ms (min %.4f)' % ('$lib', '$*', r['kernel_ms'], r['kernel_ms_min']))
"
}
for r in 256 512 768; do
run sdr_amd/lib/diag/libsdr_amd_abl3.so SDR_AMD_FFT_RESIDENT=$r
run sdr_amd/lib/diag/libsdr_amd_full.so SDR_AMD_FFT_RESIDENT=$r
done
