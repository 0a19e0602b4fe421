#!/bin/bash
# scripts/exp/gen_ablation.sh -- on the GPU box: C5 with diag builds of the library (make diag EXTRA=-D... DIAG_NAME=...)
run() { # lib [ENV=..]
  lib=$1; shift
  env "$@" SDR_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload c5 --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-44s %-28s %.4f ms (min %.4f)' % ('$lib', '$*', r['kernel_ms'], r['kernel_ms_min']))
"
}
run sdr_amd/lib/libsdr_amd.so
run sdr_amd/lib/diag/libsdr_amd_g4.so SDR_AMD_FFT_RESIDENT=1024
run sdr_amd/lib/diag/libsdr_amd_g4.so SDR_AMD_FFT_RESIDENT=768
run sdr_amd/lib/libsdr_amd.so
