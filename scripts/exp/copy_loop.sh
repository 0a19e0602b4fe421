#!/bin/bash
# scripts/exp/copy_loop.sh -- on the GPU box: fir_fft4096's own loop reduced to loads, stores and tickets
# (SDR_AMD_FFT_ABLATE=1 in the diagnostic library) next to the copy skeleton on the same box.
set -u
D=$PWD/sdr_amd/lib/diag
run() {
    local label=$1 lib=$2; shift 2
    env "$@" SDR_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --workload ${W:-c2} --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-50s %.4f ms (min %.4f) %5.0f GB/s hbm %.3f  ms_per_step %.4f' % ('$label', r['kernel_ms'], r['kernel_ms_min'], r['hbm_gbs'], r['achieved_hbm'], d['ms_per_step']))
"
}
run "release" $PWD/sdr_amd/lib/libsdr_amd.so
run "diag, full arithmetic" $D/libsdr_amd_diag.so
run "copy-only loop with tickets" $D/libsdr_amd_diag.so SDR_AMD_FFT_ABLATE=1
run "copy-only loop, grid-strided 9/CU" $D/libsdr_amd_diag.so SDR_AMD_FFT_ABLATE=1 SDR_AMD_FFT_TICKETS=0
run "copy-only loop with tickets, resident 768 forced" $D/libsdr_amd_diag.so SDR_AMD_FFT_ABLATE=1 SDR_AMD_FFT_GRID=768
timeout -k 10 100 scripts/exp/membench10 | grep -E "32 KB.*(2 counters|one tile).*xch 0" | head -4
