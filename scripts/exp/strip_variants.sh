#!/bin/bash
# NOTE: the compile-time knobs this script's libraries were built with (-DSDR_ABL_*, -DSDR_TICKET_MAP, SDR_SPREAD 4/5)
# were removed from fir_fft.hip once measured; they are in commit 218677c.  -DSDR_FFT_TICKETS and -DSDR_SPREAD=0..3 remain.
# scripts/exp/strip_variants.sh -- on the GPU box: fir_fft4096's loop with the transform stripped (no butterflies,
# no LDS traffic; barriers, product, tickets, loads and stores stay) under variations of its memory side, next to
# the copy skeleton (membench10) on the same box.
set -u
OUT=$PWD/gpurun_out/strip_variants.txt
: > "$OUT"
D=$PWD/sdr_amd/lib/diag
run() { # label, library, env...
    local label=$1 lib=$2; shift 2
    for w in ${WORKLOADS:-c2}; do
        env "$@" SDR_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-50s %-8s %.4f ms (min %.4f) %5.0f GB/s hbm %.3f' % ('$label', '$w', r['kernel_ms'], r['kernel_ms_min'], r['hbm_gbs'], r['achieved_hbm']))
" >> "$OUT" || { echo "$label $w failed"; exit 1; }
    done
}
run "release"                                   $PWD/sdr_amd/lib/libsdr_amd.so
run "stripped (spread 2, 2 counters)"           $D/libsdr_amd_novalu_nolds.so
run "stripped, prefetch in one burst"           $D/libsdr_amd_strip_spread0.so
run "stripped, prefetch in 8 shares"            $D/libsdr_amd_strip_spread3.so
run "stripped, 8 counters"                      $D/libsdr_amd_strip_nc8.so
run "stripped, no tickets (grid 9/CU)"          $D/libsdr_amd_novalu_nolds.so SDR_AMD_FFT_TICKETS=0
run "stripped, copy only (ablate 1, no tickets)" $D/libsdr_amd_novalu_nolds.so SDR_AMD_FFT_ABLATE=1
cat "$OUT"
timeout -k 10 100 scripts/exp/membench10 | grep -E "32 KB" | head -40
