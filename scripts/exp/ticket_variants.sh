#!/bin/bash
# NOTE: the compile-time knobs this script's libraries were built with (-DSDR_ABL_*, -DSDR_TICKET_MAP, SDR_SPREAD 4/5)
# were removed from fir_fft.hip once measured; they are in commit 218677c.  -DSDR_FFT_TICKETS and -DSDR_SPREAD=0..3 remain.
# scripts/exp/ticket_variants.sh -- on the GPU box: the ordered-ticket tile hand-out of fir_fft4096 against
# its variants (diagnostic libraries built by `make -C sdr_amd/csrc diag EXTRA=... DIAG_NAME=...`).
set -u
OUT=$PWD/gpurun_out/ticket_variants.txt
: > "$OUT"
D=$PWD/sdr_amd/lib/diag
run() { # label, library, env...
    local label=$1 lib=$2; shift 2
    for w in ${WORKLOADS:-c2 c1big c4}; do
        env "$@" SDR_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-34s %-8s %.4f ms (min %.4f) %5.0f GB/s hbm %.3f' % ('$label', '$w', r['kernel_ms'], r['kernel_ms_min'], r['hbm_gbs'], r['achieved_hbm']))
" >> "$OUT" || { echo "$label $w failed"; exit 1; }
    done
}
run "release (8 counters, spread 2)" $PWD/sdr_amd/lib/libsdr_amd.so
run "no tickets, grid 9/CU"          $D/libsdr_amd_diag.so SDR_AMD_FFT_TICKETS=0
run "tickets, 2 counters"            $D/libsdr_amd_nc2.so
run "tickets, 32 counters"           $D/libsdr_amd_nc32.so
run "tickets, counter = (bid/8)%8"   $D/libsdr_amd_map1.so
run "tickets, spread 0"              $D/libsdr_amd_spread0.so
run "tickets, spread 1"              $D/libsdr_amd_spread1.so
run "tickets, spread 3"              $D/libsdr_amd_spread3.so
run "tickets, grid 2/CU"             $D/libsdr_amd_diag.so SDR_AMD_FFT_RESIDENT=512
run "tickets, grid 6/CU"             $D/libsdr_amd_diag.so SDR_AMD_FFT_RESIDENT=1536
cat "$OUT"
