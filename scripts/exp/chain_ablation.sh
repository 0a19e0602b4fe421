#!/bin/bash
# NOTE: the compile-time knobs this script's libraries were built with (-DSDR_ABL_*, -DSDR_TICKET_MAP, SDR_SPREAD 4/5)
# were removed from fir_fft.hip once measured; they are in commit 218677c.  -DSDR_FFT_TICKETS and -DSDR_SPREAD=0..3 remain.
# scripts/exp/chain_ablation.sh -- on the GPU box: what each part of the 4096-point transform pair costs the
# ticketed persistent loop of fir_fft4096 (diagnostic libraries, garbage results, timing only).
set -u
OUT=$PWD/gpurun_out/chain_ablation.txt
: > "$OUT"
D=$PWD/sdr_amd/lib/diag
run() { # label, library, env...
    local label=$1 lib=$2; shift 2
    for w in ${WORKLOADS:-c2 c1big c3}; do
        env "$@" SDR_AMD_LIB=$lib timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-44s %-8s %.4f ms (min %.4f) %5.0f GB/s hbm %.3f' % ('$label', '$w', r['kernel_ms'], r['kernel_ms_min'], r['hbm_gbs'], r['achieved_hbm']))
" >> "$OUT" || { echo "$label $w failed"; exit 1; }
    done
}
run "release"                                  $PWD/sdr_amd/lib/libsdr_amd.so
run "no butterflies / twiddles (LDS + barriers)" $D/libsdr_amd_novalu.so
run "no LDS traffic (VALU + barriers)"          $D/libsdr_amd_nolds.so
run "no forward barrier"                        $D/libsdr_amd_nobar.so
run "no VALU, no LDS (barriers + memory)"       $D/libsdr_amd_novalu_nolds.so
run "no LDS, no forward barrier (VALU + memory)" $D/libsdr_amd_nolds_nobar.so
cat "$OUT"
