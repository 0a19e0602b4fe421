for r in 768 512 640 768 512; do
for w in c2 c1big; do
SDR_AMD_FFT_RESIDENT=$r SDR_AMD_LIB=$PWD/sdr_amd/lib/diag/libsdr_amd_d.so timeout -k 10 120 python3 bench.py --workload $w --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('resident $r %-8s %.4f ms (min %.4f) hbm %.3f' % ('$w', r['kernel_ms'], r['kernel_ms_min'], r['achieved_hbm']))
"
done
done
