"""Direct-form kernel: outputs per lane R = 4 / 8 per tap count, against the FFT path
(2^27 complex / 2^28 real samples, device-resident, 60 timed launches after 20)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sdr_amd as S

def run(taps, cplx, n, path, R):
    os.environ["SDR_AMD_FFT"] = path
    os.environ["SDR_AMD_R"] = str(R)
    st = S.FirState(taps, cplx)
    x = torch.rand((n, 2) if cplx else (n,), device="cuda") - 0.5
    y = torch.empty_like(x)
    for _ in range(20): st.process_tensor(x, y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(60): st.process_tensor(x, y)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / 60

for cplx, n in ((True, 1 << 27), (False, 1 << 28)):
    for kind in ("lowpass", "hilbert"):
        for nt in (7, 15, 23, 31, 39, 47, 63, 79, 95, 111):
            if kind == "lowpass" and nt > 63: continue
            taps = S.fir_lowpass(nt, 0.1) if kind == "lowpass" else S.hilbert(nt)
            r4 = run(taps, cplx, n, "0", 4)
            r8 = run(taps, cplx, n, "0", 8)
            f = run(taps, cplx, n, "1", 8)
            best = min((r4, "R=4"), (r8, "R=8"), (f, "fft"))[1]
            print("%-7s %-7s %3d taps: R=4 %.3f  R=8 %.3f  fft %.3f ms -> %s" % ("complex" if cplx else "real", kind, nt, r4, r8, f, best), flush=True)
