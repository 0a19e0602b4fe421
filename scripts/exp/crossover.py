"""Direct form vs overlap-save FFT per tap count (2^27 complex / 2^28 real samples, device-resident)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sdr_amd as S

def run(taps, cplx, n, path):
    os.environ["SDR_AMD_FFT"] = path
    st = S.FirState(taps, cplx)
    x = torch.rand((n, 2) if cplx else (n,), device="cuda") - 0.5
    y = torch.empty_like(x)
    for _ in range(10): st.process_tensor(x, y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(40): st.process_tensor(x, y)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / 40, st.kernel_name

for cplx, n in ((True, 1 << 27), (False, 1 << 28)):
    for kind in ("lowpass", "hilbert"):
        for nt in (15, 23, 31, 39, 47, 55, 63, 79, 95, 111, 127):
            taps = S.fir_lowpass(nt, 0.1) if kind == "lowpass" else S.hilbert(nt)
            d, kd = run(taps, cplx, n, "0")
            f, kf = run(taps, cplx, n, "1")
            print("%-7s %-7s %3d taps: direct %.3f ms  fft %.3f ms  -> %s" % ("complex" if cplx else "real", kind, nt, d, f, "fft" if f < d else "direct"), flush=True)
