"""scripts/exp/mfma_vs_fft.py -- on the GPU box: fir_mfma against fir_fft4096 around the upper end of the matrix-core range
(33..57 kept taps), dense complex, dense real and Hilbert-shaped real designs, 2^28 samples, alternated twice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sdr_amd as S
n = 1 << 28
stream = torch.cuda.current_stream().cuda_stream


def time_it(taps, cplx, path):
    w = 2 if cplx else 1
    st = S.FirState(taps, cplx, path=path)
    for _ in range(25):
        st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    return ms, n * 4 * w * 2 / ms / 1e6 / 8000, st.kernel_name


for cplx, kind in ((True, "dense"), (False, "dense"), (False, "hilbert")):
    w = 2 if cplx else 1
    x = torch.rand((n, w), dtype=torch.float32, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    for kept in (33, 37, 41, 45, 49, 53, 57):
        taps = S.hilbert(2 * kept - 1) if kind == "hilbert" else S.fir_lowpass(kept, 0.1)
        line = "%-7s %-7s %2d kept taps:" % ("complex" if cplx else "real", kind, kept)
        for rnd in range(2):
            for path in ("mfma", "fft"):
                try:
                    ms, frac, name = time_it(taps, cplx, path)
                    line += "  %s %.4f ms (%.3f)" % (path, ms, frac)
                except Exception as e:
                    line += "  %s n/a" % path
        print(line, flush=True)
    del x, y
