"""scripts/exp/complex_as_real_planes.py -- on the GPU box: a complex stream through fir_mfma as complex samples (P = 2, W = 2)
against the same bytes as a real stream of twice the length with zero-interleaved taps (P = 2, W = 1, stride 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sdr_amd as S
stream = torch.cuda.current_stream().cuda_stream
for log2n in (28, 27):
    n = 1 << log2n
    x = torch.rand((n, 2), dtype=torch.float32, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    for T in (9, 15, 21, 31, 41):
        g = S.fir_lowpass(T, 0.1)
        g2 = np.zeros(2 * T - 1, np.float32); g2[::2] = g
        line = "2^%d complex, %2d taps:" % (log2n, T)
        for rnd in range(2):
            for label, st, m in (("complex", S.FirState(g, True, path="mfma"), n), ("as real x2", S.FirState(g2, False, path="mfma"), 2 * n)):
                for _ in range(25):
                    st.process_device(x.data_ptr(), m, y.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    st.process_device(x.data_ptr(), m, y.data_ptr(), stream=stream)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 40
                line += "  %s [%s] %.4f ms (%.3f)" % (label, st.kernel_name, ms, n * 16 / ms / 1e6 / 8000)
        print(line, flush=True)
