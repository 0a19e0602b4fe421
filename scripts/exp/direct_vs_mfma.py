"""scripts/exp/direct_vs_mfma.py -- on the GPU box: fir_direct against fir_mfma at the lower end of the matrix-core range
(13..29 kept taps), dense complex and dense real, 2^28 samples, each pair measured twice in turn."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sdr_amd as S
n = 1 << 28
stream = torch.cuda.current_stream().cuda_stream
SWEEP = sys.argv[1] if len(sys.argv) > 1 else "dense"
for cplx in (True, False):
    w = 2 if cplx else 1
    x = torch.rand((n, w), dtype=torch.float32, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    for kept in {"dense": (13, 15, 16, 17, 19, 21, 23, 25, 29), "small": (2, 3, 4, 6, 8, 10, 12), "hilbert": (7, 11, 15, 23, 31, 35, 39, 47)}[SWEEP]:
        taps = S.hilbert(kept) if SWEEP == "hilbert" else S.fir_lowpass(kept, 0.1) if kept > 2 else S.fir_lowpass(3, 0.1)[:2].copy()
        line = "%-7s %s %2d taps:" % ("complex" if cplx else "real", SWEEP, kept)
        for rnd in range(2):
            for path in ("direct", "mfma"):
                try:
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
                    line += "  %s %.4f ms (%.3f)" % (path, ms, n * 4 * w * 2 / ms / 1e6 / 8000)
                except Exception as e:
                    line += "  %s n/a" % path
        print(line, flush=True)
    del x, y
