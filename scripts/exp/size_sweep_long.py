"""scripts/exp/size_sweep_long.py -- on the GPU box: 4095-tap complex filter, ms per call against the call size, for the
kernel sdr_fir_create picks by itself (16384-point tiles when the call fills the chip, else the 4096-point forms) and
for each family pinned: where the per-call choice (fir_fft.hip: launch_fir_fft) should sit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import sdr_amd

rng = np.random.default_rng(0)
ntaps = int(sys.argv[1]) if len(sys.argv) > 1 else 4095
taps = sdr_amd.fir_bandpass(ntaps, 0.05, 0.2)
dev = torch.device("cuda:0")
for n in (1 << 18, 1 << 19, 1 << 20, 3 << 19, 1 << 21, 5 << 19, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 26):
    x = torch.rand((n, 2), dtype=torch.float32, device=dev) * 2 - 1
    y = torch.empty_like(x)
    line = "n = %9d" % n
    for path in ("auto", "fft4k", "fft16k"):
        st = sdr_amd.FirState(taps, True, path=path)
        for _ in range(10):
            st.process_tensor(x, y)
        torch.cuda.synchronize()
        reps = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            st.process_tensor(x, y)
        e1.record()
        torch.cuda.synchronize()
        line += "  %s %.4f ms" % (path, e0.elapsed_time(e1) / reps)
        del st
    print(line, flush=True)
    del x, y
