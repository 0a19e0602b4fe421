"""scripts/exp/size_sweep.py -- on the GPU box: fir_fft4096 launch time against the call size, to split a
launch into its fixed part (set-up of a persistent workgroup, edge tiles, ramp and tail) and its per-sample
part.  t(n) = a + b n, fitted on n = 2^24 .. 2^28 complex samples, 127-tap lowpass; both per-launch events
and one pair of events around 30 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import sdr_amd

taps = sdr_amd.fir_lowpass(127, 0.1)
dev = torch.device("cuda:0")
rows = []
for log2n in (24, 25, 26, 27, 28):
    n = 1 << log2n
    x = torch.rand((n, 2), dtype=torch.float32, device=dev) * 2 - 1
    y = torch.empty_like(x)
    st = sdr_amd.FirState(taps, True)
    for _ in range(10):
        st.process_tensor(x, y)
    torch.cuda.synchronize()
    reps = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        st.process_tensor(x, y)
    e1.record()
    torch.cuda.synchronize()
    t_block = e0.elapsed_time(e1) / reps
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        st.process_tensor(x, y)
        ev[i + 1].record()
    torch.cuda.synchronize()
    t_each = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))
    rows.append((n, t_block, t_each))
    print("n = 2^%d  %s  %.4f ms per launch (30 launches between two events)  %.4f ms (event after every launch)  %5.0f GB/s"
          % (log2n, st.kernel_name, t_block, t_each, n * 16 / t_block / 1e6), flush=True)
    del x, y, st
ns = np.array([r[0] for r in rows], float)
ts = np.array([r[1] for r in rows], float)
b, a = np.polyfit(ns, ts, 1)
print("fit t = a + b n: a = %.1f us fixed per launch, b -> %.0f GB/s asymptotic" % (a * 1e3, 16 / b / 1e6))
