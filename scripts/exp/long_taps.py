"""scripts/exp/long_taps.py -- on the GPU box: the partitioned overlap-save variants beyond 4096 taps
(parts + 1 transforms per tile, spectra and earlier windows read from L2 mid-tile), ms per launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sdr_amd as S
n = 1 << 26
x = (torch.rand((n, 2), device="cuda") * 2 - 1).float()
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
PATHS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["auto"]
for ntaps, path in [(t, p) for t in (2050, 4095, 6143, 8191, 8192, 16383, 32767) for p in PATHS]:
    if path == "fft16k" and ntaps > 8192:
        continue
    taps = S.fir_bandpass(ntaps, 0.05, 0.2)
    f = S.FirState(taps, True, device=0, path=path)
    for _ in range(5):
        f.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for _ in range(K):
        f.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
    e1.record()
    torch.cuda.synchronize()
    print("%6d taps (%2d partitions) %-8s %-24s %.3f ms per 2^26 complex samples" % (ntaps, (ntaps + 2047) // 2048, path, f.kernel_name, e0.elapsed_time(e1) / K), flush=True)
