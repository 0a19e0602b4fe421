"""scripts/exp/buffer_offset.py -- on the GPU box: does the distance between the input and the output buffer matter?
One allocation holds both; the output starts `delta` bytes past a whole 2^k distance from the input."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sdr_amd as S
stream = torch.cuda.current_stream().cuda_stream
n = 1 << 28
B = n * 8
pool = torch.empty(2 * B + (1 << 28), dtype=torch.uint8, device="cuda")
pool.view(torch.float32)[: (2 * B + (1 << 28)) // 4].uniform_(-1, 1)
base = pool.data_ptr()
base += (-base) % (1 << 21)   # 2 MiB aligned
for name, taps in (("c2 (127 taps, fir_fft4096)", S.fir_lowpass(127, 0.1)), ("lp31 (fir_mfma)", S.fir_lowpass(31, 0.1)), ("lp15 (fir_direct)", S.fir_lowpass(15, 0.1))):
    for delta in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 16 << 20, (64 << 20) + 65536, 128 << 20):
        st = S.FirState(taps, True)
        xin, yout = base, base + B + delta
        for _ in range(30):
            st.process_device(xin, n, yout, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            st.process_device(xin, n, yout, stream=stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print("%-28s out = in + 2 GiB + %10d B: %.4f ms (%.3f)" % (name, delta, ms, n * 16 / ms / 1e6 / 8000), flush=True)
