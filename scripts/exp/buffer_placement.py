"""scripts/exp/buffer_placement.py -- on the GPU box: the same call with the two buffers placed differently."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sdr_amd as S
stream = torch.cuda.current_stream().cuda_stream
n = 1 << 28
B = n * 8


def run(name, taps, xin, yout, what):
    st = S.FirState(taps, True)
    for _ in range(30):
        st.process_device(xin, n, yout, stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        st.process_device(xin, n, yout, stream=stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("%-14s %-52s in %#x out %#x: %.4f ms (%.3f)" % (name, what, xin, yout, ms, n * 16 / ms / 1e6 / 8000), flush=True)


filters = (("c2 fft4096", S.fir_lowpass(127, 0.1)), ("lp31 mfma", S.fir_lowpass(31, 0.1)))
# (1) two separate allocations, as bench.py makes them
x = torch.rand((n, 2), dtype=torch.float32, device="cuda") * 2 - 1
y = torch.empty_like(x)
for name, taps in filters:
    run(name, taps, x.data_ptr(), y.data_ptr(), "two torch allocations (bench.py)")
    run(name, taps, y.data_ptr(), x.data_ptr(), "the same two, roles swapped")
del y
# (2) a spacer allocation between them
sp = torch.empty(3 << 28, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
for name, taps in filters:
    run(name, taps, x.data_ptr(), y.data_ptr(), "a 768 MiB allocation made in between")
del x, y, sp
torch.cuda.empty_cache()
# (3) one pool holding both
pool = torch.empty(2 * B + (1 << 28), dtype=torch.uint8, device="cuda")
pool.view(torch.float32).uniform_(-1, 1)
base = pool.data_ptr()
for name, taps in filters:
    run(name, taps, base, base + B, "one allocation, out = in + 2 GiB")
    run(name, taps, base + B + (1 << 27), base, "one allocation, in = out + 2 GiB + 128 MiB")
