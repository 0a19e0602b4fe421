import sys, os
sys.path.insert(0, "/root/repo")
import torch, sdr_amd
n = 1 << 27
x = torch.rand((n, 2), dtype=torch.float32, device="cuda") * 2 - 1
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
for ntaps in [int(a) for a in sys.argv[1:]] or (1025, 2049, 2050, 3001, 4095, 4098, 5000, 6143, 7000, 8191):
    taps = sdr_amd.fir_lowpass(ntaps, 0.1)
    st = sdr_amd.FirState(taps, True)
    for _ in range(30):
        st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    print("%5d taps %-60s %.4f ms per 2^27 complex  %5.0f GB/s  %.3f of HBM" % (ntaps, st.kernel_name, ms, n * 16 / ms / 1e6, n * 16 / ms / 1e6 / 8000), flush=True)
