"""scripts/exp/channels_long.py -- on the GPU box: 4095-tap complex filter over many channels, ms per call for the library's own
choice and each overlap-save family pinned: calls made of many short channels are all edge tiles for the 16384-point kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sdr_amd

taps = sdr_amd.fir_bandpass(4095, 0.05, 0.2)
for C, n in ((64, 1 << 20), (1024, 40000), (4096, 40000), (16, 1 << 22)):
    x = torch.rand((C, n, 2), dtype=torch.float32, device="cuda") * 2 - 1
    y = torch.empty_like(x)
    line = "%5d channels x %8d samples" % (C, n)
    for path in ("auto", "fft4k", "fft16k"):
        st = sdr_amd.FirState(taps, True, channels=C, path=path)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
        e1.record()
        torch.cuda.synchronize()
        line += "  %s %.3f ms" % (path, e0.elapsed_time(e1) / 20)
        del st
    print(line, flush=True)
    del x, y
