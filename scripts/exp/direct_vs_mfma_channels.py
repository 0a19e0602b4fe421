import sys
sys.path.insert(0, "/root/repo")
import torch
import sdr_amd as S
C, n = 64, 1 << 22
stream = torch.cuda.current_stream().cuda_stream
x = torch.rand((C, n), dtype=torch.float32, device="cuda") * 2 - 1
y = torch.empty_like(x)
for name, taps in (("lowpass 15", S.fir_lowpass(15, 0.1)), ("hilbert 31", S.hilbert(31)), ("lowpass 5", S.fir_lowpass(5, 0.1))):
    line = "%d ch x 2^22 real, %s:" % (C, name)
    for rnd in range(2):
        for path in ("direct", "mfma"):
            st = S.FirState(taps, False, channels=C, path=path)
            for _ in range(25):
                st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                st.process_device(x.data_ptr(), n, y.data_ptr(), stream=stream)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 40
            line += "  %s %.4f ms (%.3f)" % (path, ms, C * n * 8 / ms / 1e6 / 8000)
    print(line, flush=True)
# short calls, one channel: 2^16 and 2^20 samples
for m in (1 << 16, 1 << 20):
    for name, taps in (("lowpass 15", S.fir_lowpass(15, 0.1)),):
        line = "1 ch x %d real, %s:" % (m, name)
        for rnd in range(2):
            for path in ("direct", "mfma"):
                st = S.FirState(taps, False, path=path)
                for _ in range(50):
                    st.process_device(x.data_ptr(), m, y.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    st.process_device(x.data_ptr(), m, y.data_ptr(), stream=stream)
                e1.record(); torch.cuda.synchronize()
                line += "  %s %.2f us" % (path, e0.elapsed_time(e1) / 200 * 1e3)
        print(line, flush=True)
