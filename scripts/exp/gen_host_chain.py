"""scripts/exp/gen_host_chain.py -- on the GPU box: is the fused generator chain (C5) bound by the host's serial
phase recurrence (one fp32 add + conditional subtract per generator block, sdr_capi.hip gen_next_table)?
Same 2^29 samples per call, generator blocks of 2044 (the bench's) up to 2^20 samples."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sdr_amd as S
stages = [(S.hilbert(127), 63), (S.fir_lowpass(127, 0.1), 0)]
stream = torch.cuda.current_stream().cuda_stream
for B in (2044, 4088, 16352, 65408, 1 << 20):
    nb = (1 << 29) // B
    chain = S.Chain(stages, False, channels=1, device=0)
    gen = S.Gen("sine", False, 1.0, 0.1, 0.0, B, device=0)
    y = torch.empty(nb * B, dtype=torch.float32, device="cuda")
    for _ in range(6):
        chain.generate_device(gen, nb, y.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 40
    for _ in range(K):
        chain.generate_device(gen, nb, y.data_ptr(), stream=stream)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("B %8d: %6d blocks per call, %.3f ms per call (host returned after %.3f ms per call)" % (B, nb, t / K * 1e3, t_host / K * 1e3), flush=True)
