import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import sdr_amd, oracle_lib as O
from test_chain_gpu import staged_oracle, rel_err
B = 2044
cps = float(np.float32(100e3) / np.float32(1000000))
for ntaps2, calls in ((4095, [6, 60]), (4095, [66]), (9001, [70]), (2500, [66])):
    taps2 = sdr_amd.fir_bandpass(ntaps2, 0.05, 0.2)
    stages = [(sdr_amd.hilbert(127), 63), (taps2, 0)]
    total = sum(calls)
    stream = torch.cuda.current_stream().cuda_stream
    gen = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
    chain = sdr_amd.Chain(stages, False)
    out = torch.full((total * B,), 777.0, dtype=torch.float32, device="cuda")
    done = 0
    for nb in calls:
        done += chain.generate_device(gen, nb, out.data_ptr() + 4 * done, stream=stream)
    torch.cuda.synchronize()
    fused = out[:done].cpu().numpy()
    gen2 = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
    chain2 = sdr_amd.Chain(stages, False)
    buf = torch.empty(max(calls) * B, dtype=torch.float32, device="cuda")
    parts = []
    for nb in calls:
        gen2.generate_device(nb, buf.data_ptr(), stream=stream)
        o, n_out = chain2.process_tensor(buf[: nb * B])
        torch.cuda.synchronize()
        parts.append(o[:n_out].cpu().numpy())
    unfused = np.concatenate(parts)
    x, _ = O.gen_real(total, B, 1.0, cps, 0.0, sdr_amd.WAVEFORMS["sine"])
    ref = staged_oracle(stages, x)
    print("fused vs unfused", rel_err(fused, unfused))
    for name, y in (("fused", fused), ("unfused", unfused)):
        d = np.abs(y - ref)
        bad = np.nonzero(d > 1e-4)[0]
        print(ntaps2, calls, name, "rel", rel_err(y, ref), "bad", len(bad), (bad.min(), bad.max()) if len(bad) else None, flush=True)
        if len(bad):
            # runs of bad indices
            br = np.nonzero(np.diff(bad) > 1)[0]
            starts = np.r_[bad[0], bad[br + 1]][:12]
            ends = np.r_[bad[br], bad[-1]][:12]
            print("   runs:", list(zip(starts.tolist(), ends.tolist())), flush=True)
