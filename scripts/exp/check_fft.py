import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, sdr_amd, oracle_lib as O
rng = np.random.default_rng(0)
def rel(a, b):
    a = a.astype(np.complex128); b = b.astype(np.complex128)
    return np.max(np.abs(a - b)) / np.sqrt(np.mean(np.abs(b) ** 2))
for cplx in (True, False):
    for nt, kind in ((127, "lp"), (64, "lp"), (255, "lp"), (1000, "lp"), (127, "hil"), (3000, "lp"), (4095, "lp"), (4096, "lp"), (9001, "lp")):
        taps = sdr_amd.fir_lowpass(nt, 0.1) if kind == "lp" else sdr_amd.hilbert(nt)
        n = 50001
        x = rng.uniform(-1, 1, n * (2 if cplx else 1)).astype(np.float32)
        x = x.view(np.complex64) if cplx else x
        st = sdr_amd.FirState(taps, cplx)
        cuts = [0, 3, 5000, 5001, 20000, n]
        y = np.concatenate([st.process(x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
        ref = O.fir(taps, x)
        print(st.kernel_name, "cplx" if cplx else "real", nt, kind, "err %.3g" % rel(y, ref), flush=True)
