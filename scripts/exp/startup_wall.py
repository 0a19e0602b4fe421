#!/usr/bin/env python3
"""scripts/exp/startup_wall.py -- wall time of whole processes (dynamic loading, HIP runtime start, teardown), beside
scripts/exp/startup's in-process stamps: where the 0.25 s of a one-packet `gen | hilbert` pipeline go."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def wall(cmd, n=3, **kw):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        subprocess.run(cmd, shell=True, executable="/bin/bash", stdout=subprocess.DEVNULL, **kw)
        ts.append(time.perf_counter() - t0)
    return min(ts)


B = os.path.join(ROOT, "blocks", "bin")
one = 16 + 4 * 2044
print("true (process spawn floor)                 %.1f ms" % (1e3 * wall("true")))
print("scripts/exp/startup (whole process)        %.1f ms" % (1e3 * wall(os.path.join(ROOT, "scripts/exp/startup"))))
print("hilbert < /dev/null                        %.1f ms" % (1e3 * wall("%s/hilbert < /dev/null" % B)))
print("hilbert help (no GPU touched)              %.1f ms" % (1e3 * wall("%s/hilbert help 2> /dev/null" % B)))
print("gen | head -c one packet                   %.1f ms" % (1e3 * wall("%s/gen freq=100k waveform=sine sample_rate=1M mode=real | head -c %d" % (B, one))))
print("gen | head -c one packet | hilbert         %.1f ms" % (1e3 * wall("%s/gen freq=100k waveform=sine sample_rate=1M mode=real | head -c %d | %s/hilbert" % (B, one, B))))
print("python -c pass                             %.1f ms" % (1e3 * wall(sys.executable + " -c pass")))
T = os.path.join(ROOT, "scripts", "exp", "gen_trace")
if os.path.exists(T):
    print("gen_trace | head -c one packet             %.1f ms" % (1e3 * wall("%s freq=100k waveform=sine sample_rate=1M mode=real 2>/dev/null | head -c %d" % (T, one))))
    print("gen | head -c one packet (again)           %.1f ms" % (1e3 * wall("%s/gen freq=100k waveform=sine sample_rate=1M mode=real | head -c %d" % (B, one))))
    t0 = time.perf_counter()
    p = subprocess.Popen("%s freq=100k waveform=sine sample_rate=1M mode=real | head -c %d > /dev/null" % (T, one), shell=True, executable="/bin/bash", stderr=subprocess.PIPE)
    err = p.stderr.read().decode()
    p.wait()
    print("traced pipeline wall %.1f ms; trace:\n%s" % (1e3 * (time.perf_counter() - t0), err))
