import subprocess, time, os
T = "scripts/exp/gen_trace"
one = 16 + 4 * 2044
import sys
for how in ["default", "default", "_exit", "exit", "quick_exit", "reset_then_signal"]:
    os.environ["SDR_DIE"] = how
    print("--", how)
    t0 = time.time()
    p = subprocess.Popen("%s freq=100k waveform=sine sample_rate=1M mode=real | head -c %d > /dev/null" % (T, one), shell=True, executable="/bin/bash", stderr=subprocess.PIPE)
    err = p.stderr.read().decode()
    t_eof = time.time()
    p.wait()
    t1 = time.time()
    print("spawn at 0; stderr EOF at %.1f ms; wait returned at %.1f ms" % (1e3 * (t_eof - t0), 1e3 * (t1 - t0)))
    for l in err.splitlines():
        if l.startswith("[gen abs"):
            a = float(l.split()[2].rstrip("]"))
            print("   %8.1f ms  %s" % (1e3 * (a - t0), l.split("]", 1)[1]))
