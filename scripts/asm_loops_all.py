#!/usr/bin/env python3
"""scripts/asm_loops_all.py ASM_FILE -- for every kernel of a `make -C sdr_amd/csrc asm` listing: the innermost loop that
holds the tile's global stores (the persistent tile loop) and the scalar loads inside it (see scripts/asm_loops.py)."""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
names = re.findall(r'\n(_ZN7sdr_amd\w+):', s)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
for name, d in zip(names, dem):
    i = s.index('\n' + name + ':')
    k = s.index('.Lfunc_end', i)
    b = s[i:k].split('\n')
    labels = {}
    for j, l in enumerate(b):
        m = re.match(r'(\.LBB\d+_\d+):', l.strip())
        if m:
            labels[m.group(1)] = j
    best = None
    for j, l in enumerate(b):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            lo = labels[m.group(1)]
            seg = b[lo:j]
            st = sum(1 for x in seg if 'global_store' in x)
            if st >= 15 and (best is None or j - lo < best[1] - best[0]):
                best = (lo, j)
    if best is None:
        continue
    seg = b[best[0]:best[1]]
    sl = [x.strip() for x in seg if 's_load' in x]
    rl = sum(1 for x in seg if 'v_readlane' in x)
    va = sum(1 for x in seg if re.match(r'\s*v_', x))
    print("%-62s loop %5d instr, valu %4d, v_readlane %3d, scalar loads %d %s" % (
        d.replace("sdr_amd::", "").replace("(sdr_amd::FftArgs)", "").replace("void ", "")[:62], best[1] - best[0], va, rl, len(sl),
        sorted(set(x.split(",")[-1].strip() for x in sl))))
