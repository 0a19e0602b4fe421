#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from `make -C sdr_amd/csrc asm`'s
-Rpass-analysis=kernel-resource-usage output (optionally against an older file)."""
import re
import subprocess
import sys


def parse(path):
    out, name = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        if name is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                out[name][key] = int(m.group(1))
    return out


def main():
    new = parse(sys.argv[1])
    old = parse(sys.argv[2]) if len(sys.argv) > 2 else {}
    names = subprocess.run(["c++filt"] + list(new), capture_output=True, text=True).stdout.split("\n")
    for k, d in zip(new, names):
        d = d.replace("sdr_amd::", "").replace("(FftArgs)", "")
        line = "%-64s v%-4d s%-4d scr%-4d occ%-2d lds%-6d" % (d[:64], new[k].get("vgpr", -1), new[k].get("sgpr", -1),
                                                          new[k].get("scratch", -1), new[k].get("occ", -1), new[k].get("lds", -1))
        if k in old:
            line += "   was v%d scr%d occ%d" % (old[k].get("vgpr", -1), old[k].get("scratch", -1), old[k].get("occ", -1))
        print(line)


if __name__ == "__main__":
    main()
