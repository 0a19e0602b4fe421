#!/usr/bin/env python3
"""scripts/asm_loops.py ASM_FILE MANGLED_KERNEL -- loops of one kernel in a `make -C sdr_amd/csrc asm` listing, with the
number of global stores / loads, VALU / SALU / LDS instructions and the positions of SCALAR LOADS inside each: a kernel
argument re-loaded inside the persistent tile loop (s_load + s_waitcnt lgkmcnt(0), which also drains the LDS reads)
costs the overlap-save kernels 2-4 % (profiles/r03_device_state_ab.txt)."""
import re,sys
def kernel_asm(path, name):
    s=open(path).read()
    i=s.index('\n'+name+':')
    k=s.index('.Lfunc_end', i)
    return s[i:k].split('\n')
path=sys.argv[1]; name=sys.argv[2]
b=kernel_asm(path,name)
labels={}
for i,l in enumerate(b):
    m=re.match(r'(\.LBB\d+_\d+):',l.strip())
    if m: labels[m.group(1)]=i
seen=set()
for i,l in enumerate(b):
    m=re.search(r's_c?branch\w* (\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i and i-labels[m.group(1)]>300:
        lo=labels[m.group(1)]
        if lo in seen: continue
        seen.add(lo)
        seg=b[lo:i]
        st=sum(1 for x in seg if 'global_store' in x); ld=sum(1 for x in seg if 'global_load' in x); sl=[j+lo for j,x in enumerate(seg) if 's_load' in x]
        va=sum(1 for x in seg if re.match(r'\s*v_',x)); sa=sum(1 for x in seg if re.match(r'\s*s_',x)); ds=sum(1 for x in seg if re.match(r'\s*ds_',x))
        print(' loop',lo,'..',i,'len',i-lo,'stores',st,'loads',ld,'valu',va,'salu',sa,'lds',ds,'s_loads at',sl)
