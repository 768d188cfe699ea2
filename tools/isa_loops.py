#!/usr/bin/env python3
"""Instruction-class counts of the loops of one kernel in a gfx950 assembly dump (inner loops, also those nested in an outer loop).

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only pam_amd/csrc/awfl_kernels.hip -o /tmp/awfl.s
       python tools/isa_loops.py /tmp/awfl.s <mangled-kernel-name substring>
Per loop: all instructions, VALU, FP64 VALU (4 cycles per wave64 instruction), v_readlane/v_writelane (SGPR spills), v_mov.  The
pair sweeps of the flux kernel make FIVE trips per round (rotating window slots): divide their rows by 5.
profiles/r03_isa_loops.txt is the output for the round-3 kernels."""
import re,sys
def stats(path,key):
    text=open(path).read()
    m=re.search(r'^(_Z\w*%s\w*):'%key,text,flags=re.M)
    body=text[m.start():]; body=body[:body.index('s_endpgm')]
    lines=body.split('\n'); blocks={}; cur=None; hdrof={}
    i=0
    while i<len(lines):
        l=lines[i]; mm=re.match(r'^(\.LBB\d+_\d+):\s*;?\s*(.*)$',l)
        if mm:
            cur=mm.group(1); blocks[cur]=[]; com=mm.group(2); j=i+1
            while j<len(lines) and lines[j].strip().startswith(';'): com+=' '+lines[j]; j+=1
            if 'Inner Loop Header' in com: hdrof[cur]=cur
            else:
                h=re.search(r'in Loop: Header=(BB\d+_\d+)',com)
                if h: hdrof[cur]='.L'+h.group(1)
        elif cur and l.strip() and not l.strip().startswith(';'): blocks[cur].append(l)
        i+=1
    loops={}
    for b,h in hdrof.items(): loops.setdefault(h,[]).extend(blocks[b])
    out=[]
    for h in sorted(loops,key=lambda x:int(x.split('_')[1])):
        b=loops[h]; v=[l for l in b if re.match(r'\s+v_',l)]
        if len(v)<50: continue
        out.append((h,'instr',len(b),'VALU',len(v),'f64',sum('_f64' in l for l in v),'lane',sum(('readlane' in l or 'writelane' in l) for l in v),'mov',sum('v_mov' in l for l in v)))
    ninstr=sum(1 for l in lines if re.match(r'\s+[sv]_|\s+global_|\s+ds_|\s+buffer_',l))
    return out,ninstr
if __name__=='__main__':
    o,n=stats(sys.argv[1],sys.argv[2])
    for x in o: print(*x)
    print('total instructions',n)
