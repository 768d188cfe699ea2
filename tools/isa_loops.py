#!/usr/bin/env python3
"""Instruction-class counts of the inner loops of one kernel in a gfx950 assembly dump.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only pam_amd/csrc/awfl_kernels.hip -o /tmp/awfl.s
       python tools/isa_loops.py /tmp/awfl.s <mangled-kernel-name substring>
Per inner loop (one face / cell of a sweep per iteration): all instructions, VALU, FP64 VALU (4 cycles per wave64
instruction), v_rcp_f64 (quarter rate), v_readlane/v_writelane (SGPR spills), v_mov, vector-memory, LDS, scalar loads.
profiles/r02_isa_loops.txt is the output for the round-2 kernels."""
import re, sys
text = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*%s\w*):' % sys.argv[2], text, flags=re.M)
body = text[m.start():]
body = body[:body.index('s_endpgm')]
loops = {}
cur = None
for l in body.split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):\s*;?\s*(.*)$', l)
    if mm:
        lab, com = mm.group(1), mm.group(2)
        if 'Loop Header' in com and 'Inner' in com: cur = lab; loops.setdefault(cur, [])
        elif 'in Loop: Header=' in com:
            hdr = '.L' + re.search(r'Header=(BB\d+_\d+)', com).group(1)
            cur = hdr if hdr in loops else None
        else: cur = None
        continue
    if cur and l.strip() and not l.strip().startswith(';'): loops[cur].append(l)
for lab, b in loops.items():
    v = [l for l in b if re.match(r'\s+v_', l)]
    if len(v) < 50: continue
    f64 = [l for l in v if '_f64' in l]
    print(lab, 'instr', len(b), 'VALU', len(v), 'f64', len(f64), 'rcp', sum('v_rcp_f64' in l for l in v),
          'lane', sum(('readlane' in l or 'writelane' in l) for l in v), 'mov', sum('v_mov' in l for l in v), 'vmem', sum(bool(re.match(r'\s+global_', l)) for l in b),
          'lds', sum(bool(re.match(r'\s+ds_', l)) for l in b), 'smem', sum(bool(re.match(r'\s+s_load', l)) for l in b))
