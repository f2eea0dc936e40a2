#!/usr/bin/env python3
"""ISA audit: loops that pay a memory round trip per iteration.

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/x.s neuma_amd/csrc/nm_material.hip
    python tools/isa_loops.py /tmp/x.s

Prints, per kernel, every basic block the compiler marks as part of a loop whose body holds a global / buffer
load (or a returning atomic) AND an `s_waitcnt vmcnt`: a wave that is alone on its SIMD (the constitutive
kernels) or short of occupancy pays each of those in full.  Round 5 found the reverse constitutive kernel's
weight-gradient accumulate this way (22 serialised L2 round trips per thread).
"""
import re,sys,collections
# for every kernel: loops (by "Loop Header" comments) whose body has a global/buffer load AND a vmcnt wait -> candidates for serialized round trips
src=open(sys.argv[1]).read().split('\n')
kern=None; blocks=collections.OrderedDict(); cur=None
for l in src:
    m=re.match(r'^(_Z[\w]+|k_\w+):',l)
    if m: kern=m.group(1); continue
    m=re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)',l)
    if m:
        cur=(kern,m.group(1)); blocks[cur]={'hdr':m.group(2),'ins':[]}; continue
    if cur and re.match(r'\s+[a-z]',l): blocks[cur]['ins'].append(l.strip())
    if cur and 'Loop Header' in l and not blocks[cur]['hdr']: blocks[cur]['hdr']=l
    elif cur and l.strip().startswith(';') and ('Loop' in l): blocks[cur]['hdr']+=' '+l.strip()
for (k,b),d in blocks.items():
    ins=d['ins']
    gl=[i for i in ins if re.match(r'(global|buffer|flat)_load',i)]
    at=[i for i in ins if re.match(r'(global|buffer|flat)_atomic',i) and 'sc0' in i]
    w=[i for i in ins if 's_waitcnt' in i and 'vmcnt' in i]
    if ('Loop' in d['hdr']) and (gl or at) and w:
        print(k[:60],b,'| loads',len(gl),'ret-atomics',len(at),'vmwaits',[x.split('vmcnt')[1][:4] for x in w][:8],'| n',len(ins),'|',d['hdr'][:70])
