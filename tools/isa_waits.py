#!/usr/bin/env python3
"""ISA audit: vector-memory waits that sit right behind the load they wait for (an exposed round trip).

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -S --cuda-device-only -o /tmp/x.s neuma_amd/csrc/nm_mpm.hip
    python tools/isa_waits.py /tmp/x.s [kernel-name-substring] [max distance, default 40]

Walks each kernel's instructions in layout order (control flow ignored: approximate), remembers where every global / buffer /
flat load (and returning atomic) was issued, and at each `s_waitcnt vmcnt(N)` looks up the youngest load that wait covers (all
but the last N).  Prints the waits whose covered load was issued fewer than `max distance` instructions earlier, with the
source-ish context (nearest label).  A wave that is alone on its SIMD, or a latency-bound kernel, pays each of them in full;
the usual causes are a register re-pack of a vector load's result (phi with a constant), a load under a lane predicate
(branch region: the compiler's bookkeeping then drains everything), or a load sunk next to its use."""
import re, sys
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
maxd = int(sys.argv[3]) if len(sys.argv) > 3 else 40
kern, idx, loads, label = None, 0, [], ""
out = {}
for ln, l in enumerate(src):
    m = re.match(r"^(_Z[\w]+|k_\w+):", l)
    if m:
        kern, idx, loads, label = m.group(1), 0, [], ""
        continue
    if kern is None:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        label = m.group(1)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if t.startswith("s_endpgm"):
        kern = None
        continue
    idx += 1
    op = t.split()[0]
    if re.match(r"(global|buffer|flat|scratch)_load", op) or (re.match(r"(global|buffer|flat)_atomic", op) and " sc0" in t):
        loads.append((idx, ln + 1, t))
    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
    if m and loads:
        n = int(m.group(1))
        if n < len(loads):
            li, lln, lt = loads[-(n + 1)]
            d = idx - li
            if d < maxd and pat in kern:
                out.setdefault(kern, []).append((ln + 1, label, n, d, lln, lt))
        loads = loads[len(loads) - n:] if n else []
for k, v in out.items():
    print(f"== {k}: {len(v)} close waits")
    for ln, label, n, d, lln, lt in v:
        print(f"   line {ln:6d} {label:12s} vmcnt({n}) {d:3d} instructions behind line {lln}: {lt[:70]}")
