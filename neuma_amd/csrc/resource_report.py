#!/usr/bin/env python3
"""Print per-kernel register / LDS / occupancy usage of the HIP sources (hipcc -Rpass-analysis)."""
import re, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
srcs = sys.argv[1:] or ["nm_mpm.hip", "nm_material.hip", "nm_bind.hip", "nm_raster.hip", "nm_rollout.hip"]
for src in srcs:
    if not os.path.exists(os.path.join(here, src)):
        continue
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                          "-I../../include", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                         cwd=here, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print(f"== {src}")
    for k, v in rows.items():
        if "rocprim" in k:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)[:40]
        print(f"  {name:40s} vgpr {v.get('VGPRs',0):4d} agpr {v.get('AGPRs',0):4d} sgpr {v.get('SGPRs',0):4d} "
              f"spillV {v.get('VGPRs Spill',0):3d} scratch {v.get('ScratchSize',0):4d} lds {v.get('LDS Size',0):6d} occ {v.get('Occupancy',0)}")
