#!/usr/bin/env python3
"""Per-kernel resource usage from a hipcc -Rpass-analysis=kernel-resource-usage log.  usage: tools/kres.py <log> [name regex]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cur = None
rows = {}
for l in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z][\w /\[\]]*?): (\d+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
for n, d in zip(names, dem):
    short = re.sub(r"\(.*", "", d)
    if pat and not pat.search(short):
        continue
    r = rows[n]
    print(f"{short[:70]:70s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):3d} SGPR {r.get('TotalSGPRs', r.get('SGPRs', -1)):4d} scratch {r.get('ScratchSize [bytes/lane]', -1):5d} occ {r.get('Occupancy [waves/SIMD]', -1):2d} LDS {r.get('LDS Size [bytes/block]', -1):6d}")
