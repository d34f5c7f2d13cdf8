#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: one line per kernel."""
import re, subprocess, sys
txt = sys.stdin.read()
cur = None; rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r":\s+(\w[\w \[\]/]*): (\d+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("rd::", "")
    if len(sys.argv) > 1 and not re.search(sys.argv[1], n): continue
    print(f"{n[:110]:110s} vgpr={r.get('VGPRs',0):4d} agpr={r.get('AGPRs',0):3d} sgpr={r.get('TotalSGPRs',0):3d} scratch={r.get('ScratchSize [bytes/lane]',0):5d} occ={r.get('Occupancy [waves/SIMD]',0)} lds={r.get('LDS Size [bytes/block]',0)}")
