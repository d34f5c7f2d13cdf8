#!/bin/bash
# tools/c5_overlap_timeline.sh [M] -- the kernel timeline of tools/c5_timeline.sh for the call on an own-stream context with launch overlap on (tools/c5_overlap.py): repeated calls
# neither fork nor join, so the fused launches of consecutive calls follow each other end to start
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
out=/tmp/c5otl; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o x -- python $R/tools/c5_overlap.py ${1:-512} 30 2>/dev/null | tail -n 1)
python3 - $out <<'PY'
import csv, sys, glob, re
rows=[]
for f in glob.glob(sys.argv[1] + "/**/x_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", r["Kernel_Name"].replace("rd::", ""))[:70], r["Grid_Size_X"]))
rows.sort()
tail = rows[-16:]
t0 = tail[0][0]
for s, e, n, g in tail:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} us grid {g:>7s} {n}")
PY
