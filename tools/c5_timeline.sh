#!/bin/bash
# tools/c5_timeline.sh -- kernel timeline (rocprofv3 kernel trace) of the mixed-model call: start / end of each kernel of a few consecutive calls
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
out=/tmp/c5tl; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o x -- python $R/tools/bench_configs.py --only c5 --reps 30 > /dev/null 2>&1)
python3 - $out <<'PY'
import csv, sys, glob, re
rows=[]
for f in glob.glob(sys.argv[1] + "/**/x_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", r["Kernel_Name"].replace("rd::", ""))[:70], r["Grid_Size_X"]))
rows.sort()
big = [i for i, r in enumerate(rows) if "fused" in r[2] and int(r[3]) > 8192]
last = big[-1]
tail = rows[max(0, last - 7):last + 4]
t0 = tail[0][0]
for s, e, n, g in tail:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} us grid {g:>7s} {n}")
PY
