#!/bin/bash
# tools/ktrace.sh <tag> <command ...> -- on the GPU box: rocprofv3 kernel trace of a command, per (kernel, grid size) count / median / min / mean
# duration in microseconds -> gpurun_out/<tag>.ktrace.txt (kernel names shortened)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
out=/tmp/ktrace_$tag; rm -rf $out; mkdir -p $out $R/gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o x -- "$@" > $R/gpurun_out/$tag.stdout 2> $R/gpurun_out/$tag.stderr)
python3 - $out > $R/gpurun_out/$tag.ktrace.txt <<'PY'
import csv, sys, glob, collections, re
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/x_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = re.sub(r"rd::", "", name)
        short = re.sub(r"\(.*$", "", short)[:150]
        agg[(short, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{len(v):7d} med {v[len(v)//2]/1e3:9.2f} min {v[0]/1e3:9.2f} mean {sum(v)/len(v)/1e3:9.2f} us  grid {k[1]:8d} wg {k[2]:4d}  {k[0]}")
PY
