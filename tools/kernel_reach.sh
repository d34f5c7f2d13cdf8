#!/bin/bash
# tools/kernel_reach.sh <tag> -- which of the library's kernels does the GPU test suite actually launch?  rocprofv3 --kernel-trace --stats over
# `pytest -m gpu`, set against the host stubs in librdhip.so (nm): gpurun_out/<tag>/kernel_reach.txt lists, per kernel TEMPLATE, how many instantiations
# the library carries and how many of them the suite launched, then the instantiations it never launched (untested or unreachable: the candidates
# for pruning the 110 MB library -- VERDICT r04 item 6).
tag=${1:-reach}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o reach -- python -m pytest $root/tests -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1)
tail -2 $out/pytest.log
nm -C $root/robotdynamics.jl_amd/lib/librdhip.so | grep "__device_stub__" | sed 's/^[0-9a-f]* . //' > $out/stubs.txt
python3 - $out <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
def norm(s):
    s = s.replace("__device_stub__", "").strip()
    s = re.sub(r"^void ", "", s)
    depth, cut = 0, len(s)                      # cut the argument list: the last top-level '(' 
    for i, ch in enumerate(s):
        if ch == '<': depth += 1
        elif ch == '>': depth -= 1
        elif ch == '(' and depth == 0: cut = i; break
    return re.sub(r"\s+", "", s[:cut])
stubs = sorted({norm(l) for l in open(f"{out}/stubs.txt") if l.strip()})
hit = collections.Counter()
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        hit[norm(r["Name"])] += int(r["Calls"])
tmpl = lambda n: n.split("<", 1)[0]
per = collections.defaultdict(lambda: [0, 0])
never = []
for s in stubs:
    per[tmpl(s)][0] += 1
    if hit.get(s, 0) > 0: per[tmpl(s)][1] += 1
    else: never.append(s)
unknown = sorted(n for n in hit if n not in set(stubs) and n.startswith("rd::"))
with open(f"{out}/kernel_reach.txt", "w") as fh:
    fh.write(f"{len(stubs)} kernel instantiations in librdhip.so, {sum(1 for s in stubs if hit.get(s, 0) > 0)} launched by `pytest -m gpu` ({sum(hit.values())} launches traced)\n\n")
    fh.write("template                                   in library   launched by the suite\n")
    for t, (a, b) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        fh.write(f"{t:42s} {a:10d} {b:10d}\n")
    if unknown:
        fh.write(f"\n{len(unknown)} traced rd:: kernels without a stub in the library (hipRTC plug-in instantiations):\n")
        for n in unknown[:40]: fh.write("  " + n[:200] + "\n")
    fh.write(f"\nnever launched by the suite ({len(never)}):\n")
    for n in never: fh.write("  " + n + "\n")
print(open(f"{out}/kernel_reach.txt").read()[:3000])
PY
find $out -name "*.db" -delete
find $out -name "*kernel_trace.csv" -delete        # (half a million rows with the sweep in the suite: over gpurun's 64 MiB return limit; the stats file has the counts)
