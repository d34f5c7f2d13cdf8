# rocprofv3 kernel durations of single small batches: column-split kernel (RDHIP_TUNE small_max = workgroups per CU up to which it is used) vs tile loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in "colsplit 1" "colsplit 2" "colsplit 4" "tile 1"; do
set -- $m
out=/tmp/small_$1_$2; mkdir -p $out
(cd /tmp && RDHIP_TUNE="small=$1,small_max=$2" rocprofv3 --kernel-trace --output-format csv -d $out -o x -- python $R/tools/small_probe.py > /dev/null 2>&1)
python - $out/x_kernel_trace.csv "$1$2" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "jac_ad" in r["Kernel_Name"] and "double, 0" in r["Kernel_Name"] and ">, 3>" in r["Kernel_Name"][:120]:
        agg[(r["Kernel_Name"][9:31], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: kv[0][1]):
    v.sort(); print(sys.argv[2], k, len(v), "median", v[len(v)//2], "min", v[0])
PY
done
