export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k test_jacobian_multi_fuses 2>&1 | grep -v amdgpu | tail -25
for m in colsplit tile; do
out=$R/gpurun_out/small_$m; mkdir -p $out
(cd /tmp && RDHIP_SMALL=$m rocprofv3 --kernel-trace --stats --output-format csv -d $out -o x -- python $R/tools/small_probe.py > /dev/null 2>&1)
python - $out/x_kernel_trace.csv $m <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "jac_ad" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][9:60], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: kv[0][1]):
    v.sort(); print(sys.argv[2], k, len(v), "median", v[len(v)//2], "min", v[0])
PY
find $out -name "*.db" -delete
done
