export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/early_tr; mkdir -p $out
cd /tmp
for v in lib lib_ab; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -o x -- python $R/tools/ab_lib.py $R/robotdynamics.jl_amd/$v/librdhip.so $R/robotdynamics.jl_amd/$v/librdhip.so 1 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/h -o x -- $R/robotdynamics.jl_amd/csrc/tune/tune_jac_early2 65536 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
for d in ("h", "lib", "lib_ab"):
    for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + f"/gpurun_out/early_tr/{d}/**/x_kernel_stats.csv", recursive=True):
        print(d)
        for r in csv.DictReader(open(f)):
            if "jac_ad_split" in r["Name"]: print("  ", r["Name"][40:140].replace("rd::", ""), r["Calls"], r["AverageNs"], r["MinNs"])
PY
find $out -name "*.db" -delete
