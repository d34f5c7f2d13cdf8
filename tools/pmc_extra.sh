#!/bin/bash
# tools/pmc_extra.sh <tag> -- extra PMC passes on the bench command (each counter group in its own run, no trace domains):
# LDS bank conflicts, VALU instruction counts / busy cycles of the flagship kernel.  Output: gpurun_out/<tag>/pmc_*/
tag=${1:-r01}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  name=$(echo $grp | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -o x -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/pmc_$name.log 2>&1)
  f=$out/pmc_$name/x_counter_collection.csv
  if [ -f $f ]; then python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "jac_ad_split_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k}: mean {sum(v)/len(v):.4g} over {len(v)} launches")
PY
  else echo "no output for $grp"; tail -3 $out/pmc_$name.log; fi
done
find $out -name "*.db" -delete
