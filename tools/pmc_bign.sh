#!/bin/bash
# tools/pmc_bign.sh -- why is the size sweep of the flagship not monotone around 2 M knot points?  Address-translation and memory-side
# counters of the launches at 1 M / 2 M / 4 M knot points (each counter group in its own pass, no trace domains) -> gpurun_out/pmc_bign/summary.txt
root=$PWD
out=$root/gpurun_out/pmc_bign; mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_LEVEL_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o x -- python $root/tools/big_pmc.py > $out/g$i.log 2>&1)
done
find $out -name "*.db" -delete
python3 - $out > $out/summary.txt <<'PY'
import csv, sys, glob, collections
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/x_counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "jac_ad_split_kernel" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)
    for g, label in enumerate(("1M", "2M", "4M")):
        sel = ids[4 * g + 1:4 * g + 4]                      # launches 2-4 of each size
        if not sel: continue
        names = sorted(by[sel[0]])
        print(label, {n: sum(by[i][n] for i in sel) / len(sel) for n in names})
PY
cat $out/summary.txt
