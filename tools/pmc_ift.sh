export TMPDIR=/tmp
root=$PWD; out=$root/gpurun_out/ift_pmc; mkdir -p $out
for nb in 1 4; do for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && rocprofv3 --pmc $c --output-format csv -d $out/${c}_$nb -o x -- python $root/tools/ift_probe.py $nb 12 > $out/${c}_$nb.log 2>&1)
python - $out/${c}_$nb/x_counter_collection.csv $c $nb <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if len(v) > 5: print(sys.argv[3], k, f"mean {sum(v)/len(v):.5g} last {v[-1]:.5g} n={len(v)}")
PY
done; done
find $out -name "*.db" -delete
