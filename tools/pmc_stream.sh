export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_stream; mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $out/$c -o s -- $root/robotdynamics.jl_amd/csrc/tune/tune_stream > $out/$c.log 2>&1
done
find $out -name "*.db" -delete
python3 - <<'P'
import csv,collections,glob,os
out=os.environ.get('OUT','/root/repo/gpurun_out/pmc_stream')
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{out}/{c}/**/*counter_collection.csv",recursive=True)
    if not f: print("none",c); continue
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k=("stream" if "stream" in r["Kernel_Name"] else "plain", r["Grid_Size"])
        agg.setdefault(k,[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(c,k,len(v),"mean KiB",sum(v)/len(v))
P
