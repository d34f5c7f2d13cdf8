#!/bin/bash
# tools/pmc_kernels.sh <tag> -- HBM byte counters of the non-flagship kernels (tools/pmc_kernels.py): FETCH_SIZE and WRITE_SIZE in separate --pmc passes,
# calibrated in the same session on exactly 1 GiB (tools/fp64_ubench calib), set against the algorithmic bytes.  Output: gpurun_out/<tag>/pmc_kernels.md
tag=${1:-r05pmc}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $out/k_$c -o k -- python $root/tools/pmc_kernels.py > $out/k_$c.log 2>&1)
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $out/cal_$c -o cal -- $root/tools/fp64_ubench calib > $out/cal_$c.log 2>&1)
done
find $out -name "*.db" -delete
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def counters(d):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])) if f else []:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
# calibration: fp64_ubench calib streams exactly 1 GiB through a 16-B/lane fill (write) and 16-B / 8-B per lane reads
cal_w, cal_r = counters("cal_WRITE_SIZE"), counters("cal_FETCH_SIZE")
GiB_KiB = 1024.0 * 1024.0
def factor(agg, needle):          # the largest launches of the kernel are the 1 GiB ones (tools/collect_profiles.py does the same)
    v = [x for k, vs in agg.items() if needle in k for x in vs]
    return GiB_KiB / max(v) if v else float("nan")
f_w = factor(cal_w, "fill16(")
f_r = factor(cal_r, "read16(")
kw, kr = counters("k_WRITE_SIZE"), counters("k_FETCH_SIZE")
cases = [json.loads(l) for l in open(f"{out}/k_WRITE_SIZE.log") if l.startswith("{")]
lines = [f"calibration (1 GiB streaming fill / read under the same --pmc pass): WRITE_SIZE x {f_w:.3f}, FETCH_SIZE x {f_r:.3f}", "",
         "| case | kernel | launches | written MB | read MB | traffic MB | algorithmic MB | ratio |", "|---|---|---|---|---|---|---|---|"]
rows = []
used = {}
for c in cases:
    for needle, alg in c["kernels"].items():
        w = [x for k, vs in kw.items() if needle in k for x in vs]
        r = [x for k, vs in kr.items() if needle in k for x in vs]
        if not w or not r: continue
        if needle == "rollout_split_lane_kernel":      # one kernel, two cases in a row (time-major, then trajectory-major): the launches in dispatch order, `reps` for the first case, the rest for the second
            n0 = used.get(needle, 0); n1 = n0 + c["reps"] if n0 == 0 else len(w)
            w, r = w[n0:n1], r[n0:n1]; used[needle] = n1
            if not w or not r: continue
        # (a kernel name may appear in several cases -- jac_ad_split_kernel --: cases run in order, take the launches by size class via the per-case count)
        wb = sorted(w)[len(w) // 2] * 1024 * f_w; rb = sorted(r)[len(r) // 2] * 1024 * f_r
        if needle == "jac_ad_split_kernel" or needle == "rollout_kernel" or needle == "jac_ad_small_kernel":
            # pick the launches whose written bytes are nearest the algorithmic write share
            wb = min((x * 1024 * f_w for x in w), key=lambda b: abs(b - alg)); i = [x * 1024 * f_w for x in w].index(wb); rb = r[i] * 1024 * f_r if i < len(r) else rb
        rows.append({"case": c["case"], "kernel": needle, "launches": len(w), "write_bytes": wb, "read_bytes": rb, "algorithmic_bytes": alg, "ratio": (wb + rb) / alg, "note": c.get("note", "")})
        lines.append(f"| {c['case']} | `{needle}` | {len(w)} | {wb / 1e6:.2f} | {rb / 1e6:.2f} | {(wb + rb) / 1e6:.2f} | {alg / 1e6:.2f} | **{(wb + rb) / alg:.3f}** |")
open(f"{out}/pmc_kernels.md", "w").write("\n".join(lines) + "\n")
json.dump({"write_factor": f_w, "read_factor": f_r, "rows": rows}, open(f"{out}/pmc_kernels.json", "w"), indent=1)
print("\n".join(lines))
PY
