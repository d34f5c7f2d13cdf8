#!/bin/bash
# tools/pmc_valu.sh <tag> -- the SECOND roofline of the VALU-bound kernels ("report the fp64 VALU roofline fraction next to the HBM one"):
# SQ counters of tools/pmc_valu.py's launches (rocprofv3 --pmc, one pass: the SQ block has 8 slots) -> gpurun_out/<tag>/valu_roofline.md
#   VALU instructions per launch (SQ_INSTS_VALU, per wave), the cycles the vector pipes were busy with them (SQ_ACTIVE_INST_VALU, quad-cycles), the waves'
#   resident cycles (SQ_WAVE_CYCLES) and where they went (SQ_WAIT_ANY: parked at s_waitcnt / barriers; SQ_WAIT_INST_ANY: issue stalls).
# fp64 vector peak of the part: 78.6 TF = 256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2.4 GHz x 2: a wave64 fp64 instruction occupies its SIMD for 4 cycles.
tag=${1:-r06valu}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o sq -- python $root/tools/pmc_valu.py > $out/cases.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python $root/tools/pmc_valu.py > /dev/null 2>&1)
find $out -name "*.db" -delete
python3 - $out <<'PY'
import csv, glob, sys, collections, statistics
out = sys.argv[1]
f = glob.glob(f"{out}/sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])) if f else []:
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for g in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e3
cases = [l.strip().split(" | ") for l in open(f"{out}/cases.log") if " | " in l]
lines = ["| case | kernel time us | VALU instr / launch (wave-level) | fp64-equivalent issue time at full rate us | fraction of the VALU roofline | VALU pipes busy (counter) | waves parked / issue-stalled / issuing |",
         "|---|---|---|---|---|---|---|"]
for tag, kname in cases:
    key = next((k for k in agg if k.startswith(kname[:120])), None)
    if key is None: continue
    c = {n: statistics.median(v) for n, v in agg[key].items()}
    us = next((d for k, d in dur.items() if k.startswith(kname[:120])), float("nan"))
    insts = c.get("SQ_INSTS_VALU", float("nan"))
    # every VALU instruction of these kernels' column code is fp64 (4 cycles on its SIMD); 1 024 SIMDs at 2.4 GHz
    full = insts * 4 / 1024 / 2400.0
    wc = c.get("SQ_WAVE_CYCLES", float("nan"))
    lines.append(f"| {tag} | {us:.1f} | {insts:,.0f} | {full:.1f} | **{full / us:.2f}** | {c.get('SQ_ACTIVE_INST_VALU', float('nan')) * 4 / max(c.get('SQ_BUSY_CYCLES', float('nan')), 1):.2f} of SQ_BUSY_CYCLES | "
                 f"{c.get('SQ_WAIT_ANY', 0) / wc:.2f} / {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} / {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} |")
open(f"{out}/valu_roofline.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
