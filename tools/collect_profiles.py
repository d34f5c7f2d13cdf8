#!/usr/bin/env python3
"""tools/collect_profiles.py <tag> [<dest>] -- copy the judged rocprofv3 summaries of gpurun_out/<tag>/ (written by
tools/gpu_profile.sh on the GPU box) into profiles/<dest>/ and derive profiles/hbm_traffic.json.

HBM traffic per launch = WRITE_SIZE x f_w + FETCH_SIZE x f_r (both counters are in KiB), where the correction factors come
from the calibration runs of the same session: tools/fp64_ubench `calib` streams exactly 1 GiB through a 16-B/lane fill and
16-B and 8-B/lane reads under the same --pmc pass (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half the
bytes of a coalesced streaming read; WRITE_SIZE must be calibrated on a known byte count)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dest = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join(ROOT, "gpurun_out", tag)
out = os.path.join(ROOT, "profiles", dest)
os.makedirs(out, exist_ok=True)


def counters(path):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def pick(agg, needle, counter):
    best = None
    for (k, c), v in agg.items():
        if needle in k and c == counter and (best is None or len(v) > len(best[1])):
            best = (k, v)
    return best


GIB_KIB = float(1 << 20)
summary = {"tag": tag}
cal_w = pick(counters(os.path.join(src, "cal_write", "cal_counter_collection.csv")), "fill16(", "WRITE_SIZE")
cal_r16 = pick(counters(os.path.join(src, "cal_fetch", "cal_counter_collection.csv")), "read16(", "FETCH_SIZE")
cal_r8 = pick(counters(os.path.join(src, "cal_fetch", "cal_counter_collection.csv")), "read8(", "FETCH_SIZE")
f_w = GIB_KIB / max(cal_w[1]) if cal_w else 1.0
f_r16 = GIB_KIB / max(cal_r16[1]) if cal_r16 else 2.0
f_r8 = GIB_KIB / max(cal_r8[1]) if cal_r8 else f_r16
summary["calibration"] = {"write_factor": f_w, "read16_factor": f_r16, "read8_factor": f_r8,
                          "method": "1 GiB streaming fill/read under the same --pmc pass; factor = true KiB / counter KiB"}
kern = None
for name in ("pmc_write", "pmc_fetch"):
    p = os.path.join(src, name, f"{tag}_counter_collection.csv")
    agg = counters(p)
    rows = []
    for (k, c), v in sorted(agg.items()):
        rows.append({"kernel": k, "counter": c, "launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)})
    summary[name] = rows
    with open(os.path.join(out, f"{name}_summary.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["kernel", "counter", "launches", "mean_KiB", "min_KiB", "max_KiB"])
        w.writeheader()
        w.writerows(rows)
stats = os.path.join(src, "trace", f"{tag}_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(out, "kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    kern = rows[0]["Name"]
    summary["dominant_kernel"] = {"name": kern, "calls": int(rows[0]["Calls"]), "avg_ns": float(rows[0]["AverageNs"]),
                                  "min_ns": float(rows[0]["MinNs"]), "max_ns": float(rows[0]["MaxNs"])}
# the same kernel also runs on 4 096-knot batches inside the bench command (the `small_batches_fused` comparison: grids of 64 workgroups), which
# drags the per-name average of kernel_stats.csv down; the launches of the timed workload are the ones with the full persistent grid
trace = os.path.join(src, "trace", f"{tag}_kernel_trace.csv")
if kern and os.path.exists(trace):
    rows = [r for r in csv.DictReader(open(trace)) if r["Kernel_Name"] == kern]
    gmax = max(int(r["Grid_Size_X"]) for r in rows)
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if int(r["Grid_Size_X"]) == gmax]
    summary["dominant_kernel"]["full_grid"] = {"grid_threads": gmax, "calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d),
                                               "note": "launches with the full persistent grid = the timed N = 65 536 batches; kernel_stats.csv averages over every launch of the name"}
# launch overlap: in the trace of the driver's command consecutive full-grid launches of the dominant kernel run on two streams and overlap in time.
# What the bench line's roofline.kernel_ms measures there is the time per launch of the JOB: the interval between consecutive launch starts.
if kern and os.path.exists(trace):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace)) if r["Kernel_Name"] == kern and int(r["Grid_Size_X"]) == gmax))
    ov_iv, ov_dur, sg_iv, sg_dur = [], [], [], []
    for (s0, e0), (s1, e1) in zip(rows, rows[1:]):
        if s1 - s0 > 200000:                       # a pause between phases of the command
            continue
        (ov_iv if s1 < e0 else sg_iv).append(s1 - s0)
        (ov_dur if s1 < e0 else sg_dur).append(e1 - s1)
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    summary["dominant_kernel"]["overlap_trace"] = {
        "launches_overlapping_their_predecessor": len(ov_iv), "median_start_to_start_ns": med(ov_iv), "mean_start_to_start_ns": (sum(ov_iv) / len(ov_iv)) if ov_iv else None,
        "median_duration_ns": med(ov_dur),
        "launches_not_overlapping": len(sg_iv), "median_start_to_start_ns_not_overlapping": med(sg_iv), "median_duration_ns_not_overlapping": med(sg_dur),
        "note": "overlap mode (bench default): the job's time per launch is the start-to-start interval; a launch's own duration is longer because two launches share the chip"}
single = os.path.join(src, "trace_single", f"{tag}_kernel_stats.csv")
if os.path.exists(single):
    shutil.copy(single, os.path.join(out, "kernel_stats_single_stream.csv"))
    for r in csv.DictReader(open(single)):
        if kern and r["Name"] == kern:
            summary["dominant_kernel"]["single_stream"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                                          "note": "bench.py --no-overlap: one stream, the per-kernel average duration IS the time per launch (agrees with the bench line's single_stream.kernel_ms)"}
for f in ("bench.json", "pytest_gpu.log"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(out, f))
if kern:
    key = kern.split("(")[0][:60]
    w = pick(counters(os.path.join(src, "pmc_write", f"{tag}_counter_collection.csv")), key, "WRITE_SIZE")
    r = pick(counters(os.path.join(src, "pmc_fetch", f"{tag}_counter_collection.csv")), key, "FETCH_SIZE")
    if w and r:
        # the bench command also launches this kernel on 4 096-knot batches (the `small_batches_fused` comparison): only the full-size
        # launches (counter above half of the largest reading) describe the timed workload
        full = lambda v: [x for x in v if x > 0.5 * max(v)]
        wf, rf = full(w[1]), full(r[1])
        summary["traffic_launches"] = {"write": len(wf), "fetch": len(rf), "all_write": len(w[1]), "all_fetch": len(r[1])}
        wb = sum(wf) / len(wf) * 1024 * f_w
        rb = sum(rf) / len(rf) * 1024 * f_r8              # the kernel's input loads are 8 B per lane
        summary["traffic"] = {"write_bytes_per_launch": wb, "read_bytes_per_launch": rb, "bytes_per_launch": wb + rb}
        json.dump({"bytes_per_launch": wb + rb, "write_bytes": wb, "read_bytes": rb, "kernel": kern, "source": f"profiles/{dest}/ (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes, calibrated)"},
                  open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
