#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/gpu_final.sh (gpurun_out/final/prof/{trace,fetch,write,sq}) into the tracked
summaries under profiles/:  <tag>_kernel_stats_B<batch>.csv (the --stats table), <tag>_pmc_B<batch>.json (average
counters per launch for the hot kernels) and hbm_traffic.json (what bench.py reports as roofline.traffic).
usage: summarize_prof.py <tag> <batch> <hot kernel substring> [more kernel substrings...]"""
import csv, glob, json, os, shutil, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, batch, hot = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kernels = sys.argv[3:]
base = os.path.join(REPO, "gpurun_out", "final", "prof")

def newest(pattern, needle):
    for f in sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True):
        if needle in open(f).read():
            return f
    raise SystemExit(f"no file matching {pattern} mentions {needle}")

stats = newest(os.path.join(base, "trace", "*", "*_kernel_stats.csv"), hot)
shutil.copy(stats, os.path.join(REPO, "profiles", f"{tag}_kernel_stats_B{batch}.csv"))
avg_ns = {}
for row in csv.DictReader(open(stats)):
    for k in kernels:
        if k in row["Name"]:
            avg_ns[k] = float(row["AverageNs"])
counters = {k: {} for k in kernels}
for p in ("fetch", "write", "sq"):
    try:
        f = newest(os.path.join(base, p, "*", "*_counter_collection.csv"), hot)
    except SystemExit:
        continue
    acc = {}
    for row in csv.DictReader(open(f)):
        for k in kernels:
            if k in row["Kernel_Name"]:
                acc.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    # one row per (dispatch, counter[, dimension]): sum the rows of a dispatch, average over dispatches
    per = {}
    for row in csv.DictReader(open(f)):
        for k in kernels:
            if k in row["Kernel_Name"]:
                per.setdefault((k, row["Counter_Name"]), {}).setdefault(row["Dispatch_Id"], 0.0)
                per[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (k, c), d in per.items():
        counters[k][c] = sum(d.values()) / len(d)
traffic = {}
tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
if os.path.exists(tpath):
    try:
        old = json.load(open(tpath))
        traffic = old.get("kernels", {}) if "kernels" in old else {}
    except Exception:
        traffic = {}
for hot in kernels:
  h = counters[hot]
  if "FETCH_SIZE" in h and "WRITE_SIZE" in h:
    # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64-byte units as 32 -> x2
    h["hbm_bytes_per_launch_corrected"] = (2 * h["FETCH_SIZE"] + h["WRITE_SIZE"]) * 1024
    extra = {}
    if "SQ_ACTIVE_INST_VALU" in h and "SQ_WAVE_CYCLES" in h:
        # share of the resident wavefronts' cycles in which a VALU instruction of theirs is executing, summed over the
        # wavefronts of a SIMD (waves_per_simd of them resident): how busy the vector ALUs are - what bounds this kernel
        wps = 2.0
        extra = {"valu_busy_frac": wps * h["SQ_ACTIVE_INST_VALU"] / h["SQ_WAVE_CYCLES"], "waves_per_simd": wps,
                 "valu_formula": "waves_per_simd * SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES"}
    traffic[hot] = {"batch": batch, "bytes_per_launch": h["hbm_bytes_per_launch_corrected"], "kernel": hot,
                    "source": f"profiles/{tag}_pmc_B{batch}.json", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024", **extra}
json.dump({"kernels": traffic}, open(tpath, "w"), indent=1)
h = counters[kernels[0]]
json.dump({"tag": tag, "batch": batch, "kernel_trace_avg_ns": avg_ns, "counters_avg_per_launch": counters,
           "note": "rocprofv3 --kernel-trace --stats pass and three separate --pmc passes of `python3 bench.py --batch "
                   f"{batch} --no-cpu-baseline --no-inverse` (tools/gpu_final.sh); FETCH_SIZE/WRITE_SIZE in KiB; FETCH doubled per MI355X_MICROARCH.md"},
          open(os.path.join(REPO, "profiles", f"{tag}_pmc_B{batch}.json"), "w"), indent=1)
print(json.dumps({"avg_ns": avg_ns, "hot": {k: v for k, v in h.items()}}, indent=1))
