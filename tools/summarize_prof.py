#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/gpu_final.sh (<prof>/{trace,fetch,write,sq}) into the summaries that are committed
under profiles/.  Every figure is kept PER KERNEL INSTANTIATION AND LAUNCH SHAPE: dispatches are grouped by the full kernel
name (template arguments included), the grid size and the workgroup size, so the one-ciphertext and 256-ciphertext latency
probes of bench.py never mix with the 8,192-ciphertext launches, and two instantiations of one template never share a row
(VERDICT r2, weak 7).

    <out>/<tag>_kernel_stats_B<batch>.csv     rocprofv3's own --stats table (copied)
    <out>/<tag>_kernel_groups_B<batch>.json   per (kernel, grid, workgroup): calls, avg / min / max ms from the kernel trace,
                                              average counters per dispatch from the three --pmc passes, HBM bytes per launch
                                              ((2 * FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md), VALU busy fraction
                                              (waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, waves per SIMD from the
                                              workgroup size: these kernels hold one workgroup per compute unit)
    <out>/hbm_traffic.json                    what bench.py reads for roofline.traffic: the blind-rotation groups

Every blind-rotation group also gets `cycles_per_cmux`: average launch time x the shader clock bench.py sampled during the SAME run
(the JSON line in the trace pass's log, 5th argument) / (steps x workgroup rounds per compute unit) - the quantity to compare between
boxes whose clocks differ, and with bench.py's `roofline.kernel_cycles_per_cmux`.

usage: summarize_prof.py <tag> <batch> [<prof dir> [<out dir> [<log of the traced bench run>]]]      (defaults: gpurun_out/final/prof, profiles)"""
import csv, glob, json, os, re, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, batch = sys.argv[1], int(sys.argv[2])
base = sys.argv[3] if len(sys.argv) > 3 else os.path.join(REPO, "gpurun_out", "final", "prof")
out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(REPO, "profiles")
os.makedirs(out_dir, exist_ok=True)
sclk_mhz, traced_line = None, None
if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
    for line in open(sys.argv[5]):
        if line.startswith("{") and '"roofline"' in line:
            try:
                traced_line = json.loads(line)
                sclk_mhz = float(traced_line["roofline"]["sclk_mhz"])
            except Exception:
                pass


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)
    return fs[0] if fs else None


def col(row, *names):
    for n in names:
        if n in row and row[n] != "":
            return row[n]
    raise KeyError(f"none of {names} in {list(row)[:30]}")


def short(name):
    s = re.sub(r"^void\s+", "", name.strip().strip('"'))
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):          # cut the argument list: the first '(' outside template brackets
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return s[:cut].strip().replace(" [clone .kd]", "").replace(".kd", "")


def cts_per_workgroup(kernel, wg):
    if "tp2u" in kernel: return 2                                                # two ciphertexts per workgroup sharing the key words
    if "lat" in kernel or "wide" in kernel or "quad" in kernel or "w_t64f" in kernel: return 1        # one workgroup per ciphertext
    if wg == 512: return 4                                                       # wave-pair kernels: four ciphertexts per workgroup
    if wg == 256: return 2
    return None


groups = {}


def grp(kernel, grid, wg):
    return groups.setdefault((kernel, int(grid), int(wg)), {"kernel": kernel, "grid_size": int(grid), "workgroup_size": int(wg), "dur_ns": [], "counters": {}})


trace = newest(os.path.join(base, "trace", "**", "*kernel_trace.csv")) or newest(os.path.join(base, "trace", "*", "*kernel_trace.csv"))
if trace is None:
    fs = glob.glob(os.path.join(base, "trace", "**", "*kernel_trace.csv"), recursive=True)
    trace = max(fs, key=os.path.getmtime) if fs else None
if trace is None:
    raise SystemExit(f"no kernel trace under {base}/trace")
for row in csv.DictReader(open(trace)):
    k = short(col(row, "Kernel_Name", "Name"))
    gx = int(col(row, "Grid_Size_X", "Grid_Size")) * int(row.get("Grid_Size_Y") or 1) * int(row.get("Grid_Size_Z") or 1)
    wx = int(col(row, "Workgroup_Size_X", "Workgroup_Size")) * int(row.get("Workgroup_Size_Y") or 1) * int(row.get("Workgroup_Size_Z") or 1)
    grp(k, gx, wx)["dur_ns"].append(int(col(row, "End_Timestamp")) - int(col(row, "Start_Timestamp")))

stats = glob.glob(os.path.join(base, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(out_dir, f"{tag}_kernel_stats_B{batch}.csv"))

for p in ("fetch", "write", "sq"):
    fs = glob.glob(os.path.join(base, p, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    per = {}
    for row in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = short(col(row, "Kernel_Name"))
        key = (k, int(col(row, "Grid_Size", "Grid_Size_X")), int(col(row, "Workgroup_Size", "Workgroup_Size_X")))
        d = per.setdefault((key, col(row, "Counter_Name")), {})
        # one row per (dispatch, counter[, dimension]): sum the rows of a dispatch, average over dispatches
        d[col(row, "Dispatch_Id")] = d.get(col(row, "Dispatch_Id"), 0.0) + float(col(row, "Counter_Value"))
    for (key, cname), d in per.items():
        g = grp(*key)
        g["counters"][cname] = sum(d.values()) / len(d)
        g.setdefault("counter_dispatches", {})[cname] = len(d)

rows = []
for g in groups.values():
    if not g["dur_ns"] and not g["counters"]:
        continue
    r = {"kernel": g["kernel"], "grid_size": g["grid_size"], "workgroup_size": g["workgroup_size"],
         "workgroups": g["grid_size"] // max(g["workgroup_size"], 1), "calls": len(g["dur_ns"])}
    cpw = cts_per_workgroup(g["kernel"], g["workgroup_size"]) if "blind_rotate" in g["kernel"] else None
    if cpw:
        r["ciphertexts_per_launch"] = r["workgroups"] * cpw
    if g["dur_ns"]:
        r.update(avg_ms=sum(g["dur_ns"]) / len(g["dur_ns"]) / 1e6, min_ms=min(g["dur_ns"]) / 1e6, max_ms=max(g["dur_ns"]) / 1e6)
        if cpw and sclk_mhz:
            steps = 742 if "w_t64f" in g["kernel"] else 630          # LWE dimension of the set the kernel serves in bench.py
            if "lat2u" in g["kernel"] or "tp2u" in g["kernel"]:
                steps = (steps + 1) // 2                             # the unrolled kernels take two coefficients per step
            rounds = -(-r["ciphertexts_per_launch"] // (cpw * 256))
            r.update(cycles_per_cmux=r["avg_ms"] * 1e-3 * sclk_mhz * 1e6 / (steps * rounds), sclk_mhz=sclk_mhz,
                     cycles_formula="avg_ms x sclk (sampled by bench.py in the traced run) / (steps x ceil(ciphertexts / (ciphertexts per workgroup x 256 CUs)))")
    c = g["counters"]
    if c:
        r["counters_avg_per_dispatch"] = c
        r["counter_dispatches"] = g.get("counter_dispatches", {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 -> x 2
        r["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        r["hbm_formula"] = "(2 * FETCH_SIZE + WRITE_SIZE) * 1024"
    if "SQ_ACTIVE_INST_VALU" in c and c.get("SQ_WAVE_CYCLES"):
        wps = max(1.0, g["workgroup_size"] / 64 / 4)      # one workgroup per compute unit (LDS-limited kernels): its waves over 4 SIMDs
        r.update(waves_per_simd=wps, valu_busy_frac=wps * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                 valu_formula="waves_per_simd * SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (waves_per_simd = workgroup size / 256: one workgroup per CU)")
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_ACTIVE_INST_LDS"):
        r["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"]
    rows.append(r)
rows.sort(key=lambda r: -(r.get("avg_ms", 0) * r["calls"]))
note = ("rocprofv3 --kernel-trace --stats pass and three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_*) of `python3 bench.py --batch "
        f"{batch} --no-cpu-baseline --no-inverse` (tools/gpu_final.sh); one row per (kernel instantiation, grid size, workgroup size); "
        "FETCH_SIZE / WRITE_SIZE in KiB, FETCH doubled per MI355X_MICROARCH.md")
if traced_line is not None:
    rl = traced_line.get("roofline", {})
    note += (f"; the traced run's own line: value {traced_line.get('value'):.0f} PBS/s, kernel_ms {rl.get('kernel_ms'):.3f} (events), "
             f"kernel_cycles_per_cmux {rl.get('kernel_cycles_per_cmux'):.0f}, sclk {sclk_mhz:.0f} MHz")
json.dump({"tag": tag, "batch": batch, "note": note, "groups": rows}, open(os.path.join(out_dir, f"{tag}_kernel_groups_B{batch}.json"), "w"), indent=1)
entries = [{"kernel": r["kernel"], "grid_size": r["grid_size"], "workgroup_size": r["workgroup_size"], "batch": r.get("ciphertexts_per_launch"),
            "avg_ms": r.get("avg_ms"), "calls": r["calls"], "bytes_per_launch": r.get("hbm_bytes_per_launch"),
            "valu_busy_frac": r.get("valu_busy_frac"), "waves_per_simd": r.get("waves_per_simd"), "cycles_per_cmux": r.get("cycles_per_cmux"),
            "source": f"profiles/{tag}_kernel_groups_B{batch}.json"}
           for r in rows if "blind_rotate" in r["kernel"]]
json.dump({"entries": entries, "note": "per (kernel instantiation, launch shape); bench.py matches kernel AND batch"},
          open(os.path.join(out_dir, "hbm_traffic.json"), "w"), indent=1)
for r in rows[:14]:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k not in ("counters_avg_per_dispatch", "counter_dispatches", "hbm_formula", "valu_formula")}))
