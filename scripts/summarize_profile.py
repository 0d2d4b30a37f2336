#!/usr/bin/env python3
"""Condense a scripts/profile_r2.sh output directory (gpurun_out/prof_<tag>/) into
profiles/<tag>_*.{csv,json,md}: the rocprofv3 --kernel-trace --stats table, per-dispatch means of
the PMC passes for the hot kernel (k_play; k_decide in round 1), and the HBM traffic figure bench.py reports as
roofline.traffic (labelled with the configuration it was measured on).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half of a wide coalesced stream, so
it is doubled; the correction is calibrated for 16-B-per-lane accesses -- this kernel's traffic is
mostly 4-byte scratch and record accesses, so the figure is an upper-bound estimate)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

def HOT(name):
    return "k_play" in name or "k_decide" in name


tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
args_file = os.path.join(src, "bench_args.txt")
bench_args = open(args_file).read().strip() if os.path.exists(args_file) else "--steps 40 --warmup 20 --no-cpu"
def newest(pattern):
    """gpurun merges every call's files into the same directory: take the most recent run's."""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:] if files else []


stats = newest(f"{src}/trace/**/*_kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
means = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    files = newest(f"{src}/{name}/**/*_counter_collection.csv")
    if not files:
        continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    meta = {}
    for r in csv.DictReader(open(files[0])):
        if HOT(r["Kernel_Name"]):
            a = acc[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
    for k, (n, v) in acc.items():
        means[k] = {"dispatches": n, "mean_per_dispatch": v / n}
    means["_dispatch"] = meta
kd = [r for r in csv.DictReader(open(stats)) if HOT(r["Name"])][0]
# per-dispatch durations from the kernel trace: the last `launches` dispatches are bench.py's timed region
trace_csv = newest(f"{src}/trace/**/*_kernel_trace.csv")
timed_avg_ns = None
durs = []
if trace_csv:
    for r in csv.DictReader(open(trace_csv[0])):
        if HOT(r["Kernel_Name"]):
            durs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    durs.sort()
bench = {}
p = os.path.join(src, "bench_trace.json")
if os.path.exists(p) and os.path.getsize(p):
    bench = json.loads(open(p).read().strip().splitlines()[-1])
if bench and durs:
    k = bench["roofline"]["launches"]
    timed_avg_ns = sum(d for _, d in durs[-k:]) / k
out = {"tag": tag, "kernel": kd["Name"], "calls": int(kd["Calls"]), "avg_ns": float(kd["AverageNs"]),
       "timed_region_avg_ns": timed_avg_ns,
       "pmc": means, "bench_under_trace": bench}
if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
    out["hbm_bytes_per_launch"] = (2 * means["FETCH_SIZE"]["mean_per_dispatch"] + means["WRITE_SIZE"]["mean_per_dispatch"]) * 1024
def mean(name):
    return means[name]["mean_per_dispatch"] if name in means else None


# the ceilings that bind this kernel ("VALU utilisation / occupancy ... the meaningful ceilings")
ceil = {}
if mean("SQ_THREAD_CYCLES_VALU") and mean("SQ_ACTIVE_INST_VALU"):
    ceil["lane_utilisation"] = mean("SQ_THREAD_CYCLES_VALU") / mean("SQ_ACTIVE_INST_VALU")   # of 64 lanes per VALU instruction
if mean("SQ_ACTIVE_INST_VALU") and timed_avg_ns:
    # a wave64 VALU instruction occupies its SIMD for 4 clocks; 256 CUs x 4 SIMDs at 2.4 GHz
    ceil["valu_busy"] = mean("SQ_ACTIVE_INST_VALU") * 4 / (1024 * 2.4e9 * timed_avg_ns * 1e-9)
if mean("SQ_INSTS_SALU") and timed_avg_ns:
    ceil["salu_busy"] = mean("SQ_INSTS_SALU") / (256 * 2.4e9 * timed_avg_ns * 1e-9)   # one scalar unit per CU
if "_dispatch" in means and means["_dispatch"]:
    ceil["waves_per_cu"] = int(means["_dispatch"]["Grid_Size"]) / 64 / 256   # the persistent grid = the resident wavefronts
    ceil["scratch_bytes_per_lane"] = int(means["_dispatch"]["Scratch_Size"])
if mean("SQ_WAIT_INST_ANY") and mean("SQ_WAVE_CYCLES"):
    ceil["wave_time_in_waitcnt"] = mean("SQ_WAIT_INST_ANY") / mean("SQ_WAVE_CYCLES")
ceil["ceilings_measured_on"] = f"profiles/{tag}_summary.json (rocprofv3 --pmc passes of bench.py {bench_args}), offline like `traffic`"
out["ceilings"] = ceil
if "hbm_bytes_per_launch" in out and "--workload" not in bench_args:   # the headline workload only: bench.py quotes this file
    json.dump({"bytes_per_launch": out["hbm_bytes_per_launch"], "source": f"profiles/{tag}_summary.json",
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes, mean over the hot kernel's dispatches",
               "measured_on": f"bench.py {bench_args} rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, {tag}",
               "ceilings": ceil},
              open("profiles/traffic.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
with open(f"profiles/{tag}_summary.md", "w") as f:
    f.write(f"# rocprofv3 summary {tag}: `bench.py {bench_args}` (one MI355X)\n\n")
    f.write(f"kernel-trace --stats: `{kd['Name']}` calls {kd['Calls']}, average {float(kd['AverageNs'])/1e6:.3f} ms "
            f"(min {float(kd['MinNs'])/1e6:.3f}, max {float(kd['MaxNs'])/1e6:.3f}), {kd['Percentage']} % of GPU time\n\n")
    if bench:
        r = bench["roofline"]
        f.write(f"bench.py under the trace: {bench['value']/1e6:.1f} M env-steps/s, HIP-event average launch {r['avg_launch_ms']:.3f} ms "
                f"over the {r['launches']} timed launches (the stats row above also counts the warm-up launches, "
                f"which are early-game rounds with fewer legal actions; the kernel-trace average over the SAME last "
                f"{r['launches']} dispatches is {timed_avg_ns/1e6:.3f} ms)\n\n")
    f.write("| counter | mean per hot-kernel dispatch |\n|---|---|\n")
    for k, v in means.items():
        if k != "_dispatch":
            f.write(f"| {k} | {v['mean_per_dispatch']:.4g} |\n")
    f.write(f"\ndispatch: {means.get('_dispatch')}\n")
    if "hbm_bytes_per_launch" in out:
        f.write(f"\nHBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes): {out['hbm_bytes_per_launch']/1e9:.3f} GB\n")
print(json.dumps(out)[:400])
