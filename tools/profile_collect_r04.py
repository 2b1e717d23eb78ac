"""Reduces the rocprofv3 outputs of tools/profile_r04.sh (gpurun_out/prof4) into profiles/r04/.  Per config: the profiler's own kernel
statistics (all launches), a summary over the TIMED launches only (the last `steps` dispatches of the step's kernels in the kernel trace:
median / mean / min / max -- the statistics file also averages the cold-clock warm-up launches), FETCH_SIZE / WRITE_SIZE per launch with
the guide's gfx950 correction (FETCH_SIZE x 2), and the SQ counters."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof4")
dst = os.path.join(ROOT, "profiles", "r04")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    fs = sorted(glob.glob(os.path.join(src, sub, "**", pat), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit("missing %s/%s" % (sub, pat))
    return fs[-1]


def short(name):
    return name.split("(")[0].replace("void viekf::", "")


def timed_summary(tag, kernels, steps):
    """per step kernel: the durations of its last `steps` launches (the timed region of bench.py: nothing runs after it with --no-secondary)"""
    rows = list(csv.DictReader(open(find(tag + "_stats", "*kernel_trace.csv"))))
    out = {}
    for k in kernels:
        d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if k in r["Kernel_Name"]]
        d.sort()
        if not d:
            continue
        last = sorted(x[1] for x in d[-steps:])
        out[k] = {"launches_all": len(d), "launches_timed": len(last), "median_ns": last[len(last) // 2], "mean_ns": sum(last) / len(last),
                  "min_ns": last[0], "max_ns": last[-1]}
    return out


def counter(sub, name, kernels):
    acc = {}
    for r in csv.DictReader(open(find(sub, "*counter_collection.csv"))):
        if r["Counter_Name"] != name:
            continue
        for k in kernels:
            if k in r["Kernel_Name"]:
                acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in acc.items()}


def config(tag, B, N, kernels, steps, with_sq):
    shutil.copy(find(tag + "_stats", "*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    ts = timed_summary(tag, kernels, steps)
    line = [l for l in open(os.path.join(src, tag + "_stats.log")) if l.startswith("{")]
    bench = json.loads(line[-1]) if line else {}
    json.dump({"config": {"batch": B, "n_feat": N}, "command": "tools/profile_r04.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py ... --steps %d" % steps,
               "timed_launches": ts, "bench_ms_per_step_under_profiler": bench.get("ms_per_step"),
               "kernel": bench.get("roofline", {}).get("kernel")}, open(os.path.join(dst, tag + "_kernel_summary.json"), "w"), indent=1)
    fe, wr = counter(tag + "_fetch", "FETCH_SIZE", kernels), counter(tag + "_write", "WRITE_SIZE", kernels)
    traffic = sum(2 * v["mean"] * 1024.0 for v in fe.values()) + sum(v["mean"] * 1024.0 for v in wr.values())
    json.dump({"config": {"batch": B, "n_feat": N}, "command": "tools/profile_r04.sh (separate rocprofv3 --pmc passes)",
               "per_kernel_kb": {"FETCH_SIZE": fe, "WRITE_SIZE": wr},
               "note": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so it is doubled "
                       "(calibrated for 16-B/lane streams; the fused kernel reads P with 8-B strided loads); WRITE_SIZE as is; per launch of each kernel, summed over the step's kernels",
               "traffic_bytes_per_launch": traffic, "traffic_bytes_per_step": traffic}, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
    if with_sq:
        sq = {}
        for sub in ("_sq1", "_sq2", "_sq3"):
            try:
                f = find(tag + sub, "*counter_collection.csv")
            except SystemExit:
                continue
            acc = {}
            for r in csv.DictReader(open(f)):
                if any(k in r["Kernel_Name"] for k in kernels):
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                sq[k] = sum(v) / len(v)
        json.dump({"config": {"batch": B, "n_feat": N}, "mean_per_launch": sq}, open(os.path.join(dst, tag + "_pmc_sq.json"), "w"), indent=1)
        print(tag, "SQ (millions per launch):", {k: round(v / 1e6, 2) for k, v in sq.items()})
    print(tag, {k: (v["median_ns"], round(v["mean_ns"])) for k, v in ts.items()}, "traffic MB %.1f" % (traffic / 1e6), "bench ms", bench.get("ms_per_step"))


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_default_run.json"))
config("resident_B1024_N50", 1024, 50, ["k_step_resident"], 50, True)
config("resident_B256_N25", 256, 25, ["k_step_resident"], 200, True)
config("stream_B1024_N150", 1024, 150, ["k_propagate_wide", "k_update_feat_panelsvc"], 20, False)
for sub, name in (("seq_stats", "seq_cadence_B1024_N50"), ("seqi_stats", "seq_cadence_independent_B1024_N50")):
    shutil.copy(find(sub, "*kernel_stats.csv"), os.path.join(dst, name + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "seq_shared.txt"), os.path.join(dst, "seq_cadence_B1024_N50_bench.txt"))
shutil.copy(os.path.join(src, "seq_indep.txt"), os.path.join(dst, "seq_cadence_independent_B1024_N50_bench.txt"))
