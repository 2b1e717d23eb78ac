"""Reduces the rocprofv3 outputs of tools/profile_run.sh (under gpurun_out/prof) into profiles/<round>/resident_*."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
tag = "resident_B1024_N50"


def find(sub, pat):
    fs = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    if not fs:
        raise SystemExit("missing %s/%s" % (sub, pat))
    return max(fs, key=os.path.getmtime)   # (gpurun merges into gpurun_out: older runs' files stay around)


shutil.copy(find("stats", "*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))


def counter(sub, name, out):
    rows = [r for r in csv.DictReader(open(find(sub, "*counter_collection.csv"))) if r["Counter_Name"] == name and "k_step_resident" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows]
    with open(os.path.join(dst, out), "w") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in rows:
            w.writerow([r["Dispatch_Id"], r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"], name, r["Counter_Value"]])
    return {"counter": name, "launches": len(vals), "mean_kb": sum(vals) / len(vals), "min_kb": min(vals), "max_kb": max(vals)}


fe = counter("fetch", "FETCH_SIZE", tag + "_pmc_fetch.csv")
wr = counter("write", "WRITE_SIZE", tag + "_pmc_write.csv")
fb, wb = fe["mean_kb"] * 1024.0, wr["mean_kb"] * 1024.0
j = {"config": {"batch": 1024, "n_feat": 50, "kernel": "k_step_resident<7,3> (two 256-thread workgroups per CU)"},
     "command": "tools/profile_run.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py "
                "--steps 10 --warmup 2 --no-cpu-baseline (separate passes)",
     "raw": {"fetch": fe, "write": wr}, "fetch_bytes_raw": fb, "fetch_bytes_x2_gfx950": 2 * fb, "write_bytes": wb,
     "traffic_bytes_per_launch": 2 * fb + wb,
     "note": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so it is "
             "doubled; the rule is calibrated for 16-B/lane streams, this kernel reads P with 8-B strided loads (uncalibrated "
             "width). WRITE_SIZE is taken as is."}
json.dump(j, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
sq = {}
nl = 0
for sub in ("sq1", "sq2"):
    try:
        rows = [r for r in csv.DictReader(open(find(sub, "*counter_collection.csv"))) if "k_step_resident" in r["Kernel_Name"]]
    except SystemExit:
        rows = []
    acc = {}
    for r in rows:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        sq[k] = sum(v) / len(v)
        nl = len(v)
if sq:
    json.dump({"config": j["config"], "mean_per_launch": sq, "launches": nl}, open(os.path.join(dst, tag + "_pmc_sq.json"), "w"), indent=1)
    print("SQ:", {k: round(v / 1e6, 1) for k, v in sq.items()}, "(millions per launch)")
ks = [r for r in csv.DictReader(open(os.path.join(dst, tag + "_kernel_stats.csv"))) if "k_step_resident" in r["Name"]]
print("kernel avg ns", ks[0]["AverageNs"], "calls", ks[0]["Calls"], " traffic MB %.1f (fetch raw %.1f, write %.1f)" % ((2 * fb + wb) / 1e6, fb / 1e6, wb / 1e6))
print(open(os.path.join(dst, tag + "_bench.json")).read()[:400])
