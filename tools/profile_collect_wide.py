"""Reduces the rocprofv3 outputs of tools/profile_wide.sh (under gpurun_out/profw) into profiles/<round>/stream_B1024_N150_*."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "profw")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
tag = "stream_B1024_N150"


def find(sub, pat):
    fs = sorted(glob.glob(os.path.join(src, sub, "**", pat), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit("missing %s/%s" % (sub, pat))
    return fs[-1]


shutil.copy(find("stats", "*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))


def counter(sub, name):
    acc = {}
    for r in csv.DictReader(open(find(sub, "*counter_collection.csv"))):
        if r["Counter_Name"] != name or "viekf::k_" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0]
        if "k_propagate_" in k or "k_update_feat" in k:
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean_kb": sum(v) / len(v)} for k, v in acc.items()}


fe, wr = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
traffic = sum(2 * v["mean_kb"] * 1024.0 for v in fe.values()) + sum(v["mean_kb"] * 1024.0 for v in wr.values())
j = {"config": {"batch": 1024, "n_feat": 150},
     "command": "tools/profile_wide.sh (separate rocprofv3 --pmc passes)",
     "per_kernel_kb": {"FETCH_SIZE": fe, "WRITE_SIZE": wr},
     "note": "FETCH_SIZE is doubled for traffic on gfx950 (MI355X_MICROARCH.md); per launch of each kernel",
     "traffic_bytes_per_step": traffic}
json.dump(j, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
for r in list(csv.DictReader(open(os.path.join(dst, tag + "_kernel_stats.csv"))))[:2]:
    print(r["Name"][:50], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
print("traffic GB/step %.1f" % (traffic / 1e9), {k[:40]: round(v["mean_kb"] / 1e6, 2) for k, v in fe.items()}, {k[:40]: round(v["mean_kb"] / 1e6, 2) for k, v in wr.items()})
print(open(os.path.join(dst, tag + "_bench.json")).read()[:300])
