"""Summarise gpurun_out/prof_*_<tag>[_<workload>] (rocprofv3 csv) into profiles/<tag>[_<workload>]_*: kernel stats csv + PMC
json + the bench line; the PMC summary is also written as profiles/pmc_latest_<workload>.json (and, for the headline,
profiles/bench_pmc_hbm_latest.json), which bench.py reads for roofline.traffic / valu_issue.
    python tools/summarize_profiles.py <tag> [config3|config1|config2|config4]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "config3"
suf = "" if wl == "config3" else "_" + wl
stem = "bench" if wl == "config3" else wl
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
for f in glob.glob(os.path.join(go, f"prof_stats_{tag}{suf}", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(pr, f"{tag}_{stem}_kernel_stats.csv"))
if os.path.exists(os.path.join(go, f"bench_{tag}{suf}.json")):      # absent on the first pass of tools/profile_bench.sh (PMC summary before the line)
    shutil.copy(os.path.join(go, f"bench_{tag}{suf}.json"), os.path.join(pr, f"{tag}_{stem}_line.json"))
out = {}
for name in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(go, f"prof_{name}_{tag}{suf}", "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if not any(s in kn for s in ("shoot", "refine", "bracket", "es_block", "slab_", "worker", "unsure", "scatter", "pack_")):
                continue
            agg[r["Counter_Name"]][kn[:90]].append(float(r["Counter_Value"]))
        for cn, d in agg.items():
            key = "mean_KB_per_dispatch" if cn in ("FETCH_SIZE", "WRITE_SIZE") else "mean_per_dispatch"
            out[cn] = {k: {"dispatches": len(v), key: sum(v) / len(v)} for k, v in d.items()}
json.dump(out, open(os.path.join(pr, f"{tag}_{stem}_pmc.json"), "w"), indent=1)
hbm = {k: out[k] for k in ("FETCH_SIZE", "WRITE_SIZE") if k in out}
for k in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):                       # bench.py's valu_issue: cycles and instructions per grid launch
    if k in out:
        hbm[k] = {n: v for n, v in out[k].items() if "shoot_grid" in n}
for k in ("FETCH_SIZE", "WRITE_SIZE"):
    if k in hbm:
        hbm[k] = {n: v for n, v in hbm[k].items() if "shoot_grid" in n}
hbm["round"] = tag
json.dump(hbm, open(os.path.join(pr, f"pmc_latest_{wl}.json"), "w"), indent=1)
if wl == "config3":
    json.dump(hbm, open(os.path.join(pr, "bench_pmc_hbm_latest.json"), "w"), indent=1)
print(json.dumps(hbm, indent=1)[:3000])
