"""Summarise gpurun_out/prof_*_<tag> (rocprofv3 csv) into profiles/<tag>_*: kernel stats csv + PMC json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
for f in glob.glob(os.path.join(go, f"prof_stats_{tag}", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(pr, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(go, f"bench_{tag}.json"), os.path.join(pr, f"{tag}_bench_line.json"))
out = {}
for name in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(go, f"prof_{name}_{tag}", "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if not any(s in kn for s in ("shoot", "refine", "bracket", "es_block", "slab_", "worker")):
                continue
            agg[r["Counter_Name"]][kn[:90]].append(float(r["Counter_Value"]))
        for cn, d in agg.items():
            key = "mean_KB_per_dispatch" if cn in ("FETCH_SIZE", "WRITE_SIZE") else "mean_per_dispatch"
            out[cn] = {k: {"dispatches": len(v), key: sum(v) / len(v)} for k, v in d.items()}
json.dump(out, open(os.path.join(pr, f"{tag}_bench_pmc.json"), "w"), indent=1)
hbm = {k: out[k] for k in ("FETCH_SIZE", "WRITE_SIZE") if k in out}
for k in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):                       # bench.py's valu_issue: cycles and instructions per grid launch
    if k in out:
        hbm[k] = {n: v for n, v in out[k].items() if "shoot_grid_kernel" in n}
hbm["round"] = tag
json.dump(hbm, open(os.path.join(pr, "bench_pmc_hbm_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
