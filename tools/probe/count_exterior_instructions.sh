#!/bin/bash
# GPU box: VALU wave-instructions of the grid launch with 2 interior nodes (everything but the march), per point
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/prof_ext -- python3 $R/tools/probe/time_exterior_share.py > $R/gpurun_out/prof_ext.log 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$R/gpurun_out/prof_ext/*/*counter_collection.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "shoot_grid_kernel" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[(r["Dispatch_Id"])].append((r["Counter_Name"], float(r["Counter_Value"])))
vals = sorted({(d, dict(v).get("SQ_INSTS_VALU", 0)) for d, v in by.items()}, key=lambda x: int(x[0]))
print([round(v / (16777216 / 64) / 4, 1) for _, v in vals])   # VALU instructions per point (4 points per lane)
PY
