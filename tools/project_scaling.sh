#!/bin/bash
# Run on the GPU box: per-rank compute time of a strong-scaling run, measured on ONE GPU by giving it rank 0's tile
# (python bench.py --share-of E); prints E, ms per step and the projected whole-job rate (collective not included).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for E in 1 2 4 8; do
  python $R/bench.py --share-of $E --no-cpu-baseline --no-extra-mode --steps 40 --warmup 3 > $R/gpurun_out/share_of_$E.json 2> $R/gpurun_out/share_of_$E.err || exit 1
done
python - <<PY
import json
for E in (1, 2, 4, 8):
    j = json.loads(open("$R/gpurun_out/share_of_%d.json" % E).read().strip().splitlines()[-1])
    print(E, "rows", j["config"]["k_rows_per_gpu"], "ms/step %.3f" % j["ms_per_step"], "grid ms %.3f" % j["roofline"]["avg_launch_ms_unshared"],
          "brackets", j["config"]["brackets_per_step"], "projected %.3e" % j.get("projected_whole_job_value", j["value"]))
PY
